// A whole SMALL 'pool' SAGEConv layer in one workgroup: forward in one launch, backward in one launch.
//
// The live layer (DGL SAGEConv(aggregator_type='pool'), imported at R/train/graphsage/pytorch/graphsage_dgl.py:3;
// parameterisation R/inference_optimized.py:136-139,260,276):
//     P = relu(h . Wp^T + bp)                  [n_src, Hin]
//     neigh[d] = max_j P[idx[d, j]]            (0 / no winner when slot 0 is -1; first slot wins ties)
//     y = act(h[:n_dst] . Ws^T + neigh . Wn^T + bs + bn)     [n_dst, Hout]
// At the settings the reference ships for its small datasets (R/settings/pubmed.json, arxiv.json: embedding_size 32, batch 32,
// i.e. the OUTPUT layer of a batch is n_src <= 832 rows x 32 features -> <= 40 classes) that is 1-2 MFLOP: as separate
// launches (fc_pool GEMM, reduce, bias add, skinny GEMM; backward: 2 weight gradients, 2 input gradients, zero fill,
// scatter, fc_pool input gradient, weight gradient + split-K reduce, add) it is 15 launches of 5-16 us of pure latency each —
// half of a 32-seed train step.  Here one 1024-thread workgroup keeps P (forward) / dh (backward) in LDS and walks the phases
// with __syncthreads() between them: exact fp32 FMA arithmetic on the vector ALU (no MFMA: the whole layer is ~1 us of
// arithmetic), LDS float atomics for the winner scatter of the backward.
//
// Limits (checked by the entry points): Hin <= 64, Hout <= 64, n_src * Hin floats + the small arrays within 160 KB of LDS.
#include "ogl_common.h"

#define SL_THREADS 1024
#define SL_MAX_H 64

static inline int64_t sl_lds_fwd(int64_t n_src, int64_t n_dst, int Hin, int Hout) {
  return 4 * (n_src * Hin + (int64_t)Hin * (Hin + 1) + n_dst * Hin) + 64;
}
static inline int64_t sl_lds_bwd(int64_t n_src, int64_t n_dst, int Hin, int Hout) {
  return 4 * (n_src * Hin + (int64_t)Hin * (Hin + 1) + 3 * n_dst * Hin + n_dst * Hout) + 64;
}

extern "C" int ogl_small_pool_layer_fits(int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout) {
  if (n_src <= 0 || n_dst <= 0 || n_dst > n_src || fanout <= 0 || Hin <= 0 || Hout <= 0) return 0;
  if (Hin > SL_MAX_H || Hout > SL_MAX_H || n_dst * (int64_t)Hin > 8192) return 0;
  const int64_t cap = 150 * 1024;
  return sl_lds_fwd(n_src, n_dst, Hin, Hout) <= cap && sl_lds_bwd(n_src, n_dst, Hin, Hout) <= cap;
}

__global__ void __launch_bounds__(SL_THREADS) k_small_pool_fwd(const float* __restrict__ h, int64_t ldh, int64_t n_src,
                                                               const int32_t* __restrict__ idx, int64_t n_dst, int S, int Hin,
                                                               const float* __restrict__ Wp, int64_t ldwp, const float* __restrict__ bp,
                                                               const float* __restrict__ Ws, int64_t ldws, const float* __restrict__ bs,
                                                               const float* __restrict__ Wn, int64_t ldwn, const float* __restrict__ bn,
                                                               int Hout, int relu_out, float* __restrict__ neigh_out, int64_t ldn,
                                                               int32_t* __restrict__ argmax_out, float* __restrict__ y, int64_t ldy) {
  extern __shared__ float sl_smem[];
  float* P = sl_smem;                                  // [n_src, Hin]
  float* W = P + n_src * Hin;                          // [Hin, Hin + 1]  (odd stride: conflict-free rows)
  float* NB = W + Hin * (Hin + 1);                     // [n_dst, Hin]
  const int tid = threadIdx.x;
  for (int i = tid; i < Hin * Hin; i += SL_THREADS) W[(i / Hin) * (Hin + 1) + i % Hin] = Wp[(int64_t)(i / Hin) * ldwp + i % Hin];
  __syncthreads();
  // P = relu(h . Wp^T + bp): thread -> (row r, feature j); the 32-64 threads of a row read the same h row (broadcast)
  for (int64_t o = tid; o < n_src * Hin; o += SL_THREADS) {
    const int64_t r = o / Hin;
    const int j = (int)(o - r * Hin);
    const float* hr = h + r * ldh;
    const float* wj = W + j * (Hin + 1);
    float acc = bp ? bp[j] : 0.f;
    for (int k = 0; k < Hin; ++k) acc = fmaf(hr[k], wj[k], acc);
    P[o] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  // neigh[d, j] = max over the sampled sources (first slot on ties), argmax = the winning block-local source row
  for (int64_t o = tid; o < n_dst * Hin; o += SL_THREADS) {
    const int64_t d = o / Hin;
    const int j = (int)(o - d * Hin);
    const int32_t* row = idx + d * S;
    float best = 0.f;
    int32_t arg = -1;
    if (row[0] >= 0) {
      best = -INFINITY;
      for (int s = 0; s < S; ++s) {
        const int32_t r = row[s];
        if (r < 0 || r >= n_src) continue;
        const float v = P[(int64_t)r * Hin + j];
        if (v > best) { best = v; arg = r; }
      }
      if (arg < 0) best = 0.f;
    }
    NB[o] = best;
    neigh_out[d * ldn + j] = best;
    if (argmax_out) argmax_out[o] = arg;
  }
  __syncthreads();
  // y = act(h[:n_dst] . Ws^T + neigh . Wn^T + bs + bn)
  for (int64_t o = tid; o < n_dst * Hout; o += SL_THREADS) {
    const int64_t d = o / Hout;
    const int c = (int)(o - d * Hout);
    const float* hd = h + d * ldh;
    const float* nd = NB + d * Hin;
    const float* ws = Ws + (int64_t)c * ldws;
    const float* wn = Wn + (int64_t)c * ldwn;
    float acc = (bs ? bs[c] : 0.f) + (bn ? bn[c] : 0.f);
    for (int k = 0; k < Hin; ++k) acc = fmaf(hd[k], ws[k], acc);
    for (int k = 0; k < Hin; ++k) acc = fmaf(nd[k], wn[k], acc);
    y[d * ldy + c] = relu_out ? fmaxf(acc, 0.f) : acc;
  }
}

extern "C" int ogl_small_pool_layer_fwd(const float* h, int64_t ldh, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout,
                                        int Hin, const float* Wp, int64_t ldwp, const float* bp, const float* Ws, int64_t ldws,
                                        const float* bs, const float* Wn, int64_t ldwn, const float* bn, int Hout, int relu_out,
                                        float* neigh, int64_t ldn, int32_t* argmax, float* y, int64_t ldy, ogl_stream_t stream) {
  if (!ogl_small_pool_layer_fits(n_src, n_dst, fanout, Hin, Hout)) return OGL_EINVAL;
  if (!h || !idx || !Wp || !Ws || !Wn || !neigh || !y || ldh < Hin || ldwp < Hin || ldws < Hin || ldwn < Hin || ldn < Hin || ldy < Hout)
    return OGL_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    OGL_CHECK_HIP(hipFuncSetAttribute((const void*)k_small_pool_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_small_pool_fwd, dim3(1), dim3(SL_THREADS), (size_t)sl_lds_fwd(n_src, n_dst, Hin, Hout), (hipStream_t)stream, h, ldh,
                     n_src, idx, n_dst, fanout, Hin, Wp, ldwp, bp, Ws, ldws, bs, Wn, ldwn, bn, Hout, relu_out, neigh, ldn, argmax, y, ldy);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// Backward of the same layer.  dy [n_dst, Hout] (the ReLU mask of y is applied here when relu_out), the forward's neigh / argmax.
// Outputs: dWp [Hin, Hin], dbp [Hin], dWs / dWn [Hout, Hin], dbs / dbn [Hout] (each nullable), dh [n_src, Hin] (nullable; fully
// written: zeros where nothing flows).
__global__ void __launch_bounds__(SL_THREADS) k_small_pool_bwd(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y,
                                                               int64_t ldy, int relu_out, const float* __restrict__ h, int64_t ldh,
                                                               int64_t n_src, int64_t n_dst, int Hin, int Hout,
                                                               const float* __restrict__ neigh, int64_t ldn, const int32_t* __restrict__ argmax,
                                                               const float* __restrict__ Wp, int64_t ldwp, const float* __restrict__ Ws,
                                                               int64_t ldws, const float* __restrict__ Wn, int64_t ldwn,
                                                               float* __restrict__ dWp, int64_t lddwp, float* __restrict__ dbp,
                                                               float* __restrict__ dWs, int64_t lddws, float* __restrict__ dbs,
                                                               float* __restrict__ dWn, int64_t lddwn, float* __restrict__ dbn,
                                                               float* __restrict__ dh, int64_t lddh) {
  extern __shared__ float sl_smem[];
  float* DH = sl_smem;                                 // [n_src, Hin]
  float* W = DH + n_src * Hin;                         // [Hin, Hin + 1]
  float* G = W + Hin * (Hin + 1);                      // [n_dst, Hin]  routed gradient of the winners
  float* DXS = G + n_dst * Hin;                        // [n_dst, Hin]  fc_self input gradient
  float* NB = DXS + n_dst * Hin;                       // [n_dst, Hin]  neigh (forward output)
  float* DY = NB + n_dst * Hin;                        // [n_dst, Hout] masked output gradient
  const int tid = threadIdx.x;
  for (int64_t i = tid; i < n_src * Hin; i += SL_THREADS) DH[i] = 0.f;
  for (int i = tid; i < Hin * Hin; i += SL_THREADS) W[(i / Hin) * (Hin + 1) + i % Hin] = Wp[(int64_t)(i / Hin) * ldwp + i % Hin];
  for (int64_t o = tid; o < n_dst * Hout; o += SL_THREADS) {
    const int64_t d = o / Hout;
    const int c = (int)(o - d * Hout);
    float v = dy[d * lddy + c];
    if (relu_out && !(y[d * ldy + c] > 0.f)) v = 0.f;
    DY[o] = v;
  }
  for (int64_t o = tid; o < n_dst * Hin; o += SL_THREADS) NB[o] = neigh[(o / Hin) * ldn + o % Hin];
  __syncthreads();
  // weight / bias gradients of the combine: thread -> (class c, feature k)
  for (int o = tid; o < Hout * Hin; o += SL_THREADS) {
    const int c = o / Hin, k = o - c * Hin;
    float as = 0.f, an = 0.f;
    for (int64_t d = 0; d < n_dst; ++d) {
      const float g = DY[d * Hout + c];
      as = fmaf(g, h[d * ldh + k], as);
      an = fmaf(g, NB[d * Hin + k], an);
    }
    if (dWs) dWs[(int64_t)c * lddws + k] = as;
    if (dWn) dWn[(int64_t)c * lddwn + k] = an;
  }
  for (int c = tid; c < Hout; c += SL_THREADS) {
    float a = 0.f;
    for (int64_t d = 0; d < n_dst; ++d) a += DY[d * Hout + c];
    if (dbs) dbs[c] = a;
    if (dbn) dbn[c] = a;
  }
  // input gradients of the combine: thread -> (dst d, feature k); the winners' routed gradient G = dneigh . [neigh > 0]
  for (int64_t o = tid; o < n_dst * Hin; o += SL_THREADS) {
    const int64_t d = o / Hin;
    const int k = (int)(o - d * Hin);
    float dn = 0.f, dx = 0.f;
    for (int c = 0; c < Hout; ++c) {
      const float g = DY[d * Hout + c];
      dn = fmaf(g, Wn[(int64_t)c * ldwn + k], dn);
      dx = fmaf(g, Ws[(int64_t)c * ldws + k], dx);
    }
    DXS[o] = dx;
    G[o] = (NB[o] > 0.f && argmax[o] >= 0) ? dn : 0.f;
  }
  __syncthreads();
  // fc_pool: dWp[j, k] = sum_d G[d, j] h[argmax[d, j], k], dbp[j] = sum_d G[d, j]   (thread -> (j, k): no atomics)
  for (int o = tid; o < Hin * Hin; o += SL_THREADS) {
    const int j = o / Hin, k = o - j * Hin;
    float a = 0.f;
    for (int64_t d = 0; d < n_dst; ++d) {
      const float g = G[d * Hin + j];
      if (g != 0.f) a = fmaf(g, h[(int64_t)argmax[d * Hin + j] * ldh + k], a);
    }
    if (dWp) dWp[(int64_t)j * lddwp + k] = a;
  }
  for (int j = tid; j < Hin; j += SL_THREADS) {
    float a = 0.f;
    for (int64_t d = 0; d < n_dst; ++d) a += G[d * Hin + j];
    if (dbp) dbp[j] = a;
  }
  // dh[w, :] += G[d, j] Wp[j, :] for the winner w of (d, j): LDS float atomics (the few rows that win are hit many times)
  if (dh) {
    for (int64_t o = tid; o < n_dst * Hin; o += SL_THREADS) {
      const float g = G[o];
      if (g == 0.f) continue;
      const int j = (int)(o % Hin);
      float* row = DH + (int64_t)argmax[o] * Hin;
      const float* wj = W + j * (Hin + 1);
      for (int k = 0; k < Hin; ++k) atomicAdd(&row[k], g * wj[k]);
    }
    __syncthreads();
    for (int64_t o = tid; o < n_dst * Hin; o += SL_THREADS) DH[o] += DXS[o];          // the fc_self path: first n_dst rows
    __syncthreads();
    for (int64_t o = tid; o < n_src * Hin; o += SL_THREADS) dh[(o / Hin) * lddh + o % Hin] = DH[o];
  }
}

extern "C" int ogl_small_pool_layer_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, int relu_out, const float* h,
                                        int64_t ldh, int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout, const float* neigh,
                                        int64_t ldn, const int32_t* argmax, const float* Wp, int64_t ldwp, const float* Ws,
                                        int64_t ldws, const float* Wn, int64_t ldwn, float* dWp, int64_t lddwp, float* dbp,
                                        float* dWs, int64_t lddws, float* dbs, float* dWn, int64_t lddwn, float* dbn, float* dh,
                                        int64_t lddh, ogl_stream_t stream) {
  if (!ogl_small_pool_layer_fits(n_src, n_dst, fanout, Hin, Hout)) return OGL_EINVAL;
  if (!dy || !h || !neigh || !argmax || !Wp || !Ws || !Wn || (relu_out && !y)) return OGL_EINVAL;
  if (lddy < Hout || ldh < Hin || ldn < Hin || ldwp < Hin || ldws < Hin || ldwn < Hin || (dh && lddh < Hin)) return OGL_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    OGL_CHECK_HIP(hipFuncSetAttribute((const void*)k_small_pool_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_small_pool_bwd, dim3(1), dim3(SL_THREADS), (size_t)sl_lds_bwd(n_src, n_dst, Hin, Hout), (hipStream_t)stream, dy,
                     lddy, y, ldy, relu_out, h, ldh, n_src, n_dst, Hin, Hout, neigh, ldn, argmax, Wp, ldwp, Ws, ldws, Wn, ldwn, dWp,
                     lddwp, dbp, dWs, lddws, dbs, dWn, lddwn, dbn, dh, lddh);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
