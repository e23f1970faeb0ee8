// A whole SMALL 'pool' SAGEConv layer in two launches forward and two launches backward.
//
// The live layer (DGL SAGEConv(aggregator_type='pool'), imported at R/train/graphsage/pytorch/graphsage_dgl.py:3;
// parameterisation R/inference_optimized.py:136-139,260,276):
//     P = relu(h . Wp^T + bp)                  [n_src, Hin]
//     neigh[d] = max_j P[idx[d, j]]            (0 / no winner when slot 0 is -1; first slot wins ties)
//     y = act(h[:n_dst] . Ws^T + neigh . Wn^T + bs + bn)     [n_dst, Hout]
// At the settings the reference ships for its small datasets (R/settings/pubmed.json, arxiv.json: embedding_size 32, batch 32,
// i.e. the OUTPUT layer of a batch is <= 832 source rows x 32 features -> <= 40 classes) that is 1-2 MFLOP: as the general
// launches (fc_pool GEMM, reduce, bias add, skinny GEMM; backward: 2 weight gradients, 2 input gradients, zero fill, scatter,
// fc_pool input gradient, weight gradient + split-K reduce, add) it is 15 launches of 5-16 us of pure latency each — half of a
// 32-seed train step.  Here: exact fp32 FMA arithmetic on the vector ALU (no MFMA, no LDS tiling: every operand is a few KB and
// lives in L1 / L2), one thread per output element, grid-strided over enough workgroups to spread over the chip.
// (A first version did a whole pass in ONE 1024-thread workgroup with P / dh in LDS: 54-60 us per pass — one CU issues 64 FMAs per
// cycle and every FMA needed an LDS and a global operand.)
//
// Limits (ogl_small_pool_layer_fits): Hin, Hout <= 64, n_dst * max(Hin, Hout) <= 8192, n_src <= 65536.
#include "ogl_common.h"
#include <algorithm>

#define SL_MAX_H 64

extern "C" int ogl_small_pool_layer_fits(int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout) {
  if (n_src <= 0 || n_dst <= 0 || n_dst > n_src || fanout <= 0 || Hin <= 0 || Hout <= 0) return 0;
  if (Hin > SL_MAX_H || Hout > SL_MAX_H || n_src > 65536) return 0;
  return n_dst * (int64_t)(Hin > Hout ? Hin : Hout) <= 8192;
}

// ---- forward 1: P = relu(h . Wp^T + bp), one thread per (row, feature) ---------------------------------------------------------
__global__ void __launch_bounds__(256) k_small_proj(const float* __restrict__ h, int64_t ldh, int n_src, int Hin,
                                                    const float* __restrict__ Wp, int64_t ldwp, const float* __restrict__ bp,
                                                    float* __restrict__ P) {
  __shared__ float W[SL_MAX_H * (SL_MAX_H + 1)];
  for (int i = threadIdx.x; i < Hin * Hin; i += 256) W[(i / Hin) * (Hin + 1) + i % Hin] = Wp[(int64_t)(i / Hin) * ldwp + i % Hin];
  __syncthreads();
  const int total = n_src * Hin;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    const int r = o / Hin, j = o - r * Hin;
    const float* hr = h + (int64_t)r * ldh;                // the Hin threads of a row read the same addresses: broadcast
    const float* wj = W + j * (Hin + 1);
    float acc = bp ? bp[j] : 0.f;
#pragma unroll 8
    for (int k = 0; k < Hin; ++k) acc = fmaf(hr[k], wj[k], acc);
    P[o] = fmaxf(acc, 0.f);
  }
}

// ---- forward 2: neighbour max + combine; one wave per destination (lane = feature, then lane = output column) -------------------
__global__ void __launch_bounds__(256) k_small_combine(const float* __restrict__ h, int64_t ldh, int n_src,
                                                       const int32_t* __restrict__ idx, int n_dst, int S, int Hin,
                                                       const float* __restrict__ P, const float* __restrict__ Ws, int64_t ldws,
                                                       const float* __restrict__ bs, const float* __restrict__ Wn, int64_t ldwn,
                                                       const float* __restrict__ bn, int Hout, int relu_out,
                                                       float* __restrict__ neigh_out, int64_t ldn, int32_t* __restrict__ argmax_out,
                                                       float* __restrict__ y, int64_t ldy) {
  __shared__ float NB[4][SL_MAX_H];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d = blockIdx.x * 4 + w;
  if (d < n_dst) {
    const int32_t* row = idx + (int64_t)d * S;
    float best = 0.f;
    int32_t arg = -1;
    if (lane < Hin && row[0] >= 0) {
      best = -INFINITY;
      for (int s = 0; s < S; ++s) {
        const int32_t r = row[s];                          // wave-uniform
        if (r < 0 || r >= n_src) continue;
        const float v = P[(int64_t)r * Hin + lane];
        if (v > best) { best = v; arg = r; }
      }
      if (arg < 0) best = 0.f;
    }
    if (lane < Hin) {
      NB[w][lane] = best;
      neigh_out[(int64_t)d * ldn + lane] = best;
      if (argmax_out) argmax_out[(int64_t)d * Hin + lane] = arg;
    }
  }
  __syncthreads();
  if (d < n_dst && lane < Hout) {
    const float* hd = h + (int64_t)d * ldh;
    const float* ws = Ws + (int64_t)lane * ldws;
    const float* wn = Wn + (int64_t)lane * ldwn;
    float acc = (bs ? bs[lane] : 0.f) + (bn ? bn[lane] : 0.f);
#pragma unroll 8
    for (int k = 0; k < Hin; ++k) acc = fmaf(hd[k], ws[k], acc);
#pragma unroll 8
    for (int k = 0; k < Hin; ++k) acc = fmaf(NB[w][k], wn[k], acc);
    y[(int64_t)d * ldy + lane] = relu_out ? fmaxf(acc, 0.f) : acc;
  }
}

extern "C" int ogl_small_pool_layer_fwd(const float* h, int64_t ldh, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout,
                                        int Hin, const float* Wp, int64_t ldwp, const float* bp, const float* Ws, int64_t ldws,
                                        const float* bs, const float* Wn, int64_t ldwn, const float* bn, int Hout, int relu_out,
                                        float* neigh, int64_t ldn, int32_t* argmax, float* y, int64_t ldy, float* workspace,
                                        ogl_stream_t stream) {
  if (!ogl_small_pool_layer_fits(n_src, n_dst, fanout, Hin, Hout)) return OGL_EINVAL;
  if (!h || !idx || !Wp || !Ws || !Wn || !neigh || !y || !workspace || ldh < Hin || ldwp < Hin || ldws < Hin || ldwn < Hin ||
      ldn < Hin || ldy < Hout)
    return OGL_EINVAL;
  float* P = workspace;                                    // [n_src, Hin]
  hipLaunchKernelGGL(k_small_proj, dim3((unsigned)std::min<int64_t>(ogl_cdiv(n_src * Hin, 256), 1024)), dim3(256), 0,
                     (hipStream_t)stream, h, ldh, (int)n_src, Hin, Wp, ldwp, bp, P);
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_small_combine, dim3((unsigned)ogl_cdiv(n_dst, 4)), dim3(256), 0, (hipStream_t)stream, h, ldh, (int)n_src, idx,
                     (int)n_dst, fanout, Hin, P, Ws, ldws, bs, Wn, ldwn, bn, Hout, relu_out, neigh, ldn, argmax, y, ldy);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- backward 1: everything that depends on dy only ----------------------------------------------------------------------------
// One flat, grid-strided item space:  [0, Hout*Hin) -> dWs / dWn (c, k);  then Hout -> dbs / dbn;  then n_dst*Hin -> (d, k): the input
// gradients of the combine, G = dneigh . [neigh > 0] (to the workspace) and dh[d, k] = dxself;  then (n_src - n_dst)*Hin -> dh = 0.
__device__ __forceinline__ float sl_dy(const float* dy, int64_t lddy, const float* y, int64_t ldy, int relu_out, int d, int c) {
  const float v = dy[(int64_t)d * lddy + c];
  return (relu_out && !(y[(int64_t)d * ldy + c] > 0.f)) ? 0.f : v;
}

__global__ void __launch_bounds__(256) k_small_bwd_a(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy,
                                                     int relu_out, const float* __restrict__ h, int64_t ldh, int n_src, int n_dst,
                                                     int Hin, int Hout, const float* __restrict__ neigh, int64_t ldn,
                                                     const int32_t* __restrict__ argmax, const float* __restrict__ Ws, int64_t ldws,
                                                     const float* __restrict__ Wn, int64_t ldwn, float* __restrict__ dWs,
                                                     int64_t lddws, float* __restrict__ dbs, float* __restrict__ dWn, int64_t lddwn,
                                                     float* __restrict__ dbn, float* __restrict__ G, float* __restrict__ dh,
                                                     int64_t lddh) {
  const int n_w = Hout * Hin, n_b = Hout, n_g = n_dst * Hin, n_z = dh ? (n_src - n_dst) * Hin : 0;
  const int total = n_w + n_b + n_g + n_z;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    if (o < n_w) {
      const int c = o / Hin, k = o - c * Hin;
      float as = 0.f, an = 0.f;
#pragma unroll 8
      for (int d = 0; d < n_dst; ++d) {
        const float g = sl_dy(dy, lddy, y, ldy, relu_out, d, c);
        as = fmaf(g, h[(int64_t)d * ldh + k], as);
        an = fmaf(g, neigh[(int64_t)d * ldn + k], an);
      }
      if (dWs) dWs[(int64_t)c * lddws + k] = as;
      if (dWn) dWn[(int64_t)c * lddwn + k] = an;
    } else if (o < n_w + n_b) {
      const int c = o - n_w;
      float a = 0.f;
#pragma unroll 8
      for (int d = 0; d < n_dst; ++d) a += sl_dy(dy, lddy, y, ldy, relu_out, d, c);
      if (dbs) dbs[c] = a;
      if (dbn) dbn[c] = a;
    } else if (o < n_w + n_b + n_g) {
      const int q = o - n_w - n_b, d = q / Hin, k = q - d * Hin;
      float dn = 0.f, dx = 0.f;
#pragma unroll 8
      for (int c = 0; c < Hout; ++c) {
        const float g = sl_dy(dy, lddy, y, ldy, relu_out, d, c);
        dn = fmaf(g, Wn[(int64_t)c * ldwn + k], dn);
        dx = fmaf(g, Ws[(int64_t)c * ldws + k], dx);
      }
      G[q] = (neigh[(int64_t)d * ldn + k] > 0.f && argmax[q] >= 0) ? dn : 0.f;
      if (dh) dh[(int64_t)d * lddh + k] = dx;                // the fc_self path; the winners' rows are added by k_small_bwd_b
    } else {
      const int q = o - n_w - n_b - n_g, r = n_dst + q / Hin, k = q % Hin;
      dh[(int64_t)r * lddh + k] = 0.f;
    }
  }
}

// ---- backward 2: fc_pool through the winners -------------------------------------------------------------------------------------
// items [0, Hin*Hin) -> dWp[j, k] = sum_d G[d, j] h[argmax[d, j], k];  then Hin -> dbp;  then n_dst*Hin*Hin -> (d, j, k): dh[argmax, k] += G Wp[j, k]
__global__ void __launch_bounds__(256) k_small_bwd_b(const float* __restrict__ h, int64_t ldh, int n_dst, int Hin,
                                                     const int32_t* __restrict__ argmax, const float* __restrict__ G,
                                                     const float* __restrict__ Wp, int64_t ldwp, float* __restrict__ dWp, int64_t lddwp,
                                                     float* __restrict__ dbp, float* __restrict__ dh, int64_t lddh) {
  const int n_w = Hin * Hin, n_b = Hin, n_s = dh ? n_dst * Hin * Hin : 0;
  const int total = n_w + n_b + n_s;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    if (o < n_w) {
      const int j = o / Hin, k = o - j * Hin;
      float a = 0.f;
#pragma unroll 8
      for (int d = 0; d < n_dst; ++d) {
        const float g = G[d * Hin + j];
        const int32_t w = argmax[d * Hin + j];
        a = fmaf(g, h[(int64_t)(w < 0 ? 0 : w) * ldh + k], a);            // g == 0 wherever there is no winner
      }
      if (dWp) dWp[(int64_t)j * lddwp + k] = a;
    } else if (o < n_w + n_b) {
      const int j = o - n_w;
      float a = 0.f;
#pragma unroll 8
      for (int d = 0; d < n_dst; ++d) a += G[d * Hin + j];
      if (dbp) dbp[j] = a;
    } else {
      // one atomic per thread: (winner entry q = (d, j), column k); consecutive threads hit consecutive floats of one row
      const int t = o - n_w - n_b, q = t / Hin, k = t - q * Hin;
      const float g = G[q];
      if (g == 0.f) continue;
      const int j = q % Hin;
      atomicAdd(&dh[(int64_t)argmax[q] * lddh + k], g * Wp[(int64_t)j * ldwp + k]);   // float atomics: summation order is not fixed
    }
  }
}

extern "C" int ogl_small_pool_layer_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, int relu_out, const float* h,
                                        int64_t ldh, int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout, const float* neigh,
                                        int64_t ldn, const int32_t* argmax, const float* Wp, int64_t ldwp, const float* Ws,
                                        int64_t ldws, const float* Wn, int64_t ldwn, float* dWp, int64_t lddwp, float* dbp,
                                        float* dWs, int64_t lddws, float* dbs, float* dWn, int64_t lddwn, float* dbn, float* dh,
                                        int64_t lddh, float* workspace, ogl_stream_t stream) {
  if (!ogl_small_pool_layer_fits(n_src, n_dst, fanout, Hin, Hout)) return OGL_EINVAL;
  if (!dy || !h || !neigh || !argmax || !Wp || !Ws || !Wn || !workspace || (relu_out && !y)) return OGL_EINVAL;
  if (lddy < Hout || ldh < Hin || ldn < Hin || ldwp < Hin || ldws < Hin || ldwn < Hin || (dh && lddh < Hin)) return OGL_EINVAL;
  float* G = workspace;                                    // [n_dst, Hin]
  const int64_t items_a = (int64_t)Hout * Hin + Hout + n_dst * Hin + (dh ? (n_src - n_dst) * Hin : 0);
  hipLaunchKernelGGL(k_small_bwd_a, dim3((unsigned)std::min<int64_t>(ogl_cdiv(items_a, 256), 1024)), dim3(256), 0, (hipStream_t)stream,
                     dy, lddy, y, ldy, relu_out, h, ldh, (int)n_src, (int)n_dst, Hin, Hout, neigh, ldn, argmax, Ws, ldws, Wn, ldwn, dWs,
                     lddws, dbs, dWn, lddwn, dbn, G, dh, lddh);
  OGL_CHECK_LAUNCH();
  const int64_t items_b = (int64_t)Hin * Hin + Hin + (dh ? n_dst * Hin * (int64_t)Hin : 0);
  hipLaunchKernelGGL(k_small_bwd_b, dim3((unsigned)std::min<int64_t>(ogl_cdiv(items_b, 256), 1024)), dim3(256), 0, (hipStream_t)stream,
                     h, ldh, (int)n_dst, Hin, argmax, G, Wp, ldwp, dWp, lddwp, dbp, dh, lddh);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// workspace (floats): forward n_src * Hin (the projected rows), backward n_dst * Hin (the routed gradient)
extern "C" int64_t ogl_small_pool_layer_workspace_floats(int64_t n_src, int64_t n_dst, int Hin) {
  if (n_src < 0 || n_dst < 0 || Hin < 0) return OGL_EINVAL;
  return n_src * Hin;
}
