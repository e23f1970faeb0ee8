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
#include "ogl_hip.h"
#include <algorithm>
#include <cstdlib>

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

// ---- the LAST small 'pool' layer of a train step together with nn.CrossEntropyLoss(reduction='mean') -------------------------------
// (R/train/graphsage/pytorch/model.py:87-107: logits = model(blocks, x); loss = loss_fcn(logits, labels); loss.backward()).
// At the 32-seed rungs that tail of the step was five launches at the dispatch floor and a fill — k_small_proj, k_small_combine, the
// one-workgroup cross entropy, k_small_bwd_a, k_small_bwd_b, the zero fill of the first layer's scatter target: 43 us of a 0.2 ms step.
// Here it is TWO launches, cut where the data crosses destinations:
//   forward (k_small_loss), one workgroup per destination d: the S neighbour rows of h staged in LDS, P = relu(rows . Wp^T + bp) for
//     THOSE rows only (a source picked by two destinations is projected twice: 800 row products instead of <= 832, and no [n_src, Hin]
//     intermediate), neighbour max, logits, the row's cross entropy, dlogits = (softmax - onehot) * grad_scale, and from dlogits the
//     two input gradients of the combine (G = dneigh masked by the winners, dh[d] = the fc_self path).  The combine's weights are
//     requested at the top of the block and parked in LDS over Wp / the neighbour rows once those are done with (a lane walking a
//     weight row in global memory is a chain of L2 latencies: what k_small_combine's 8 us are).  dh's rows behind the destinations
//     are zeroed by the row blocks; a caller buffer (the first layer's scatter target: megabytes at these rungs) by fill-only
//     blocks behind them.  No block waits for another: the sums over destinations are the next launch's.
//     (A first version finished them in the last block to count itself: a device-scope release per block writes back the L2 the
//     fill blocks are dirtying — the launch took 23 us, and 30 beside a 10 MB fill.)
//   backward (k_small_bwd_b2): block 0 sums what crosses destinations — dWs, dWn, dbs, dbn (rows staged through LDS in chunks of 32
//     destinations, destination order), the mean of the row losses (the order of k_ce_fwd_bwd_mean) and the optimiser's per-step
//     scalars (the passenger of ogl_ce_fwd_bwd_mean_gather); the other blocks are k_small_bwd_b (fc_pool through the winners).
// Every sum runs in the order of the kernels above, so the bits are theirs.  The VALUE of the mean loss exists after the backward
// launch (NaN until then: the contract of GraphSAGE.forward_loss(defer_mean=True)).  The root gradient of the loss is taken to be 1
// (ops.backward's unit gradient); the caller scales dlogits / G / dh for any other before the backward launch.
#define SLL_CHUNK 32
#define SLL_MAX_DST 128
#define SLL_MAX_S 64

struct SmallLossArgs {
  const float* h; int64_t ldh; int n_src; const int32_t* idx; int n_dst, S, Hin;
  const float* Wp; int64_t ldwp; const float* bp;
  const float* Ws; int64_t ldws; const float* bs;
  const float* Wn; int64_t ldwn; const float* bn; int Hout;
  const int64_t* labels; int64_t n_labels; const int64_t* label_ids; float grad_scale;
  float* neigh; int64_t ldn; int32_t* argmax; float* y; int64_t ldy; float* loss_rows; float* loss_mean; float* dl; int64_t lddl;
  float* G; float* dh; int64_t lddh; int dh_head_only;
  float4* zero_buf; int64_t zero_n4;
};

__device__ __forceinline__ float sll_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float sll_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void __launch_bounds__(256) k_small_loss(SmallLossArgs a) {
  __shared__ float W[SL_MAX_H * (SL_MAX_H + 1)];            // Wp, row stride Hin + 1;    then Ws
  __shared__ float R[SLL_MAX_S * (SL_MAX_H + 1)];           // the neighbours' rows of h; then Wn
  __shared__ float P[SLL_MAX_S * (SL_MAX_H + 1)];           // relu(fc_pool) of those rows
  __shared__ float NB[SL_MAX_H], DY[SL_MAX_H], HD[SL_MAX_H];
  __shared__ int ARG[SL_MAX_H];
  const int tid = threadIdx.x, Hin = a.Hin, Hout = a.Hout, S = a.S, ldw = Hin + 1;
  const int n_w = Hout * Hin;
  if ((int)blockIdx.x >= a.n_dst) {                          // fill-only blocks: the caller's buffer
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t nb = (int64_t)gridDim.x - a.n_dst;
    for (int64_t i = ((int64_t)blockIdx.x - a.n_dst) * 256 + tid; i < a.zero_n4; i += nb * 256) a.zero_buf[i] = z;
    return;
  }
  const int d = blockIdx.x;
  const int32_t* row = a.idx + (int64_t)d * S;
  // requested first, used last: the combine's weights (registers now, LDS once Wp and the neighbour rows are done with)
  float ws_r[16], wn_r[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int o = tid + 256 * q;
    ws_r[q] = wn_r[q] = 0.f;
    if (o < n_w) {
      const int c = o / Hin, k = o - c * Hin;
      ws_r[q] = a.Ws[(int64_t)c * a.ldws + k];
      wn_r[q] = a.Wn[(int64_t)c * a.ldwn + k];
    }
  }
  const float bias_c = (tid < Hout) ? (a.bs ? a.bs[tid] : 0.f) + (a.bn ? a.bn[tid] : 0.f) : 0.f;
  int64_t yl_pre = -1;                                       // the row's label (two dependent loads): requested here, used by wave 0 at the end
  if (tid < 64) {
    if (a.label_ids) {
      const int64_t id = a.label_ids[d];
      yl_pre = (id >= 0 && id < a.n_labels) ? a.labels[id] : -1;
    } else yl_pre = a.labels[d];
  }
  if (tid < Hin) HD[tid] = a.h[(int64_t)d * a.ldh + tid];
  for (int i = tid; i < Hin * Hin; i += 256) {
    const int j = i / Hin, k = i - j * Hin;
    W[j * ldw + k] = a.Wp[(int64_t)j * a.ldwp + k];
  }
  for (int i = tid; i < S * Hin; i += 256) {
    const int s = i / Hin, k = i - s * Hin;
    const int32_t r = row[s];
    R[s * ldw + k] = (r >= 0 && r < a.n_src) ? a.h[(int64_t)r * a.ldh + k] : 0.f;
  }
  if (a.dh && !a.dh_head_only) {                             // dh rows behind the destinations: zeros (the row blocks share them)
    const int64_t nz = (int64_t)(a.n_src - a.n_dst) * Hin;
    for (int64_t i = (int64_t)d * 256 + tid; i < nz; i += (int64_t)a.n_dst * 256)
      a.dh[(a.n_dst + i / Hin) * a.lddh + i % Hin] = 0.f;
  }
  if (gridDim.x == (unsigned)a.n_dst) {                      // (no fill-only blocks: the caller's buffer too)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)d * 256 + tid; i < a.zero_n4; i += (int64_t)a.n_dst * 256) a.zero_buf[i] = z;
  }
  __syncthreads();
  for (int o = tid; o < S * Hin; o += 256) {                 // the arithmetic of k_small_proj
    const int s = o / Hin, j = o - s * Hin;
    const float* hr = R + s * ldw;
    const float* wj = W + j * ldw;
    float acc = a.bp ? a.bp[j] : 0.f;
#pragma unroll 8
    for (int k = 0; k < Hin; ++k) acc = fmaf(hr[k], wj[k], acc);
    P[s * ldw + j] = fmaxf(acc, 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {                             // Ws over Wp, Wn over the neighbour rows (row stride Hin + 1)
    const int o = tid + 256 * q;
    if (o < n_w) {
      const int c = o / Hin, k = o - c * Hin;
      W[c * ldw + k] = ws_r[q];
      R[c * ldw + k] = wn_r[q];
    }
  }
  if (tid < Hin) {                                           // the neighbour max of k_small_combine (first slot wins ties)
    float best = 0.f;
    int32_t arg = -1;
    if (row[0] >= 0) {
      best = -INFINITY;
      for (int s = 0; s < S; ++s) {
        const int32_t r = row[s];
        if (r < 0 || r >= a.n_src) continue;
        const float v = P[s * ldw + tid];
        if (v > best) { best = v; arg = r; }
      }
      if (arg < 0) best = 0.f;
    }
    NB[tid] = best;
    ARG[tid] = arg;
    a.neigh[(int64_t)d * a.ldn + tid] = best;
    a.argmax[(int64_t)d * Hin + tid] = arg;
  }
  __syncthreads();
  if (tid < 64) {                                            // wave 0: logits of row d, then its cross entropy (k_ce_fwd_bwd_mean's)
    const int c = tid;
    float x = -INFINITY;
    if (c < Hout) {
      const float* ws = W + c * ldw;
      const float* wn = R + c * ldw;
      float acc = bias_c;
#pragma unroll 8
      for (int k = 0; k < Hin; ++k) acc = fmaf(HD[k], ws[k], acc);
#pragma unroll 8
      for (int k = 0; k < Hin; ++k) acc = fmaf(NB[k], wn[k], acc);
      a.y[(int64_t)d * a.ldy + c] = acc;
      x = acc;
    }
    const float m = sll_wave_max(x);
    const float s = sll_wave_sum(c < Hout ? expf(x - m) : 0.f);
    const float lse = m + logf(s);
    const int64_t yl = yl_pre;
    const bool ok = yl >= 0 && yl < Hout;
    const float xy = __shfl(x, ok ? (int)yl : 0);
    if (c == 0) a.loss_rows[d] = ok ? lse - xy : 0.f;
    if (c < Hout) {
      const float g = a.grad_scale * (expf(x - lse) - ((ok && c == (int)yl) ? 1.f : 0.f));
      a.dl[(int64_t)d * a.lddl + c] = g;
      DY[c] = g;
    }
  }
  __syncthreads();
  if (tid < Hin) {                                           // the (d, k) items of k_small_bwd_a
    const int k = tid;
    float dn = 0.f, dx = 0.f;
#pragma unroll 8
    for (int c = 0; c < Hout; ++c) {
      const float g = DY[c];
      dn = fmaf(g, R[c * ldw + k], dn);
      dx = fmaf(g, W[c * ldw + k], dx);
    }
    a.G[d * Hin + k] = (NB[k] > 0.f && ARG[k] >= 0) ? dn : 0.f;
    if (a.dh) a.dh[(int64_t)d * a.lddh + k] = dx;
  }
  if (d == 0 && tid == 0) *a.loss_mean = __builtin_nanf("");     // (its value is the backward launch's)
}

struct SmallBwdArgs {
  const float* h; int64_t ldh; int n_dst, Hin, Hout;
  const int32_t* argmax; const float* G; const float* Wp; int64_t ldwp;
  const float* neigh; int64_t ldn; const float* dl; int64_t lddl; const float* loss_rows;
  float* dWp; int64_t lddwp; float* dbp; float* dWs; int64_t lddws; float* dbs; float* dWn; int64_t lddwn; float* dbn;
  float* dh; int64_t lddh; float* loss_mean;
  int64_t* adam_step; float* adam_scal; double adam_lr, adam_b1, adam_b2;
};

__global__ void __launch_bounds__(256) k_small_bwd_b2(SmallBwdArgs a) {
  const int tid = threadIdx.x, Hin = a.Hin, Hout = a.Hout;
  if (blockIdx.x == 0) {
    // ---- everything that sums over destinations of the combine, in destination order ----
    __shared__ float DYc[SLL_CHUNK * SL_MAX_H], Hc[SLL_CHUNK * SL_MAX_H], Nc[SLL_CHUNK * SL_MAX_H];
    if (a.adam_step && tid == 255) {
      const int64_t t = *a.adam_step + 1;
      *a.adam_step = t;
      a.adam_scal[0] = (float)(a.adam_lr / (1.0 - pow(a.adam_b1, (double)t)));
      a.adam_scal[1] = (float)(1.0 / sqrt(1.0 - pow(a.adam_b2, (double)t)));
    }
    if (tid >= 192) {                                        // wave 3, fixed order: lane l sums rows l, l + 64, ...; then the lanes
      const int lane = tid - 192;
      float t = 0.f;
      for (int r = lane; r < a.n_dst; r += 64) t += a.loss_rows[r];
      t = sll_wave_sum(t);
      if (lane == 0) *a.loss_mean = t / (float)a.n_dst;
    }
    float as[16], an[16];                                    // outputs o = tid + 256 q of the [Hout, Hin] weight gradients
#pragma unroll
    for (int q = 0; q < 16; ++q) as[q] = an[q] = 0.f;
    float ab = 0.f;
    const int n_w = Hout * Hin;
    for (int d0 = 0; d0 < a.n_dst; d0 += SLL_CHUNK) {
      const int nd = min(SLL_CHUNK, a.n_dst - d0);
      __syncthreads();
      for (int i = tid; i < nd * Hout; i += 256) {
        const int dd = i / Hout, c = i - dd * Hout;
        DYc[i] = a.dl[(int64_t)(d0 + dd) * a.lddl + c];
      }
      for (int i = tid; i < nd * Hin; i += 256) {
        const int dd = i / Hin, k = i - dd * Hin;
        Hc[i] = a.h[(int64_t)(d0 + dd) * a.ldh + k];
        Nc[i] = a.neigh[(int64_t)(d0 + dd) * a.ldn + k];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = tid + 256 * q;
        if (o < n_w) {
          const int c = o / Hin, k = o - c * Hin;
          float s0 = as[q], s1 = an[q];
          for (int dd = 0; dd < nd; ++dd) {
            const float g = DYc[dd * Hout + c];
            s0 = fmaf(g, Hc[dd * Hin + k], s0);
            s1 = fmaf(g, Nc[dd * Hin + k], s1);
          }
          as[q] = s0; an[q] = s1;
        }
      }
      if (tid < Hout)
        for (int dd = 0; dd < nd; ++dd) ab += DYc[dd * Hout + tid];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int o = tid + 256 * q;
      if (o < n_w) {
        const int c = o / Hin, k = o - c * Hin;
        if (a.dWs) a.dWs[(int64_t)c * a.lddws + k] = as[q];
        if (a.dWn) a.dWn[(int64_t)c * a.lddwn + k] = an[q];
      }
    }
    if (tid < Hout) {
      if (a.dbs) a.dbs[tid] = ab;
      if (a.dbn) a.dbn[tid] = ab;
    }
    return;
  }
  // ---- the other blocks: k_small_bwd_b's flat item space ----
  const int n_w = Hin * Hin, n_b = Hin, n_s = a.dh ? a.n_dst * Hin * Hin : 0;
  const int total = n_w + n_b + n_s;
  const int nblk = gridDim.x - 1;
  for (int o = (blockIdx.x - 1) * 256 + tid; o < total; o += nblk * 256) {
    if (o < n_w) {
      const int j = o / Hin, k = o - j * Hin;
      float acc = 0.f;
#pragma unroll 8
      for (int d = 0; d < a.n_dst; ++d) {
        const float g = a.G[d * Hin + j];
        const int32_t w = a.argmax[d * Hin + j];
        acc = fmaf(g, a.h[(int64_t)(w < 0 ? 0 : w) * a.ldh + k], acc);
      }
      if (a.dWp) a.dWp[(int64_t)j * a.lddwp + k] = acc;
    } else if (o < n_w + n_b) {
      const int j = o - n_w;
      float acc = 0.f;
#pragma unroll 8
      for (int d = 0; d < a.n_dst; ++d) acc += a.G[d * Hin + j];
      if (a.dbp) a.dbp[j] = acc;
    } else {
      const int t = o - n_w - n_b, q = t / Hin, k = t - q * Hin;
      const float g = a.G[q];
      if (g == 0.f) continue;
      const int j = q % Hin;
      atomicAdd(&a.dh[(int64_t)a.argmax[q] * a.lddh + k], g * a.Wp[(int64_t)j * a.ldwp + k]);
    }
  }
}

extern "C" int ogl_small_pool_loss_fits(int64_t n_src, int64_t n_dst, int fanout, int Hin, int Hout) {
  return ogl_small_pool_layer_fits(n_src, n_dst, fanout, Hin, Hout) && n_dst <= SLL_MAX_DST && fanout <= SLL_MAX_S;
}

extern "C" int ogl_small_pool_layer_fwd_ce_bwd(const float* h, int64_t ldh, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout,
                                               int Hin, const float* Wp, int64_t ldwp, const float* bp, const float* Ws, int64_t ldws,
                                               const float* bs, const float* Wn, int64_t ldwn, const float* bn, int Hout,
                                               const int64_t* label_table, int64_t n_labels, const int64_t* label_ids, float grad_scale,
                                               float* neigh, int64_t ldn, int32_t* argmax, float* y, int64_t ldy, float* loss_rows,
                                               float* loss_mean, float* dlogits, int64_t lddl, float* G, float* dh, int64_t lddh,
                                               int dh_head_only, float* zero_buf, int64_t zero_floats, ogl_stream_t stream) {
  if (!ogl_small_pool_loss_fits(n_src, n_dst, fanout, Hin, Hout)) return OGL_EINVAL;
  if (!h || !idx || !Wp || !Ws || !Wn || !label_table || !neigh || !argmax || !y || !loss_rows || !loss_mean || !dlogits || !G)
    return OGL_EINVAL;
  if (ldh < Hin || ldwp < Hin || ldws < Hin || ldwn < Hin || ldn < Hin || ldy < Hout || lddl < Hout || (dh && lddh < Hin) || n_labels < 0)
    return OGL_EINVAL;
  if (zero_floats < 0 || (zero_floats > 0 && (!zero_buf || ((uintptr_t)zero_buf & 15) || (zero_floats & 3)))) return OGL_EINVAL;
  SmallLossArgs a;
  a.h = h; a.ldh = ldh; a.n_src = (int)n_src; a.idx = idx; a.n_dst = (int)n_dst; a.S = fanout; a.Hin = Hin;
  a.Wp = Wp; a.ldwp = ldwp; a.bp = bp; a.Ws = Ws; a.ldws = ldws; a.bs = bs; a.Wn = Wn; a.ldwn = ldwn; a.bn = bn; a.Hout = Hout;
  a.labels = label_table; a.n_labels = n_labels; a.label_ids = label_ids; a.grad_scale = grad_scale;
  a.neigh = neigh; a.ldn = ldn; a.argmax = argmax; a.y = y; a.ldy = ldy; a.loss_rows = loss_rows; a.loss_mean = loss_mean;
  a.dl = dlogits; a.lddl = lddl; a.G = G; a.dh = dh; a.lddh = lddh; a.dh_head_only = dh_head_only;
  a.zero_buf = (float4*)zero_buf; a.zero_n4 = zero_floats / 4;
  const int64_t extra = std::min<int64_t>(768, ogl_cdiv(zero_floats, 4096));     // fill-only blocks behind the n_dst row blocks
  hipLaunchKernelGGL(k_small_loss, dim3((unsigned)(n_dst + extra)), dim3(256), 0, (hipStream_t)stream, a);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// the rest of that layer's backward: the sums over destinations of the combine (dWs, dWn, dbs, dbn, the mean loss, the optimiser's
// per-step scalars) in block 0, fc_pool through the winners (dWp, dbp, the atomics into dh) in the others
extern "C" int ogl_small_pool_layer_bwd_pool(const float* h, int64_t ldh, int64_t n_dst, int Hin, int Hout, const int32_t* argmax,
                                             const float* G, const float* Wp, int64_t ldwp, const float* neigh, int64_t ldn,
                                             const float* dlogits, int64_t lddl, const float* loss_rows, float* dWp, int64_t lddwp,
                                             float* dbp, float* dWs, int64_t lddws, float* dbs, float* dWn, int64_t lddwn, float* dbn,
                                             float* dh, int64_t lddh, float* loss_mean, int64_t* step_dev, float* scalars_dev, double lr,
                                             double beta1, double beta2, ogl_stream_t stream) {
  if (n_dst <= 0 || n_dst > SLL_MAX_DST || Hin <= 0 || Hin > SL_MAX_H || Hout <= 0 || Hout > SL_MAX_H) return OGL_EINVAL;
  if (n_dst * (int64_t)(Hin > Hout ? Hin : Hout) > 8192) return OGL_EINVAL;
  if (!h || !argmax || !G || !Wp || !neigh || !dlogits || !loss_rows || !loss_mean) return OGL_EINVAL;
  if (ldh < Hin || ldwp < Hin || ldn < Hin || lddl < Hout || (dWp && lddwp < Hin) || (dWs && lddws < Hin) || (dWn && lddwn < Hin) ||
      (dh && lddh < Hin) || (step_dev != nullptr) != (scalars_dev != nullptr))
    return OGL_EINVAL;
  SmallBwdArgs a;
  a.h = h; a.ldh = ldh; a.n_dst = (int)n_dst; a.Hin = Hin; a.Hout = Hout; a.argmax = argmax; a.G = G; a.Wp = Wp; a.ldwp = ldwp;
  a.neigh = neigh; a.ldn = ldn; a.dl = dlogits; a.lddl = lddl; a.loss_rows = loss_rows; a.dWp = dWp; a.lddwp = lddwp; a.dbp = dbp;
  a.dWs = dWs; a.lddws = lddws; a.dbs = dbs; a.dWn = dWn; a.lddwn = lddwn; a.dbn = dbn; a.dh = dh; a.lddh = lddh; a.loss_mean = loss_mean;
  a.adam_step = step_dev; a.adam_scal = scalars_dev; a.adam_lr = lr; a.adam_b1 = beta1; a.adam_b2 = beta2;
  const int64_t items_b = (int64_t)Hin * Hin + Hin + (dh ? n_dst * Hin * (int64_t)Hin : 0);
  hipLaunchKernelGGL(k_small_bwd_b2, dim3((unsigned)(1 + std::min<int64_t>(ogl_cdiv(items_b, 256), 1024))), dim3(256), 0,
                     (hipStream_t)stream, a);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- fc_pool of the FIRST layer of a 32-seed step: P = relu(X[ids] . W^T + b) on the exact-fp32 MFMA, small tiles ----------------------
// 700-2 800 gathered rows x 128-500 features x as many outputs: 0.05-0.4 GFLOP.  On the on-the-fly split-bf16 GEMM (k_gemm, 64 x 64 tiles)
// that is 96 blocks of 32 dependent k-steps — 25 us at the pubmed-like rung, the largest launch of the step — because every step
// converts its operands to three bf16 planes before its MFMAs.  Here: v_mfma_f32_32x32x2_f32 on fp32 operands as they are (no
// conversion; one b32 LDS read per operand per 2 048 MACs), 32 x 64 block tiles (192 blocks for 768 x 500) whose FOUR waves are two
// column halves x two halves of every 64-deep k-slab — so all four SIMDs of a CU multiply although the tile is two MFMA tiles wide —,
// the k-halves summed through LDS in the epilogue, the next slab's six float4 loads per thread in flight under the current slab's
// MFMAs.  (Two vector-ALU versions came first: 16-deep slabs waited out a global-load latency per slab, 27 us; 64-deep slabs were
// LDS-bound at 3 B of fragment reads per FMA, 25 us.)  Exact fp32 products, fp32 accumulate; k ascending inside a half.
#define SPR_BM 32
#define SPR_BN 64
#define SPR_BK 64
typedef float spr_f32x16 __attribute__((ext_vector_type(16)));
// (A form with the block's 32 gathered A rows RESIDENT in LDS for the whole product — all their loads requested up front, only the
// weight slabs streaming — was measured slower in the step, round 5: 0.1027 against 0.0958-0.0985 ms per pubmed-like step; removed.)
__global__ void __launch_bounds__(256) k_small_proj_rows(const float* __restrict__ X, int64_t ldx, const int64_t* __restrict__ ids,
                                                         int64_t n_table, int M, int K, const float* __restrict__ W, int64_t ldw, int N,
                                                         const float* __restrict__ bias, int relu, float* __restrict__ Y, int64_t ldy,
                                                         const int64_t* __restrict__ m_live) {
  // (m_live: a device scalar — the rows at or behind it are padding of an upper-bound launch: a captured 32-seed step sized for the
  // largest input block it can ever see needs no size read-back before its train graph.  The grid's y extent is capped and a block
  // walks the row tiles y, y + gridDim.y, ... below the live count: an upper bound of 21 632 rows would otherwise dispatch 5 408
  // workgroups, 52 KB of LDS each, to do the work of 700 — measured +10 us per step)
  const int64_t m_eff = m_live ? min((int64_t)M, max((int64_t)0, *m_live)) : (int64_t)M;
  extern __shared__ __attribute__((aligned(16))) float spr_smem[];
  typedef float (*ATile)[SPR_BM + 2];
  typedef float (*BTile)[SPR_BK][SPR_BN + 4];
  const int nk = (K + SPR_BK - 1) / SPR_BK;
  // layout: A [2 * 64][34] (two slabs) then B [2][64][68]
  ATile As = (ATile)spr_smem;
  BTile Bs = (BTile)(spr_smem + (size_t)2 * SPR_BK * (SPR_BM + 2));
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int wx = wv & 1, kh = wv >> 1;                       // column half, k half
  const int n0 = blockIdx.x * SPR_BN;
  for (int m0 = blockIdx.y * SPR_BM; m0 < m_eff; m0 += gridDim.y * SPR_BM) {
  // loaders: A row (t >> 3), float4s at k = 4 (t & 7) + {0, 32};  B row (t >> 2), float4s at k = 4 (t & 3) + {0, 16, 32, 48}
  const int ar = t >> 3, ak = (t & 7) * 4, br = t >> 2, bk = (t & 3) * 4;
  const float* arow = nullptr;
  {
    const int m = m0 + ar;
    if (m < M) {
      const int64_t id = ids ? ids[m] : (int64_t)m;
      if (id >= 0 && id < n_table) arow = X + id * ldx;
    }
  }
  const float* brow = (n0 + br < N) ? W + (int64_t)(n0 + br) * ldw : nullptr;
  float4 pa[2][2], pb[2][4];
  auto fetch = [&](int k0, float4* qa, float4* qb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + ak + 32 * u;                        // (K % 4 == 0: a float4 is inside the row or outside it)
      qa[u] = (arow && k < K) ? *(const float4*)(arow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + bk + 16 * u;
      qb[u] = (brow && k < K) ? *(const float4*)(brow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto park = [&](int buf, const float4* qa, const float4* qb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = buf * SPR_BK + ak + 32 * u;
      As[k + 0][ar] = qa[u].x; As[k + 1][ar] = qa[u].y; As[k + 2][ar] = qa[u].z; As[k + 3][ar] = qa[u].w;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = bk + 16 * u;
      Bs[buf][k + 0][br] = qb[u].x; Bs[buf][k + 1][br] = qb[u].y; Bs[buf][k + 2][br] = qb[u].z; Bs[buf][k + 3][br] = qb[u].w;
    }
  };
  spr_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int fr = lane & 31, fk = lane >> 5;                  // MFMA operand lane map: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]
  fetch(0, pa[0], pb[0]);
  if (nk > 1) fetch(SPR_BK, pa[1], pb[1]);
  park(0, pa[0], pb[0]);
  if (nk > 2) fetch(2 * SPR_BK, pa[0], pb[0]);
  __syncthreads();
  auto slab = [&](int kt, int buf) {
    const int abase = buf * SPR_BK;
    float fa[16], fb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {                           // this wave's half of the slab: k = 32 kh + 2 q + fk
      fa[q] = As[abase + kh * 32 + 2 * q + fk][fr];
      fb[q] = Bs[buf][kh * 32 + 2 * q + fk][wx * 32 + fr];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q], fb[q], acc, 0, 0, 0);
  };
  // slab kt lives in LDS buffer kt & 1; slab kt + 1 in register set (kt + 1) & 1, slab kt + 2 in set kt & 1 (two trips per turn: the
  // register sets alternate at compile time)
  for (int kt = 0; kt < nk; kt += 2) {
    slab(kt, 0);
    if (kt + 1 < nk) park(1, pa[1], pb[1]);
    if (kt + 3 < nk) fetch((kt + 3) * SPR_BK, pa[1], pb[1]);
    __syncthreads();
    if (kt + 1 >= nk) break;
    slab(kt + 1, 1);
    if (kt + 2 < nk) park(0, pa[0], pb[0]);
    if (kt + 4 < nk) fetch((kt + 4) * SPR_BK, pa[0], pb[0]);
    __syncthreads();
  }
  // epilogue: the upper k-halves through LDS (the operand buffers are done with), the lower ones add, finish and store
  float* red = (float*)&Bs[0][0][0];                         // [2 column halves][16 regs][64 lanes] floats = 8 KB of the 34 KB B buffers
  if (kh == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wx * 16 + i) * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (kh == 0) {
    const int n = n0 + wx * 32 + fr;
    const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * fk;    // C/D lane map: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)
      if (m < M && n < N) {
        const float v = (acc[i] + red[(wx * 16 + i) * 64 + lane]) + bv;
        Y[(int64_t)m * ldy + n] = relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
  __syncthreads();                                           // (the next row tile parks its first slabs where `red` lives)
  }
}

extern "C" int ogl_small_proj_rows(const float* x, int64_t ldx, const int64_t* x_rows, int64_t n_table, int64_t M, int K, const float* w,
                                   int64_t ldw, int N, const float* bias, int relu, float* y, int64_t ldy, const int64_t* m_live_dev,
                                   ogl_stream_t stream) {
  if (M < 0 || M > 65536 || K <= 0 || (K & 3) || N <= 0 || N > 4096 || n_table <= 0) return OGL_EINVAL;
  if (M == 0) return OGL_OK;
  if (!x || !w || !y || ldx < K || ldw < K || ldy < N || (ldx & 3) || (ldw & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return OGL_EINVAL;
  const size_t b_bytes = sizeof(float) * 2 * SPR_BK * (SPR_BN + 4);
  const size_t strm_bytes = sizeof(float) * 2 * SPR_BK * (SPR_BM + 2) + b_bytes;
  const dim3 grid((unsigned)ogl_cdiv(N, SPR_BN), (unsigned)std::min<int64_t>(ogl_cdiv(M, SPR_BM), m_live_dev ? 96 : 65535));
  hipLaunchKernelGGL(k_small_proj_rows, grid, dim3(256), strm_bytes, (hipStream_t)stream, x, ldx, x_rows, n_table, (int)M, K, w, ldw, N, bias,
                     relu, y, ldy, m_live_dev);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- the FIRST 'pool' layer of a 32-seed step behind its fc_pool product: neighbour max + combine in one launch, and the head of its
// backward in one launch --------------------------------------------------------------------------------------------------------------
// The live layer's first layer (R/train/graphsage/pytorch/graphsage_dgl.py:26-31 -> DGL SAGEConv 'pool'; the reference's small settings:
// in_feats 500 / 128, embedding_size 32, <= 832 destinations): P = relu(fc_pool(X[src])) is a real product and stays on the GEMM kernel;
// everything behind it is latency.  Before: the max aggregator (9 us) + a skinny dual-input product (14 us) forward; the ReLU mask
// (5 us) + the [n1, 32] x [32, F] input-gradient product (8 us) + the winners' scatter (11 us) backward.
//   forward, one workgroup per destination d (thread t = float4 column t of a row, F <= 1024):
//     neigh[d] = max_j P[idx[d, j]] with its argmax (the order of k_reduce_fwd_v4: first valid slot, then strictly greater), then
//     y[d, c] = act(bs[c] + bn[c] + sum_k X[ids[d], k] Ws[c, k] + neigh[d, k] Wn[c, k]): wave w forms output columns 8w .. 8w + 7, a
//     lane its float4 columns' share of the 8 sums, a halving butterfly (7 + 3 shuffles instead of 8 x 6) joins them;
//   backward, one workgroup per destination: dy = dout . [y > 0];  dneigh[k] = sum_c dy[c] Wn[c, k] (thread = float4 column);
//     then EITHER the dense dneigh row (the planned image path of the layer-0 weight gradient consumes it) OR the winners' scatter
//     dP[argmax[d, k], k] += dneigh[k] . [neigh[d, k] > 0] with float atomics into a zeroed [n_src, F] matrix (k_reduce_bwd_max's).
// Sums over k run lane-parallel and are joined by the butterfly: fp32 rounding differs from the GEMM kernels' order (tolerances of
// tests/test_gpu_small_layer.py), the max / argmax are exact.
#define SFL_H 32

template <int NCH>
__global__ void __launch_bounds__(256) k_small_first_fwd(const float* __restrict__ P, int64_t ldp, int n_src, const int32_t* __restrict__ idx,
                                                         int n_dst, int S, int F, const float* __restrict__ table, int64_t ldt,
                                                         const int64_t* __restrict__ ids, int64_t n_table, const float* __restrict__ Ws,
                                                         int64_t ldws, const float* __restrict__ bs, const float* __restrict__ Wn,
                                                         int64_t ldwn, const float* __restrict__ bn, int H, int relu_out,
                                                         float* __restrict__ neigh, int64_t ldn, int32_t* __restrict__ argmax,
                                                         float* __restrict__ y, int64_t ldy, const int64_t* __restrict__ n_live) {
  // one WORKGROUP per destination (a first version gave it one wave: 128 weight loads in a row per wave, 42 us for 101 destinations):
  // phase 1 — thread t owns float4 column t of the row: the S neighbour rows of P, all loads of four rows in flight, max + argmax;
  // phase 2 — wave w owns output columns 8w .. 8w + 7: its lanes cover the float4 columns, 32 weight loads in flight per lane,
  // the lane sums joined by a halving butterfly.
  __shared__ float4 NB4[256], XS4[256];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int d = blockIdx.x;
  const int f4 = (F + 3) >> 2;
  if (n_live && d >= *n_live) {
    // a PADDED destination row of a captured step's upper-bound block (no neighbours, no row of its own): what the full path would
    // write — zeros, no winners, act(bias) — without its loads (832 rows of which 100-230 are live at the 32-seed rungs)
    for (int k = t; k < F; k += 256) {
      neigh[(int64_t)d * ldn + k] = 0.f;
      if (argmax) argmax[(int64_t)d * F + k] = -1;
    }
    if (t < H) {
      const float o = (bs ? bs[t] : 0.f) + (bn ? bn[t] : 0.f);
      y[(int64_t)d * ldy + t] = relu_out ? fmaxf(o, 0.f) : o;
    }
    return;
  }
  const int col = min(t, f4 - 1);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int arg[4] = {-1, -1, -1, -1};
  bool any = false;
  const int32_t mine = lane < S ? idx[(int64_t)d * S + lane] : -1;         // S <= 64 (every wave holds the row)
  // phase 2's weights do not depend on phase 1: requested HERE (NCH <= 2: 16 NCH float4 per lane), they arrive under the index ->
  // row -> max chain instead of behind it
  constexpr bool EARLY_W = NCH <= 2;
  float4 ew1[EARLY_W ? 8 : 1][EARLY_W ? NCH : 1], ew2[EARLY_W ? 8 : 1][EARLY_W ? NCH : 1];
  if constexpr (EARLY_W) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = min(wv * 8 + i, H - 1);
#pragma unroll
      for (int q = 0; q < NCH; ++q) {
        const int ch = min(q * 64 + lane, f4 - 1);
        ew1[i][q] = *(const float4*)(Ws + (int64_t)c * ldws + ch * 4);
        ew2[i][q] = *(const float4*)(Wn + (int64_t)c * ldwn + ch * 4);
      }
    }
  }
  for (int j0 = 0; j0 < S; j0 += 4) {
    float4 v[4];
    int r[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
      const int q = __shfl(mine, j < S ? j : S - 1);
      ok[u] = j < S && q >= 0 && q < n_src;
      r[u] = ok[u] ? q : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ((const float4*)(P + (int64_t)r[u] * ldp))[col];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;                                             // block-uniform
      if (!any) {
        acc = v[u];
        arg[0] = arg[1] = arg[2] = arg[3] = r[u];
      } else {
        if (v[u].x > acc.x) { acc.x = v[u].x; arg[0] = r[u]; }
        if (v[u].y > acc.y) { acc.y = v[u].y; arg[1] = r[u]; }
        if (v[u].z > acc.z) { acc.z = v[u].z; arg[2] = r[u]; }
        if (v[u].w > acc.w) { acc.w = v[u].w; arg[3] = r[u]; }
      }
      any = true;
    }
  }
  // the destination's own row (X[ids[d]]; an id outside the table: zeros); neigh / argmax out; both rows to LDS, zero past F
  const int64_t id = ids ? ids[d] : (int64_t)d;
  const bool have = id >= 0 && id < n_table;
  if (t < f4) {
    float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) xs = ((const float4*)(table + id * ldt))[t];
    float av[4] = {acc.x, acc.y, acc.z, acc.w};
    float xv[4] = {xs.x, xs.y, xs.z, xs.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = t * 4 + e;
      if (k < F) {
        neigh[(int64_t)d * ldn + k] = av[e];
        if (argmax) argmax[(int64_t)d * F + k] = arg[e];
      } else {
        av[e] = 0.f; xv[e] = 0.f;
      }
    }
    NB4[t] = make_float4(av[0], av[1], av[2], av[3]);
    XS4[t] = make_float4(xv[0], xv[1], xv[2], xv[3]);
  }
  __syncthreads();
  float part[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = wv * 8 + i;
    float a = 0.f;
    if (c < H) {
#pragma unroll
      for (int q = 0; q < NCH; ++q) {
        const int ch = q * 64 + lane;
        if (ch < f4) {
          // (rows of Ws / Wn are 16-byte aligned; the last float4 of a row may reach past F: its x / neigh factors are zero)
          float4 w1, w2;
          if constexpr (EARLY_W) { w1 = ew1[i][q]; w2 = ew2[i][q]; }
          else {
            w1 = *(const float4*)(Ws + (int64_t)c * ldws + ch * 4);
            w2 = *(const float4*)(Wn + (int64_t)c * ldwn + ch * 4);
          }
          const float4 xs = XS4[ch], nb = NB4[ch];
          a = fmaf(xs.x, w1.x, a); a = fmaf(xs.y, w1.y, a); a = fmaf(xs.z, w1.z, a); a = fmaf(xs.w, w1.w, a);
          a = fmaf(nb.x, w2.x, a); a = fmaf(nb.y, w2.y, a); a = fmaf(nb.z, w2.z, a); a = fmaf(nb.w, w2.w, a);
        }
      }
    }
    part[i] = a;
  }
  // halving butterfly: 8 values over 64 lanes -> value i in the lanes with (lane >> 3) == i, then the 8 lanes of a group summed
#pragma unroll
  for (int n = 4, off = 32; n >= 1; n >>= 1, off >>= 1) {
    const bool hi = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const float keep = hi ? part[i + n] : part[i];
      const float send = hi ? part[i] : part[i + n];
      part[i] = keep + __shfl_xor(send, off);
    }
  }
  part[0] += __shfl_xor(part[0], 4);
  part[0] += __shfl_xor(part[0], 2);
  part[0] += __shfl_xor(part[0], 1);
  const int c = wv * 8 + (lane >> 3);
  if ((lane & 7) == 0 && c < H) {
    const float o = part[0] + ((bs ? bs[c] : 0.f) + (bn ? bn[c] : 0.f));
    y[(int64_t)d * ldy + c] = relu_out ? fmaxf(o, 0.f) : o;
  }
}

// The gradient of the layer's OUTPUT routed in from the small last layer that consumed it (ogl_small_pool_layer_fwd_ce_bwd's G / head
// rows / argmax): dout[d, k] = head[d, k] (d < n_head) + sum over the records q = (d1, j) with arg[q] == d of G[q] W[j, k] — the
// winners' scatter of that layer's fc_pool as a GATHER by the consumer: no atomics (record order), no zeroed [n_src, H] matrix, and
// that layer's backward launch (k_small_bwd_b2) disappears from the step.  n <= SFB_ROUTE_MAX records.
#define SFB_ROUTE_MAX 2048
struct SfbRoute { const int32_t* arg; const float* G; const float* W; int64_t ldw; const float* head; int64_t ldh; int n_head, Hr, n; };

__global__ void __launch_bounds__(256) k_small_first_bwd(const float* __restrict__ dout, int64_t lddo, const float* __restrict__ y, int64_t ldy,
                                                         int relu_out, int n_dst, int H, int F, const float* __restrict__ Wn, int64_t ldwn,
                                                         const float* __restrict__ neigh, int64_t ldn, const int32_t* __restrict__ argmax,
                                                         float* __restrict__ dy, int64_t lddy, float* __restrict__ dneigh, int64_t lddn,
                                                         float* __restrict__ dP, int64_t lddp, int n_src, int mask_dneigh, SfbRoute rt,
                                                         const int64_t* __restrict__ n_live) {
  // one workgroup per destination, thread t = float4 column t of the row (F <= 1024): H independent weight loads per thread
  __shared__ float G[SFL_H];
  __shared__ float MG[SFB_ROUTE_MAX];
  __shared__ short MJ[SFB_ROUTE_MAX];
  __shared__ int cnt_w[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int d = blockIdx.x;
  const int f4 = (F + 3) >> 2;
  if (n_live && rt.arg && d >= *n_live && d >= rt.n_head) {
    // a padded row: nobody's winner, no head row — its gradient is zero; written without scanning the route
    if (t < H) dy[(int64_t)d * lddy + t] = 0.f;
    if (dneigh)
      for (int k = t; k < F; k += 256) dneigh[(int64_t)d * lddn + k] = 0.f;
    return;
  }
  int nm = 0;
  if (rt.arg) {                                              // this destination's records, compacted in record order
    for (int q0 = 0; q0 < rt.n; q0 += 256) {
      const int q = q0 + t;
      float g = 0.f;
      bool flag = false;
      if (q < rt.n && rt.arg[q] == d) { g = rt.G[q]; flag = g != 0.f; }
      const unsigned long long bal = __ballot(flag);
      const int pre = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) cnt_w[wv] = __popcll(bal);
      __syncthreads();
      int off = nm;
      for (int w = 0; w < wv; ++w) off += cnt_w[w];
      if (flag) { MG[off + pre] = g; MJ[off + pre] = (short)(q % rt.Hr); }
      nm += cnt_w[0] + cnt_w[1] + cnt_w[2] + cnt_w[3];
      __syncthreads();
    }
  }
  if (t < H) {
    float g;
    if (rt.arg) {
      g = d < rt.n_head ? rt.head[(int64_t)d * rt.ldh + t] : 0.f;
      for (int m = 0; m < nm; ++m) g = fmaf(MG[m], rt.W[(int64_t)MJ[m] * rt.ldw + t], g);
    } else {
      g = dout[(int64_t)d * lddo + t];
    }
    if (relu_out && !(y[(int64_t)d * ldy + t] > 0.f)) g = 0.f;
    dy[(int64_t)d * lddy + t] = g;
    G[t] = g;
  }
  __syncthreads();
  if (t >= f4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int c = 0; c < H; ++c) {                                           // (c ascending: the order of the input-gradient product's k loop)
    const float gc = G[c];
    const float4 w = *(const float4*)(Wn + (int64_t)c * ldwn + t * 4);
    acc.x = fmaf(gc, w.x, acc.x); acc.y = fmaf(gc, w.y, acc.y);
    acc.z = fmaf(gc, w.z, acc.z); acc.w = fmaf(gc, w.w, acc.w);
  }
  const float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = t * 4 + e;
    if (k >= F) continue;
    bool win = true;
    int32_t w = 0;
    if (dP || mask_dneigh) {
      w = argmax[(int64_t)d * F + k];
      win = w >= 0 && w < n_src && neigh[(int64_t)d * ldn + k] > 0.f;
    }
    if (dneigh) dneigh[(int64_t)d * lddn + k] = (mask_dneigh && !win) ? 0.f : av[e];
    if (dP && win) atomicAdd(&dP[(int64_t)w * lddp + k], av[e]);
  }
}

extern "C" int ogl_small_first_layer_fits(int64_t n_src, int64_t n_dst, int fanout, int F, int H) {
  return n_src > 0 && n_dst > 0 && n_dst <= n_src && n_dst <= 8192 && fanout > 0 && fanout <= 64 && F >= 16 && F <= 1024 && H > 0 &&
         H <= SFL_H && n_src < (1 << 30);
}

extern "C" int ogl_small_first_layer_fwd(const float* P, int64_t ldp, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout, int F,
                                         const float* table, int64_t ldt, const int64_t* ids, int64_t n_table, const float* Ws,
                                         int64_t ldws, const float* bs, const float* Wn, int64_t ldwn, const float* bn, int H,
                                         int relu_out, float* neigh, int64_t ldn, int32_t* argmax, float* y, int64_t ldy,
                                         const int64_t* n_live_dev, ogl_stream_t stream) {
  if (!ogl_small_first_layer_fits(n_src, n_dst, fanout, F, H)) return OGL_EINVAL;
  if (!P || !idx || !table || !Ws || !Wn || !neigh || !y || n_table <= 0) return OGL_EINVAL;
  const int64_t f4x4 = ((int64_t)F + 3) / 4 * 4;
  // float4 access: 16-byte rows everywhere a row is read or written as float4 (P, the table, the two weights)
  if (ldp < f4x4 || ldt < f4x4 || (ldp & 3) || (ldt & 3) || (ldws & 3) || (ldwn & 3) || ldws < F || ldwn < F || ldn < F || ldy < H)
    return OGL_EINVAL;
  if (((uintptr_t)P & 15) || ((uintptr_t)table & 15) || ((uintptr_t)Ws & 15) || ((uintptr_t)Wn & 15)) return OGL_EINVAL;
  const dim3 grid((unsigned)n_dst), block(256);
  const int nch = (int)ogl_cdiv(ogl_cdiv(F, 4), 64);
#define SFL_FWD(N)                                                                                                                      \
  hipLaunchKernelGGL(k_small_first_fwd<N>, grid, block, 0, (hipStream_t)stream, P, ldp, (int)n_src, idx, (int)n_dst, fanout, F, table, \
                     ldt, ids, n_table, Ws, ldws, bs, Wn, ldwn, bn, H, relu_out, neigh, ldn, argmax, y, ldy, n_live_dev)
  if (nch == 1) SFL_FWD(1); else if (nch == 2) SFL_FWD(2); else if (nch == 3) SFL_FWD(3); else SFL_FWD(4);
#undef SFL_FWD
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// dy [n_dst, H] always; dneigh [n_dst, F] (nullable; mask_dneigh: zero where no winner takes it) and / or dP [n_src, F] (nullable, ZEROED
// by the caller: float atomics add into it)
extern "C" int ogl_small_first_layer_bwd(const float* dout, int64_t lddo, const float* y, int64_t ldy, int relu_out, int64_t n_dst, int H,
                                         int F, const float* Wn, int64_t ldwn, const float* neigh, int64_t ldn, const int32_t* argmax,
                                         float* dy, int64_t lddy, float* dneigh, int64_t lddn, float* dP, int64_t lddp, int64_t n_src,
                                         int mask_dneigh, const int32_t* route_arg, const float* route_G, int64_t route_n, int route_H,
                                         const float* route_W, int64_t route_ldw, const float* route_head, int64_t route_ldh,
                                         int64_t route_n_head, const int64_t* n_live_dev, ogl_stream_t stream) {
  if (n_dst <= 0 || n_dst > 8192 || H <= 0 || H > SFL_H || F < 16 || F > 1024 || n_src <= 0 || n_src >= (1 << 30)) return OGL_EINVAL;
  if ((!dout && !route_arg) || !Wn || !dy || (relu_out && !y) || (!dneigh && !dP) || ((dP || mask_dneigh) && (!argmax || !neigh)))
    return OGL_EINVAL;
  if (route_arg && (!route_G || !route_W || !route_head || route_n <= 0 || route_n > SFB_ROUTE_MAX || route_H <= 0 || route_H > 64 ||
                    route_ldw < H || route_ldh < H || route_n_head < 0 || route_n % route_H))
    return OGL_EINVAL;
  if ((!route_arg && lddo < H) || (relu_out && ldy < H) || lddy < H || ldwn < F || (ldwn & 3) || ((uintptr_t)Wn & 15) || (dneigh && lddn < F) ||
      (dP && lddp < F) || ((dP || mask_dneigh) && ldn < F))
    return OGL_EINVAL;
  SfbRoute rt;
  rt.arg = route_arg; rt.G = route_G; rt.W = route_W; rt.ldw = route_ldw; rt.head = route_head; rt.ldh = route_ldh;
  rt.n_head = (int)route_n_head; rt.Hr = route_H > 0 ? route_H : 1; rt.n = (int)route_n;
  hipLaunchKernelGGL(k_small_first_bwd, dim3((unsigned)n_dst), dim3(256), 0, (hipStream_t)stream, dout, lddo, y, ldy, relu_out, (int)n_dst, H, F,
                     Wn, ldwn, neigh, ldn, argmax, dy, lddy, dneigh, lddn, dP, lddp, (int)n_src, mask_dneigh, rt, n_live_dev);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// The three weight gradients of that layer from RECORDS, one launch.  Each is "output row r = sum_d g(d, r) . row(d, r)":
//   dWp[j, :] = sum_d G[d, j]  X[ids[argmax[d, j]], :]   (G = the masked dneigh of ogl_small_first_layer_bwd: the winners' records —
//               n_dst F records of F MACs each instead of the dense [F, n_src] x [n_src, F] product, its zeroed scatter target and its
//               transposes / plan + image: at the 32-seed rungs, 101-228 destinations, 4-25 M MACs against 170-420 M)
//   dWs[c, :] = sum_d dy[d, c] X[ids[d], :]              dWn[c, :] = sum_d dy[d, c] neigh[d, :]      (+ the three bias gradients)
// One workgroup per output row: its column of weights compacted in destination order into LDS (the ReLU leaves about half, padded
// destinations nothing), then thread t = float4 column t of the gathered rows, eight row loads in flight.  Sums in destination order:
// reproducible, no atomics.  (The record idea fed to the MFMA tiles lost at the Reddit shape — 4.2 M records re-read per column tile,
// DESIGN section 8; here the rows come from L2 and there is nothing to tile.)
#define SFD_MAX_DST 2048
#define SFD_MAX_SEG 6
struct SfdArgs {
  ogl_rec_seg_t seg[SFD_MAX_SEG];
  int first[SFD_MAX_SEG + 1];                  // segment i owns blocks [first[i], first[i + 1]); one more block = the tail
  int nseg;
  const float* loss_rows; int n_loss; float* loss_mean;       // tail block (optional): the mean of the row losses ...
  int64_t* adam_step; float* adam_scal; double adam_lr, adam_b1, adam_b2;   // ... and the optimiser's per-step scalars
};

__global__ void __launch_bounds__(256) k_small_first_dw(SfdArgs a) {
  __shared__ float GS[SFD_MAX_DST + 8];
  __shared__ int64_t RS[SFD_MAX_DST + 8];
  __shared__ int cnt_w[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if ((int)blockIdx.x >= a.first[a.nseg]) {                  // the tail block
    if (a.loss_mean && wv == 0) {                            // fixed order: lane l sums rows l, l + 64, ...; then the lanes (k_ce_fwd_bwd_mean's)
      float s = 0.f;
      for (int r = lane; r < a.n_loss; r += 64) s += a.loss_rows[r];
      s = sll_wave_sum(s);
      if (lane == 0) *a.loss_mean = s / (float)a.n_loss;
    }
    if (a.adam_step && t == 255) {
      const int64_t st = *a.adam_step + 1;
      *a.adam_step = st;
      a.adam_scal[0] = (float)(a.adam_lr / (1.0 - pow(a.adam_b1, (double)st)));
      a.adam_scal[1] = (float)(1.0 / sqrt(1.0 - pow(a.adam_b2, (double)st)));
    }
    return;
  }
  int si = 0;
#pragma unroll
  for (int q = 1; q < SFD_MAX_SEG; ++q)
    if (q < a.nseg && (int)blockIdx.x >= a.first[q]) si = q;
  const ogl_rec_seg_t& sg = a.seg[si];
  const int j = blockIdx.x - a.first[si];
  const int F = sg.F;
  // (n_live: the live destination rows of a captured step's upper-bound block — rows behind them have zero weights anyway)
  const int n_dst = sg.n_live ? (int)min((int64_t)sg.n_dst, max((int64_t)0, *sg.n_live)) : (int)sg.n_dst;
  const int f4 = (F + 3) >> 2;
  int base = 0;                                              // (block-uniform: every thread keeps the same count)
  for (int d0 = 0; d0 < n_dst; d0 += 256) {
    const int d = d0 + t;
    float g = 0.f;
    int64_t row = -1;
    if (d < n_dst) {
      g = sg.G[(int64_t)d * sg.ldg + j];
      if (g != 0.f) {
        int64_t w = d;
        if (sg.arg) {
          const int32_t q = sg.arg[(int64_t)d * sg.ldarg + j];
          w = (q >= 0 && q < sg.n_idx) ? q : -1;
        }
        if (w >= 0) {
          const int64_t id = sg.ids ? sg.ids[w] : w;
          if (id >= 0 && id < sg.n_rows) row = id;
        }
      }
    }
    const bool flag = row >= 0;
    const unsigned long long bal = __ballot(flag);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) cnt_w[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int q = 0; q < wv; ++q) off += cnt_w[q];
    if (flag) { GS[off + pre] = g; RS[off + pre] = row; }
    base += cnt_w[0] + cnt_w[1] + cnt_w[2] + cnt_w[3];
    __syncthreads();
  }
  const int n = base;
  if (t < 8) { GS[n + t] = 0.f; RS[n + t] = 0; }             // padding of the last group of eight: zero weight, row 0
  __syncthreads();
  if (t < f4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < n; i += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = ((const float4*)(sg.rows + RS[i + u] * sg.ldr))[t];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float g = GS[i + u];
        acc.x = fmaf(g, x[u].x, acc.x); acc.y = fmaf(g, x[u].y, acc.y);
        acc.z = fmaf(g, x[u].z, acc.z); acc.w = fmaf(g, x[u].w, acc.w);
      }
    }
    const float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t * 4 + e < F) sg.dW[(int64_t)j * sg.lddw + t * 4 + e] = av[e];
  }
  if ((sg.db || sg.db2) && wv == 3) {                        // fixed order: lane l sums entries l, l + 64, ...; then the lanes
    float b = 0.f;
    for (int i = lane; i < n; i += 64) b += GS[i];
    b = sll_wave_sum(b);
    if (lane == 0) {
      if (sg.db) sg.db[j] = b;
      if (sg.db2) sg.db2[j] = b;
    }
  }
}

// The general form: up to six row groups ("segments", include/ogl_hip.h: ogl_rec_seg_t) in one launch, + an optional tail block that
// finishes a deferred mean loss and advances the optimiser's step count — what lets a 32-seed step's LAST layer hand its whole
// backward (weight gradients from its own records, the mean, Adam's scalars) to the first layer's launch.
extern "C" int ogl_record_weight_grads(const ogl_rec_seg_t* segs, int nseg, const float* loss_rows, int64_t n_loss, float* loss_mean,
                                       int64_t* step_dev, float* scalars_dev, double lr, double beta1, double beta2, ogl_stream_t stream) {
  if (!segs || nseg <= 0 || nseg > SFD_MAX_SEG) return OGL_EINVAL;
  if ((loss_mean && (!loss_rows || n_loss <= 0)) || (step_dev != nullptr) != (scalars_dev != nullptr)) return OGL_EINVAL;
  SfdArgs a;
  a.nseg = nseg;
  int blocks = 0;
  for (int i = 0; i < nseg; ++i) {
    const ogl_rec_seg_t& s = segs[i];
    if (!s.G || !s.rows || !s.dW || s.n_out <= 0 || s.n_dst <= 0 || s.n_dst > SFD_MAX_DST || s.F < 4 || s.F > 1024 || s.n_rows <= 0)
      return OGL_EINVAL;
    const int64_t f4x4 = ((int64_t)s.F + 3) / 4 * 4;
    if (s.ldg < s.n_out || s.lddw < s.F || s.ldr < f4x4 || (s.ldr & 3) || ((uintptr_t)s.rows & 15) || (s.arg && (s.ldarg < s.n_out || s.n_idx <= 0)))
      return OGL_EINVAL;
    a.seg[i] = s;
    a.first[i] = blocks;
    blocks += s.n_out;
  }
  a.first[nseg] = blocks;
  for (int i = nseg + 1; i <= SFD_MAX_SEG; ++i) a.first[i] = blocks;
  a.loss_rows = loss_rows; a.n_loss = (int)n_loss; a.loss_mean = loss_mean;
  a.adam_step = step_dev; a.adam_scal = scalars_dev; a.adam_lr = lr; a.adam_b1 = beta1; a.adam_b2 = beta2;
  const int tail = (loss_mean || step_dev) ? 1 : 0;
  hipLaunchKernelGGL(k_small_first_dw, dim3((unsigned)(blocks + tail)), dim3(256), 0, (hipStream_t)stream, a);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

