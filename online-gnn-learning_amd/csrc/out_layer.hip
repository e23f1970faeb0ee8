// Backward of the combine of a 'pool' layer with FEW output columns (the output layer: n_dst seeds x <= 64 classes) in two launches.
//
//     y = h[:n_dst] . Ws^T + neigh . Wn^T + bs + bn,   neigh[d, c] = max_j P[idx[d, j], c]   (argmax[d, c] = the winning source row)
// (R/train/graphsage/pytorch/aggregator_dgl.py:171,199-206 as DGL's SAGEConv('pool') runs it).  With dy [n_dst, N], N <= 64:
//     dx_self = dy . Ws              [n_dst, K]     -> added to the first n_dst rows of the layer's input gradient by its consumer
//     dneigh  = dy . Wn              [n_dst, K]     -> scattered to the winners: dP[argmax[d, c], c] += dneigh[d, c] where neigh[d, c] > 0
//     dWs = dy^T . h[:n_dst], dWn = dy^T . neigh, db = column sums of dy
// As general launches that is two 41-deep GEMMs (10 us each: pure latency), a scatter kernel (14 us) over a [n_dst, K] matrix written
// and read back in between, and two skinny weight-gradient launches (12 us each, 76 blocks).  Here:
//   k_out_bwd_inputs   one wave per (2 destination rows, 256 columns): the dy rows sit in one register (lane n = dy[d, n]) and are
//                      broadcast per term (v_readlane), every lane owns 4 columns of both products (2 x 2 x 4 accumulators, 16 float4
//                      loads of W rows from L2 in flight per trip), stores dx_self
//                      and adds the dneigh values straight into dP (float atomics, as the scatter kernel did) — dneigh never exists.
//   k_out_bwd_weights  the skinny weight-gradient scheme (linear.hip: k_bwd_weight_skinny) for BOTH products in one grid.
// fp32 FMA arithmetic on the vector ALU (50 MFLOP in all); HBM / L2-latency-bound integer + float work, no MFMA.
#include "ogl_common.h"

#define OB_MAX_N 64

#define OB_ROWS 2                   // destination rows per wave
// VEC: K and the row strides of W are multiples of 4 (every lane owns 4 whole columns or none): 16-byte row loads.  Every load
// is unconditional (clamped addresses, the value selected afterwards): a branch around a load makes the compiler drain the loads
// in flight at its join.
template <bool VEC>
__global__ void __launch_bounds__(64) k_out_bwd_inputs(const float* __restrict__ dy, int64_t lddy, int64_t n_dst, int N, int K,
                                                       const float* __restrict__ Ws, int64_t ldws, const float* __restrict__ Wn,
                                                       int64_t ldwn, const int32_t* __restrict__ argmax,
                                                       const float* __restrict__ neigh, int64_t ldn, int64_t n_src,
                                                       float* __restrict__ dx_self, int64_t ldx, float* __restrict__ dP, int64_t ldp,
                                                       const float* __restrict__ loss_rows, int64_t n_loss, float* __restrict__ loss_mean) {
  const int lane = threadIdx.x;
  if (loss_mean && blockIdx.x == 0 && blockIdx.y == 0) {
    // the mean of the per-seed losses the forward launch (ogl_out_layer_fwd_ce) left for its successor: a kernel boundary is the
    // cheapest grid-wide barrier there is (a last-block-done count inside the forward costs every block a device-scope fence, i.e.
    // an L2 write-back on this part).  Order: lane l sums rows l, l + 64, ...; then the lanes — ogl_ce_fwd_bwd_mean's.
    float t = 0.f;
    for (int64_t r = lane; r < n_loss; r += 64) t += loss_rows[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) *loss_mean = t / (float)n_loss;
  }
  const int64_t d0 = (int64_t)blockIdx.x * OB_ROWS;
  const int c = (blockIdx.y * 64 + lane) * 4;
  const bool cin = c < K;
  int ce[4];                                               // this lane's columns, clamped into the row
#pragma unroll
  for (int e = 0; e < 4; ++e) ce[e] = c + e < K ? c + e : K - 1;
  // the wave's dy rows: lane n holds dy[d, n] (N <= 64), broadcast per term with v_readlane — no load inside the loop but W's
  float dyv[OB_ROWS];
  int am[OB_ROWS][4];
  float nb[OB_ROWS][4];
#pragma unroll
  for (int r = 0; r < OB_ROWS; ++r) {
    const int64_t dr = d0 + r < n_dst ? d0 + r : n_dst - 1;
    const float v = dy[dr * lddy + (lane < N ? lane : 0)];
    dyv[r] = (lane < N && d0 + r < n_dst) ? v : 0.f;
    // what the epilogue needs, requested now: winners and ReLU masks of this lane's 4 columns
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      am[r][e] = argmax ? argmax[dr * (int64_t)K + ce[e]] : 0;
      nb[r][e] = argmax ? neigh[dr * ldn + ce[e]] : 1.f;
    }
  }
  float as[OB_ROWS][4], an[OB_ROWS][4];
#pragma unroll
  for (int r = 0; r < OB_ROWS; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) as[r][e] = an[r][e] = 0.f;
  for (int n0 = 0; n0 < N; n0 += 8) {                      // 16 row loads of W in flight per trip
    float ws[8][4], wn[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = n0 + u < N ? n0 + u : N - 1;           // a term past N re-reads the last row and is weighted by 0
      if (VEC) {
        const float4 a = *(const float4*)(Ws + (int64_t)n * ldws + ce[0]), b = *(const float4*)(Wn + (int64_t)n * ldwn + ce[0]);
        ws[u][0] = a.x; ws[u][1] = a.y; ws[u][2] = a.z; ws[u][3] = a.w; wn[u][0] = b.x; wn[u][1] = b.y; wn[u][2] = b.z; wn[u][3] = b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ws[u][e] = Ws[(int64_t)n * ldws + ce[e]]; wn[u][e] = Wn[(int64_t)n * ldwn + ce[e]]; }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int r = 0; r < OB_ROWS; ++r) {
        // lanes >= N hold 0: a term past N contributes nothing (n0 + u <= 56 + 7)
        const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dyv[r]), (n0 + u) & 63));
#pragma unroll
        for (int e = 0; e < 4; ++e) { as[r][e] += g * ws[u][e]; an[r][e] += g * wn[u][e]; }
      }
  }
  if (!cin) return;
#pragma unroll
  for (int r = 0; r < OB_ROWS; ++r) {
    const int64_t d = d0 + r;
    if (d >= n_dst) break;
    float* xo = dx_self + d * ldx + c;
    if (VEC && (ldx & 3) == 0) *(float4*)xo = make_float4(as[r][0], as[r][1], as[r][2], as[r][3]);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (c + e < K) xo[e] = as[r][e];
    }
    if (!argmax) {                                                 // DENSE form: dneigh[d, :] stored (a mean / sum aggregator's backward follows)
      float* no = dP + d * ldp + c;
      if (VEC && (ldp & 3) == 0) *(float4*)no = make_float4(an[r][0], an[r][1], an[r][2], an[r][3]);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < K) no[e] = an[r][e];
      }
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int a = am[r][e];
      if (c + e >= K || a < 0 || a >= n_src) continue;
      if (!(nb[r][e] > 0.f)) continue;                             // the winner's ReLU mask (see k_reduce_bwd_max)
      atomicAdd(&dP[(int64_t)a * ldp + c + e], an[r][e]);
    }
  }
}

static int out_bwd_inputs(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self, int64_t ldws,
                          const float* w_neigh, int64_t ldwn, const int32_t* argmax, const float* neigh, int64_t ldn,
                          int64_t n_src, float* dx_self, int64_t ldx, float* dP, int64_t ldp, const float* loss_rows, int64_t n_loss,
                          float* loss_mean, ogl_stream_t stream) {
  if (n_dst < 0 || N <= 0 || N > OB_MAX_N || K <= 0 || lddy < N || ldws < K || ldwn < K || ldn < K || ldx < K || ldp < K || n_src < 0)
    return OGL_EINVAL;
  if (n_dst == 0) return OGL_OK;
  if (!dy || !w_self || !w_neigh || !argmax || !neigh || !dx_self || !dP) return OGL_EINVAL;
  if (((uintptr_t)w_self & 15) || ((uintptr_t)w_neigh & 15) || ((uintptr_t)dx_self & 15)) return OGL_EINVAL;
  dim3 grid((unsigned)ogl_cdiv(n_dst, OB_ROWS), (unsigned)ogl_cdiv(K, 256));
  if ((K & 3) == 0 && (ldws & 3) == 0 && (ldwn & 3) == 0)
    hipLaunchKernelGGL(k_out_bwd_inputs<true>, grid, dim3(64), 0, (hipStream_t)stream, dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn,
                       argmax, neigh, ldn, n_src, dx_self, ldx, dP, ldp, loss_rows, n_loss, loss_mean);
  else
    hipLaunchKernelGGL(k_out_bwd_inputs<false>, grid, dim3(64), 0, (hipStream_t)stream, dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn,
                       argmax, neigh, ldn, n_src, dx_self, ldx, dP, ldp, loss_rows, n_loss, loss_mean);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

extern "C" int ogl_out_layer_bwd_inputs(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self, int64_t ldws,
                                        const float* w_neigh, int64_t ldwn, const int32_t* argmax, const float* neigh, int64_t ldn,
                                        int64_t n_src, float* dx_self, int64_t ldx, float* dP, int64_t ldp, ogl_stream_t stream) {
  return out_bwd_inputs(dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn, argmax, neigh, ldn, n_src, dx_self, ldx, dP, ldp, nullptr, 0,
                        nullptr, stream);
}

// ... which also finishes the loss of the forward launch: *loss_mean = sum(loss_rows[0 .. n_loss)) / n_loss (one wave of block 0)
extern "C" int ogl_out_layer_bwd_inputs_mean(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self,
                                             int64_t ldws, const float* w_neigh, int64_t ldwn, const int32_t* argmax, const float* neigh,
                                             int64_t ldn, int64_t n_src, float* dx_self, int64_t ldx, float* dP, int64_t ldp,
                                             const float* loss_rows, int64_t n_loss, float* loss_mean, ogl_stream_t stream) {
  if (!loss_rows || !loss_mean || n_loss <= 0 || n_dst <= 0) return OGL_EINVAL;
  return out_bwd_inputs(dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn, argmax, neigh, ldn, n_src, dx_self, ldx, dP, ldp, loss_rows,
                        n_loss, loss_mean, stream);
}

// Both weight gradients of the combine in one grid: blockIdx.y selects the product (x, dw); scheme and summation order of
// k_bwd_weight_skinny (linear.hip): a block owns 8 columns k of [dw | db] for all N outputs, thread (mg, kk) runs over the rows
// m = mg, mg + 64, ..., the 64 row groups are summed through LDS in a fixed order.  db (the ones column) comes from product 0.
#define OB_KT 8
#define OB_MG 64
template <int NV>
__global__ void __launch_bounds__(OB_KT * OB_MG) k_out_bwd_weights(const float* __restrict__ dy, int64_t ldy, int64_t M, int N, int K,
                                                                  const float* __restrict__ x0, int64_t ldx0,
                                                                  const int64_t* __restrict__ x0_rows, int64_t x0_nrows,
                                                                  const float* __restrict__ x1,
                                                                  int64_t ldx1, float* __restrict__ dw0, int64_t lddw0,
                                                                  float* __restrict__ dw1, int64_t lddw1, float* __restrict__ db,
                                                                  float* __restrict__ db2) {
  __shared__ float red[OB_MG][16][OB_KT + 1];
  const int tid = threadIdx.x, kk = tid & (OB_KT - 1), mg = tid / OB_KT;
  const bool second = blockIdx.y == 1;
  const float* __restrict__ x = second ? x1 : x0;
  const int64_t ldx = second ? ldx1 : ldx0;
  float* __restrict__ dw = second ? dw1 : dw0;
  const int64_t lddw = second ? lddw1 : lddw0;
  const int64_t k = (int64_t)blockIdx.x * OB_KT + kk;
  const bool kx = k < K, kone = (k == K);
  const int nvr = (N + 3) / 4;
  float4 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t m0 = mg; m0 < M; m0 += 2 * OB_MG) {
    float xv[2];
    float4 d4[2][NV];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t m = m0 + u * OB_MG;
      const bool live = m < M;
      const int64_t mm = live ? m : mg;
      int64_t r = mm;
      bool ok = live;
      if (!second && x0_rows) { r = x0_rows[mm]; ok = ok && r >= 0 && r < x0_nrows; }     // product 0 may gather its rows from a table
      const float xr = x[(ok ? r : 0) * ldx + (kx ? k : 0)];
      xv[u] = kone ? (live ? 1.f : 0.f) : ((kx && ok) ? xr : 0.f);
      const float4* dr = (const float4*)(dy + mm * ldy);
#pragma unroll
      for (int i = 0; i < NV; ++i) d4[u][i] = dr[i < nvr ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        acc[i].x += d4[u][i].x * xv[u]; acc[i].y += d4[u][i].y * xv[u];
        acc[i].z += d4[u][i].z * xv[u]; acc[i].w += d4[u][i].w * xv[u];
      }
  }
#pragma unroll
  for (int c4 = 0; c4 < NV; c4 += 4) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c4 + i < NV) {
        red[mg][4 * i + 0][kk] = acc[c4 + i].x; red[mg][4 * i + 1][kk] = acc[c4 + i].y;
        red[mg][4 * i + 2][kk] = acc[c4 + i].z; red[mg][4 * i + 3][kk] = acc[c4 + i].w;
      }
    __syncthreads();
    if (tid < 16 * OB_KT) {
      const int i = tid / OB_KT, c = tid & (OB_KT - 1);
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < OB_MG; ++g) sum += red[g][i][c];
      const int n = 4 * c4 + i;
      const int64_t kc = (int64_t)blockIdx.x * OB_KT + c;
      if (n < N && n < 4 * NV) {
        if (kc < K) dw[n * lddw + kc] = sum;
        else if (kc == K && !second) { if (db) db[n] = sum; if (db2) db2[n] = sum; }
      }
    }
  }
}

extern "C" int ogl_out_layer_bwd_weights(const float* dy, int64_t lddy, int64_t M, int N, int K, const float* x_self, int64_t ldxs,
                                         const int64_t* x_self_rows, int64_t x_self_nrows, const float* x_neigh, int64_t ldxn,
                                         float* dw_self, int64_t lddws, float* dw_neigh, int64_t lddwn, float* db, float* db2,
                                         ogl_stream_t stream) {
  if (M <= 0 || M > 4096 || N <= 0 || N > OB_MAX_N || K <= 0 || ldxs < K || ldxn < K || lddws < K || lddwn < K) return OGL_EINVAL;
  if (x_self_rows && x_self_nrows <= 0) return OGL_EINVAL;
  if (!dy || !x_self || !x_neigh || !dw_self || !dw_neigh) return OGL_EINVAL;
  if (lddy % 4 != 0 || lddy < (N + 3) / 4 * 4 || ((uintptr_t)dy & 15)) return OGL_EINVAL;     // dy rows are read as float4s
  dim3 grid((unsigned)ogl_cdiv((int64_t)K + 1, OB_KT), 2), block(OB_KT * OB_MG);
  const int nv = (N + 3) / 4;
#define OGL_OBW(NV_)                                                                                                                \
  hipLaunchKernelGGL(k_out_bwd_weights<NV_>, grid, block, 0, (hipStream_t)stream, dy, lddy, M, N, K, x_self, ldxs, x_self_rows,            \
                     x_self_nrows, x_neigh, ldxn, dw_self, lddws, dw_neigh, lddwn, db, db2)
  if (nv <= 4) OGL_OBW(4); else if (nv <= 8) OGL_OBW(8); else if (nv <= 12) OGL_OBW(12); else OGL_OBW(16);
#undef OGL_OBW
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- forward of the same layer, fused with the loss ---------------------------------------------------------------------------
// The output layer of a train step is three latency-bound launches for 512 seeds — the neighbour max over the pooled rows
// (12 us), the [n_dst, 2K] -> N projection (17 us), the cross entropy (18 us) — on a chip that is otherwise idle.  Here one block
// owns R destinations from the gather to the loss:
//   A  every thread owns one 16-byte column chunk: the destination's S pooled rows in flight 16 at a time (unconditional loads, a
//      missing neighbour reads row 0 and is skipped: k_reduce_fwd_v4's rule), max + first-winner argmax in registers; the pooled row
//      and the destination's own input row go to LDS (and neigh / argmax to memory: the backward reads them);
//   B  wave w takes the classes w, w + 4, ...: lanes stride the 2K-long reduction in float4s (W rows from L2, the two activation
//      rows from LDS), a butterfly sum per (class, destination), + both biases -> the logits in LDS;
//   C  wave r finishes destination r: log-sum-exp over its N <= 64 logits (one per lane), the per-seed loss, dlogits scaled by
//      1 / B (the arithmetic and order of k_ce_fwd_bwd_mean_grid: identical bits for identical logits); the LAST block to finish
//      sums the row losses in the one-workgroup order into the mean.
// The grid also zeroes a caller buffer on the side (the atomic-scatter target of the backward pass that follows).
// Replaces ogl_reduce_fwd(max) + ogl_linear_fwd(dual) + ogl_ce_fwd_bwd_mean_grid of the live layer's last SAGEConv + the loss
// (DGL SAGEConv('pool') imported at R/train/graphsage/pytorch/graphsage_dgl.py:3; nn.CrossEntropyLoss R/.../pytorch/model.py:20,105,198).
#define OF_THREADS 256
#define OF_MAX_K 1024
#define OF_MAX_S 64
#define OF_U 16

struct OutFwdArgs {
  const float* P; int64_t ldp; int64_t n_src;           // relu(fc_pool(h)) rows [n_src, K]
  const int32_t* idx; int S;                            // block-local neighbour indices [n_dst, S] (-1: none)
  const float* h; int64_t ldh;                          // the layer's input; rows [0, n_dst) are the destinations' own
  const float* Ws; int64_t ldws; const float* Wn; int64_t ldwn; const float* bs; const float* bn;
  int64_t n_dst; int K; int N;
  float* neigh; int64_t ldn; int32_t* argmax;           // out [n_dst, K] (argmax dense, row stride K; null: not kept)
  float* logits; int64_t ldl;                           // out [n_dst, N]
  const int64_t* labels; const int64_t* label_ids; int64_t n_labels;   // label of row i = labels[label_ids ? label_ids[i] : i]
  float grad_scale;
  float* loss_rows; float* dlogits; int64_t lddl; float* loss_mean; unsigned* counter;   // dlogits / loss_mean nullable
  float4* zero_buf; int64_t zero_n4;
};

__device__ __forceinline__ float of_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float of_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// MEAN: the neighbour reduction is the MEAN over the S sampled rows of P (the in-repo 'mean' layer's aggregator over its own input,
// R/train/graphsage/pytorch/aggregator_dgl.py:156-159: P = h, w_self / w_neigh = the two column blocks of fc_neigh's concat weight),
// summed in slot order and divided by S as ogl_reduce_fwd(OGL_REDUCE_MEAN) does; no argmax.
template <int R, bool MEAN = false>
__global__ void __launch_bounds__(OF_THREADS) k_out_fwd_ce(OutFwdArgs a) {
  __shared__ float4 hrow[R][OF_MAX_K / 4], nrow[R][OF_MAX_K / 4];
  __shared__ int sidx[R][OF_MAX_S];
  __shared__ float slog[R][64];
  __shared__ int last;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (a.zero_n4 > 0) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * OF_THREADS + tid; i < a.zero_n4; i += (int64_t)gridDim.x * OF_THREADS) a.zero_buf[i] = z;
  }
  const int64_t d0 = (int64_t)blockIdx.x * R;
  if (tid < R * 64) {
    const int r = tid >> 6, j = tid & 63;
    const int64_t d = d0 + r;
    sidx[r][j] = (d < a.n_dst && j < a.S) ? a.idx[d * a.S + j] : -1;
  }
  __syncthreads();
  const int K4 = a.K >> 2;
  const bool cin = tid < K4;
  const int ch = cin ? tid : K4 - 1;
  // ---- A: neighbour max of the pooled rows, the destination's own row
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t d = d0 + r;
    const bool live = d < a.n_dst;                                      // (block-uniform)
    const float4 hv = ((const float4*)(a.h + (live ? d : 0) * a.ldh))[ch];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int arg[4] = {-1, -1, -1, -1};
    bool any = false;
    for (int j0 = 0; j0 < a.S && live; j0 += OF_U) {
      float4 v[OF_U];
      int q[OF_U];
      bool ok[OF_U];
#pragma unroll
      for (int u = 0; u < OF_U; ++u) {
        const int j = j0 + u;
        const int qq = sidx[r][j < OF_MAX_S ? j : OF_MAX_S - 1];
        ok[u] = j < a.S && qq >= 0 && qq < a.n_src;
        q[u] = ok[u] ? qq : 0;
      }
#pragma unroll
      for (int u = 0; u < OF_U; ++u) v[u] = ((const float4*)(a.P + (int64_t)q[u] * a.ldp))[ch];
#pragma unroll
      for (int u = 0; u < OF_U; ++u) {
        if (!ok[u]) continue;                                           // block-uniform
        if constexpr (MEAN) {
          acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        } else if (!any) {
          acc = v[u];
          arg[0] = arg[1] = arg[2] = arg[3] = q[u];
        } else {
          if (v[u].x > acc.x) { acc.x = v[u].x; arg[0] = q[u]; }
          if (v[u].y > acc.y) { acc.y = v[u].y; arg[1] = q[u]; }
          if (v[u].z > acc.z) { acc.z = v[u].z; arg[2] = q[u]; }
          if (v[u].w > acc.w) { acc.w = v[u].w; arg[3] = q[u]; }
        }
        any = true;
      }
    }
    if constexpr (MEAN) {
      if (any) { const float fS = (float)a.S; acc.x /= fS; acc.y /= fS; acc.z /= fS; acc.w /= fS; }
    }
    if (cin) {
      hrow[r][tid] = live ? hv : make_float4(0.f, 0.f, 0.f, 0.f);
      nrow[r][tid] = acc;
      if (live) {
        ((float4*)(a.neigh + d * a.ldn))[tid] = acc;
        if (a.argmax) ((int4*)(a.argmax + d * (int64_t)a.K))[tid] = make_int4(arg[0], arg[1], arg[2], arg[3]);
      }
    }
  }
  __syncthreads();
  // ---- B: logits[r][c] = h_r . Ws[c] + neigh_r . Wn[c] + bs[c] + bn[c]; wave wv takes classes wv, wv + 4, ...
  // (the W rows of the NEXT class are requested before the current class is multiplied: one L2 latency per class otherwise)
  constexpr int NIT = OF_MAX_K / 256;
  float4 wbuf[2][2][NIT];
  auto load_w = [&](int c, int b) __attribute__((always_inline)) {
    const int cc = c < a.N ? c : a.N - 1;
    const float4* wsr = (const float4*)(a.Ws + (int64_t)cc * a.ldws);
    const float4* wnr = (const float4*)(a.Wn + (int64_t)cc * a.ldwn);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {                                  // every load unconditional (clamped), masked below
      const int k4 = lane + 64 * it;
      wbuf[b][0][it] = wsr[k4 < K4 ? k4 : 0]; wbuf[b][1][it] = wnr[k4 < K4 ? k4 : 0];
    }
  };
  auto class_c = [&](int c, int b) __attribute__((always_inline)) {
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k4 = lane + 64 * it;
      if (k4 < K4) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float4 hv = hrow[r][k4], nv = nrow[r][k4];
          const float4 w0 = wbuf[b][0][it], w1 = wbuf[b][1][it];
          s[r] += hv.x * w0.x + hv.y * w0.y + hv.z * w0.z + hv.w * w0.w;
          s[r] += nv.x * w1.x + nv.y * w1.y + nv.z * w1.z + nv.w * w1.w;
        }
      }
    }
    const float bias = (a.bs ? a.bs[c] : 0.f) + (a.bn ? a.bn[c] : 0.f);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float t = of_wave_sum(s[r]);
      if (lane == 0) slog[r][c] = t + bias;
    }
  };
  if (wv < a.N) load_w(wv, 0);
  for (int c = wv; c < a.N; c += 8) {                                   // two classes per trip: static buffer indices
    load_w(c + 4, 1);
    class_c(c, 0);
    if (c + 4 < a.N) {
      load_w(c + 8, 0);
      class_c(c + 4, 1);
    }
  }
  __syncthreads();
  // ---- C: the loss of destination wv (R <= 4 waves)
  if (wv < R && d0 + wv < a.n_dst) {
    const int64_t row = d0 + wv;
    const float x = lane < a.N ? slog[wv][lane] : -INFINITY;
    if (lane < a.N) a.logits[row * a.ldl + lane] = x;
    const float m = of_wave_max(x);
    const float s = of_wave_sum(lane < a.N ? expf(x - m) : 0.f);
    const float lse = m + logf(s);
    int64_t y;
    if (a.label_ids) {
      const int64_t id = a.label_ids[row];
      y = (id >= 0 && id < a.n_labels) ? a.labels[id] : -1;
    } else y = a.labels[row];
    const bool ok = y >= 0 && y < a.N;
    const float xy = __shfl(x, ok ? (int)y : 0);
    if (lane == 0) a.loss_rows[row] = ok ? lse - xy : 0.f;
    const float dl = a.grad_scale * (expf(x - lse) - ((ok && lane == (int)y) ? 1.f : 0.f));
    if (a.dlogits && lane < a.N) a.dlogits[row * a.lddl + lane] = dl;
  }
  if (!a.loss_mean) return;
  if (!a.counter) {
    // the mean is left to the successor launch (ogl_out_layer_bwd_inputs_mean): until then the tensor reads NaN, not stale memory
    if (blockIdx.x == 0 && tid == 0) *a.loss_mean = __builtin_nanf("");
    return;
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) last = atomicAdd(a.counter, 1u) == gridDim.x - 1u;
  __syncthreads();
  if (last && tid < 64) {
    __threadfence();
    float t = 0.f;
    for (int64_t r = lane; r < a.n_dst; r += 64) t += __builtin_nontemporal_load(a.loss_rows + r);
    t = of_wave_sum(t);
    if (lane == 0) { *a.loss_mean = t / (float)a.n_dst; *a.counter = 0u; }
  }
}

extern "C" int ogl_out_layer_fwd_ce_fits(int64_t n_dst, int fanout, int K, int N) {
  return (n_dst > 0 && n_dst <= (1 << 20) && fanout > 0 && fanout <= OF_MAX_S && K >= 4 && K <= OF_MAX_K && (K & 3) == 0 && N > 0 && N <= 64) ? 1 : 0;
}

static int out_fwd_ce(const float* P, int64_t ldp, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout,
                      const float* h, int64_t ldh, int K, const float* w_self, int64_t ldws, const float* w_neigh,
                      int64_t ldwn, const float* b_self, const float* b_neigh, int N, float* neigh, int64_t ldn,
                      int32_t* argmax, float* logits, int64_t ldl, const int64_t* label_table, int64_t n_labels,
                      const int64_t* label_ids, float grad_scale, float* loss_rows, float* dlogits, int64_t lddl,
                      float* loss_mean, unsigned int* counter, float* zero_buf, int64_t zero_floats, int rows_per_block,
                      ogl_stream_t stream, int mean = 0) {
  if (!ogl_out_layer_fwd_ce_fits(n_dst, fanout, K, N) || n_src <= 0 || n_src < n_dst) return OGL_EINVAL;
  if (ldp < K || ldh < K || ldws < K || ldwn < K || ldn < K || ldl < N || (dlogits && lddl < N) || n_labels < 0) return OGL_EINVAL;
  if (!P || !idx || !h || !w_self || !w_neigh || !neigh || !logits || !label_table || !loss_rows) return OGL_EINVAL;
  // (loss_mean without a counter: the mean is DEFERRED — this launch writes NaN there, ogl_out_layer_bwd_inputs_mean the value)
  // 16-byte row accesses everywhere
  if (((ldp | ldh | ldws | ldwn | ldn) & 3) || (((uintptr_t)P | (uintptr_t)h | (uintptr_t)w_self | (uintptr_t)w_neigh | (uintptr_t)neigh |
                                                 (uintptr_t)argmax) & 15))
    return OGL_EINVAL;
  if (zero_floats < 0 || (zero_floats > 0 && (!zero_buf || ((uintptr_t)zero_buf & 15) || (zero_floats & 3)))) return OGL_EINVAL;
  OutFwdArgs a;
  a.P = P; a.ldp = ldp; a.n_src = n_src; a.idx = idx; a.S = fanout; a.h = h; a.ldh = ldh;
  a.Ws = w_self; a.ldws = ldws; a.Wn = w_neigh; a.ldwn = ldwn; a.bs = b_self; a.bn = b_neigh;
  a.n_dst = n_dst; a.K = K; a.N = N; a.neigh = neigh; a.ldn = ldn; a.argmax = argmax; a.logits = logits; a.ldl = ldl;
  a.labels = label_table; a.label_ids = label_ids; a.n_labels = n_labels; a.grad_scale = grad_scale;
  a.loss_rows = loss_rows; a.dlogits = dlogits; a.lddl = lddl; a.loss_mean = loss_mean; a.counter = counter;
  a.zero_buf = (float4*)zero_buf; a.zero_n4 = zero_floats / 4;
  // one destination per block while that still is at most two blocks per CU; two beyond
  const int R = rows_per_block > 0 ? rows_per_block : (n_dst <= 512 ? 1 : 2);
  if (mean) {
    if (argmax) return OGL_EINVAL;
    if (R == 1) hipLaunchKernelGGL((k_out_fwd_ce<1, true>), dim3((unsigned)n_dst), dim3(OF_THREADS), 0, (hipStream_t)stream, a);
    else if (R == 2) hipLaunchKernelGGL((k_out_fwd_ce<2, true>), dim3((unsigned)ogl_cdiv(n_dst, 2)), dim3(OF_THREADS), 0, (hipStream_t)stream, a);
    else return OGL_EINVAL;
    OGL_CHECK_LAUNCH();
    return OGL_OK;
  }
  if (R == 1) hipLaunchKernelGGL(k_out_fwd_ce<1>, dim3((unsigned)n_dst), dim3(OF_THREADS), 0, (hipStream_t)stream, a);
  else if (R == 2) hipLaunchKernelGGL(k_out_fwd_ce<2>, dim3((unsigned)ogl_cdiv(n_dst, 2)), dim3(OF_THREADS), 0, (hipStream_t)stream, a);
  else if (R == 4) hipLaunchKernelGGL(k_out_fwd_ce<4>, dim3((unsigned)ogl_cdiv(n_dst, 4)), dim3(OF_THREADS), 0, (hipStream_t)stream, a);
  else return OGL_EINVAL;
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

extern "C" int ogl_out_layer_fwd_ce(const float* P, int64_t ldp, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout,
                                    const float* h, int64_t ldh, int K, const float* w_self, int64_t ldws, const float* w_neigh,
                                    int64_t ldwn, const float* b_self, const float* b_neigh, int N, float* neigh, int64_t ldn,
                                    int32_t* argmax, float* logits, int64_t ldl, const int64_t* label_table, int64_t n_labels,
                                    const int64_t* label_ids, float grad_scale, float* loss_rows, float* dlogits, int64_t lddl,
                                    float* loss_mean, unsigned int* counter, float* zero_buf, int64_t zero_floats, int rows_per_block,
                                    ogl_stream_t stream) {
  return out_fwd_ce(P, ldp, n_src, idx, n_dst, fanout, h, ldh, K, w_self, ldws, w_neigh, ldwn, b_self, b_neigh, N, neigh, ldn, argmax, logits,
                    ldl, label_table, n_labels, label_ids, grad_scale, loss_rows, dlogits, lddl, loss_mean, counter, zero_buf, zero_floats,
                    rows_per_block, stream);
}

// *loss_mean = sum(loss_rows[0 .. n)) / n in the order of ogl_out_layer_bwd_inputs_mean (lane l sums rows l, l + 64, ...; then the lanes):
// the deferred mean of ogl_out_layer_fwd_ce / _bwd when no ogl_out_layer_bwd_inputs_mean launch follows.  One wave.
__global__ void __launch_bounds__(64) k_loss_mean_finish(const float* __restrict__ loss_rows, int64_t n, float* __restrict__ loss_mean) {
  const int lane = threadIdx.x;
  float t = 0.f;
  for (int64_t r = lane; r < n; r += 64) t += loss_rows[r];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  if (lane == 0) *loss_mean = t / (float)n;
}

extern "C" int ogl_loss_mean_finish(const float* loss_rows, int64_t n, float* loss_mean, ogl_stream_t stream) {
  if (!loss_rows || !loss_mean || n <= 0) return OGL_EINVAL;
  hipLaunchKernelGGL(k_loss_mean_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_rows, n, loss_mean);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// The in-repo 'mean' layer as the LAST layer of a train step, fused with the loss (k_out_fwd_ce<R, true>): neigh[d, :] = the mean
// over the fanout sampled rows of P (slot order, divided by fanout: ogl_reduce_fwd(OGL_REDUCE_MEAN)'s arithmetic), logits = h[d] . w_self^T
// + neigh[d] . w_neigh^T + b_self (+ b_neigh) where w_self / w_neigh are the two column blocks of fc_neigh's concat weight (row stride
// ldws = ldwn = its width), then the cross entropy as ogl_out_layer_fwd_ce (deferred mean with counter == NULL).  Replaces
// mailbox.mean + torch.cat + nn.Linear + nn.CrossEntropyLoss (R/train/graphsage/pytorch/aggregator_dgl.py:156-159,199-206;
// pytorch/model.py:105).
extern "C" int ogl_out_layer_fwd_ce_mean(const float* P, int64_t ldp, int64_t n_src, const int32_t* idx, int64_t n_dst, int fanout,
                                         const float* h, int64_t ldh, int K, const float* w_self, int64_t ldws, const float* w_neigh,
                                         int64_t ldwn, const float* b_self, const float* b_neigh, int N, float* neigh, int64_t ldn,
                                         float* logits, int64_t ldl, const int64_t* label_table, int64_t n_labels,
                                         const int64_t* label_ids, float grad_scale, float* loss_rows, float* dlogits, int64_t lddl,
                                         float* loss_mean, unsigned int* counter, int rows_per_block, ogl_stream_t stream) {
  return out_fwd_ce(P, ldp, n_src, idx, n_dst, fanout, h, ldh, K, w_self, ldws, w_neigh, ldwn, b_self, b_neigh, N, neigh, ldn, nullptr, logits,
                    ldl, label_table, n_labels, label_ids, grad_scale, loss_rows, dlogits, lddl, loss_mean, counter, nullptr, 0,
                    rows_per_block, stream, 1);
}

// ogl_out_layer_bwd_inputs for an aggregator WITHOUT winners (mean / sum): dx_self [n_dst, K] = dy . w_self and dneigh [n_dst, K] =
// dy . w_neigh both STORED (the aggregator's own backward — ogl_reduce_bwd_seg_apply — follows); optionally finishes the deferred loss
// mean like ogl_out_layer_bwd_inputs_mean (loss_rows / loss_mean nullable together).
extern "C" int ogl_out_layer_bwd_inputs_dense(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self, int64_t ldws,
                                              const float* w_neigh, int64_t ldwn, float* dx_self, int64_t ldx, float* dneigh, int64_t lddn,
                                              const float* loss_rows, int64_t n_loss, float* loss_mean, ogl_stream_t stream) {
  if (n_dst < 0 || N <= 0 || N > OB_MAX_N || K <= 0 || lddy < N || ldws < K || ldwn < K || ldx < K || lddn < K) return OGL_EINVAL;
  if ((loss_rows == nullptr) != (loss_mean == nullptr) || (loss_mean && n_loss <= 0)) return OGL_EINVAL;
  if (n_dst == 0) return OGL_OK;
  if (!dy || !w_self || !w_neigh || !dx_self || !dneigh) return OGL_EINVAL;
  if (((uintptr_t)w_self & 15) || ((uintptr_t)w_neigh & 15) || ((uintptr_t)dx_self & 15) || ((uintptr_t)dneigh & 15)) return OGL_EINVAL;
  dim3 grid((unsigned)ogl_cdiv(n_dst, OB_ROWS), (unsigned)ogl_cdiv(K, 256));
  if ((K & 3) == 0 && (ldws & 3) == 0 && (ldwn & 3) == 0)
    hipLaunchKernelGGL(k_out_bwd_inputs<true>, grid, dim3(64), 0, (hipStream_t)stream, dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn,
                       (const int32_t*)nullptr, (const float*)nullptr, (int64_t)0, (int64_t)0, dx_self, ldx, dneigh, lddn, loss_rows, n_loss,
                       loss_mean);
  else
    hipLaunchKernelGGL(k_out_bwd_inputs<false>, grid, dim3(64), 0, (hipStream_t)stream, dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn,
                       (const int32_t*)nullptr, (const float*)nullptr, (int64_t)0, (int64_t)0, dx_self, ldx, dneigh, lddn, loss_rows, n_loss,
                       loss_mean);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
