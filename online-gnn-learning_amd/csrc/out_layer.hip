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
                                                       float* __restrict__ dx_self, int64_t ldx, float* __restrict__ dP, int64_t ldp) {
  const int lane = threadIdx.x;
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
      am[r][e] = argmax[dr * (int64_t)K + ce[e]];
      nb[r][e] = neigh[dr * ldn + ce[e]];
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
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int a = am[r][e];
      if (c + e >= K || a < 0 || a >= n_src) continue;
      if (!(nb[r][e] > 0.f)) continue;                             // the winner's ReLU mask (see k_reduce_bwd_max)
      atomicAdd(&dP[(int64_t)a * ldp + c + e], an[r][e]);
    }
  }
}

extern "C" int ogl_out_layer_bwd_inputs(const float* dy, int64_t lddy, int64_t n_dst, int N, int K, const float* w_self, int64_t ldws,
                                        const float* w_neigh, int64_t ldwn, const int32_t* argmax, const float* neigh, int64_t ldn,
                                        int64_t n_src, float* dx_self, int64_t ldx, float* dP, int64_t ldp, ogl_stream_t stream) {
  if (n_dst < 0 || N <= 0 || N > OB_MAX_N || K <= 0 || lddy < N || ldws < K || ldwn < K || ldn < K || ldx < K || ldp < K || n_src < 0)
    return OGL_EINVAL;
  if (n_dst == 0) return OGL_OK;
  if (!dy || !w_self || !w_neigh || !argmax || !neigh || !dx_self || !dP) return OGL_EINVAL;
  if (((uintptr_t)w_self & 15) || ((uintptr_t)w_neigh & 15) || ((uintptr_t)dx_self & 15)) return OGL_EINVAL;
  dim3 grid((unsigned)ogl_cdiv(n_dst, OB_ROWS), (unsigned)ogl_cdiv(K, 256));
  if ((K & 3) == 0 && (ldws & 3) == 0 && (ldwn & 3) == 0)
    hipLaunchKernelGGL(k_out_bwd_inputs<true>, grid, dim3(64), 0, (hipStream_t)stream, dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn,
                       argmax, neigh, ldn, n_src, dx_self, ldx, dP, ldp);
  else
    hipLaunchKernelGGL(k_out_bwd_inputs<false>, grid, dim3(64), 0, (hipStream_t)stream, dy, lddy, n_dst, N, K, w_self, ldws, w_neigh, ldwn,
                       argmax, neigh, ldn, n_src, dx_self, ldx, dP, ldp);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
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
