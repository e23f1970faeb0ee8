// Device-side prioritised replay structure: the sum tree of
// R/train/prioritized_replay/segment_tree.py:69-125 and the priority arithmetic of
// R/train/prioritized_replay/replay_buffer.py:110-245 (_normalize / add_all / update_priorities / _sample_proportional) on
// arrays in HBM, fed straight from the per-seed loss tensor of the PBR passes — no device->host transfer of losses.
//
// Tree: node[2 * cap] doubles, children of i are 2i and 2i + 1, leaves start at cap (the array form of the reference's
// tree: every partial sum is left + right of the same children, so prefix sums — and sampled indices — are the reference's).
// state[4] = {max log-priority, min log-priority, max clipped priority, min clipped priority}: the reference's RUNNING
// extrema (-1 / 99999999 when nothing has been scored).
//
// These are cold, small problems (512 leaves per train batch, <= 2e5 per priority pass): every entry point is ONE
// workgroup walking its phases with __syncthreads() between them — the phases need block-wide extrema before the
// per-element scaling, and the ancestor refresh is level-synchronous.  Integer / fp64 work, no MFMA.
#include "ogl_common.h"

#define RP_THREADS 1024

__device__ __forceinline__ double rp_block_reduce(double v, bool want_max, double* sh) {
  // wave reduce, then across the 16 waves
  for (int o = 32; o > 0; o >>= 1) {
    const double w = __shfl_xor(v, o);
    v = want_max ? fmax(v, w) : fmin(v, w);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double r = sh[0];
  for (int w = 1; w < RP_THREADS / 64; ++w) r = want_max ? fmax(r, sh[w]) : fmin(r, sh[w]);
  __syncthreads();
  return r;
}

// prio32 / prio64: the new priorities (exactly one non-null), or both null: every leaf gets the ADMISSION priority of
// R/train/graph/train_test_graph.py:78-93 — start_priority while nothing has been scored (state[2] == -1), else
// lo + 0.95 (hi - lo) of the clipped priorities seen so far.
__global__ void __launch_bounds__(RP_THREADS) k_replay_update(double* __restrict__ node, int64_t cap, const int64_t* __restrict__ idx,
                                                              const float* __restrict__ prio32, const double* __restrict__ prio64,
                                                              int64_t n, double clip_lo, double clip_hi, double offset, double alpha,
                                                              double start_priority, double* __restrict__ state,
                                                              double* __restrict__ scratch, int* __restrict__ err) {
  __shared__ double sh[RP_THREADS / 64];
  const int tid = threadIdx.x;
  double entry = 0.0;
  if (!prio32 && !prio64) {
    const double hi = state[2], lo = state[3];
    entry = hi == -1.0 ? start_priority : lo + (hi - lo) * 0.95;
  }
  __syncthreads();                                         // every thread has read the OLD state
  // clip -> running extrema of the clipped priorities
  double mx = -INFINITY, mn = INFINITY;
  for (int64_t i = tid; i < n; i += RP_THREADS) {
    double p = prio32 ? (double)prio32[i] : (prio64 ? prio64[i] : entry);
    p = p < clip_lo ? clip_lo : p;                         // np.maximum(pr, lo); a NaN stays a NaN and is caught below
    p = p > clip_hi ? clip_hi : p;
    scratch[i] = p;
    mx = fmax(mx, p); mn = fmin(mn, p);
  }
  mx = rp_block_reduce(mx, true, sh); mn = rp_block_reduce(mn, false, sh);
  if (tid == 0) { if (mx > state[2]) state[2] = mx; if (mn < state[3]) state[3] = mn; }
  // log -> running extrema of the log-priorities
  mx = -INFINITY; mn = INFINITY;
  for (int64_t i = tid; i < n; i += RP_THREADS) {
    const double lg = log(scratch[i]);
    scratch[i] = lg;
    mx = fmax(mx, lg); mn = fmin(mn, lg);
  }
  mx = rp_block_reduce(mx, true, sh); mn = rp_block_reduce(mn, false, sh);
  if (tid == 0) { if (mx > state[0]) state[0] = mx; if (mn < state[1]) state[1] = mn; }
  __threadfence_block();
  __syncthreads();
  // min-max scale with the running extrema, offset, ** alpha -> leaves
  const double lo = state[1], scale = state[0] - state[1];
  for (int64_t i = tid; i < n; i += RP_THREADS) {
    double v = scratch[i] - lo;
    if (scale > 0) v = v / scale;
    v = v + offset;
    const int64_t j = idx[i];
    if (!(v >= 0) || j < 0 || j >= cap) { *err = !(v >= 0) ? 1 : 2; continue; }      // the reference asserts v >= 0
    node[cap + j] = pow(v, alpha);
  }
  __threadfence_block();
  __syncthreads();
  if (n * 8 > cap) {                                       // many leaves: rebuild every level
    for (int64_t L = cap >> 1; L >= 1; L >>= 1) {
      for (int64_t p = L + tid; p < 2 * L; p += RP_THREADS) node[p] = node[2 * p] + node[2 * p + 1];
      __threadfence_block();
      __syncthreads();
    }
  } else {                                                 // few leaves: refresh their ancestors, level by level (threads
    for (int64_t sh_ = 1; (cap >> sh_) >= 1; ++sh_) {       // that share a parent write the same left + right)
      for (int64_t i = tid; i < n; i += RP_THREADS) {
        const int64_t j = idx[i];
        if (j < 0 || j >= cap) continue;
        const int64_t p = (cap + j) >> sh_;
        node[p] = node[2 * p] + node[2 * p + 1];
      }
      __threadfence_block();
      __syncthreads();
    }
  }
}

extern "C" int ogl_replay_update(double* node, int64_t cap, const int64_t* idx, const float* prio32, const double* prio64,
                                 int64_t n, double clip_lo, double clip_hi, double offset, double alpha, double start_priority,
                                 double* state, double* scratch, int* err_flag, ogl_stream_t stream) {
  if (cap <= 0 || (cap & (cap - 1)) || n < 0 || (prio32 && prio64) || !(alpha >= 0)) return OGL_EINVAL;
  if (n == 0) return OGL_OK;
  if (!node || !idx || !state || !scratch || !err_flag) return OGL_EINVAL;
  hipLaunchKernelGGL(k_replay_update, dim3(1), dim3(RP_THREADS), 0, (hipStream_t)stream, node, cap, idx, prio32, prio64, n, clip_lo,
                     clip_hi, offset, alpha, start_priority, state, scratch, err_flag);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// Rebuild every internal node from the leaves (after the leaf array was copied into a larger tree).
__global__ void __launch_bounds__(RP_THREADS) k_replay_rebuild(double* __restrict__ node, int64_t cap) {
  for (int64_t L = cap >> 1; L >= 1; L >>= 1) {
    for (int64_t p = L + threadIdx.x; p < 2 * L; p += RP_THREADS) node[p] = node[2 * p] + node[2 * p + 1];
    __threadfence_block();
    __syncthreads();
  }
}

extern "C" int ogl_replay_rebuild(double* node, int64_t cap, ogl_stream_t stream) {
  if (cap <= 0 || (cap & (cap - 1)) || !node) return OGL_EINVAL;
  hipLaunchKernelGGL(k_replay_rebuild, dim3(1), dim3(RP_THREADS), 0, (hipStream_t)stream, node, cap);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// sum of leaves [0, hi] in the association order of the reference's recursive range query (segment_tree.py:38-67 via
// SumSegmentTree.sum): for a prefix it is a1 + (a2 + (a3 + ...)) over the maximal left subtrees met on the way down.
__device__ double rp_prefix_sum(const double* __restrict__ node, int64_t cap, int64_t hi) {
  double term[64];
  int k = 0;
  int64_t nd = 1, nlo = 0, nhi = cap - 1;
  while (true) {
    if (hi == nhi) { term[k++] = node[nd]; break; }
    const int64_t mid = (nlo + nhi) / 2;
    if (hi <= mid) { nd = 2 * nd; nhi = mid; }
    else { term[k++] = node[2 * nd]; nd = 2 * nd + 1; nlo = mid + 1; }
  }
  double s = term[k - 1];
  for (int i = k - 2; i >= 0; --i) s = term[i] + s;
  return s;
}

// The tree walks of _sample_proportional (replay_buffer.py:164-203) for a whole batch: p_total = sum of the leaves
// [0, n_items - 1) — the reference's end-exclusive-twice quirk: the LAST stored item is outside the mass —, stratified masses
// u_strat[i] * stride + i * stride (stride = p_total / batch) and `n_redraw` proportional re-draw masses u_redraw[t] *
// p_total; out_idx[0 .. batch + n_redraw) = find_prefixsum_idx of each.  The set logic (dedup, 21 re-draws, uniform
// top-up) consumes these on the host in the reference's order.
__global__ void __launch_bounds__(256) k_replay_sample(const double* __restrict__ node, int64_t cap, int64_t n_items, int64_t batch,
                                                       const double* __restrict__ u_strat, const double* __restrict__ u_redraw,
                                                       int64_t n_redraw, int64_t* __restrict__ out_idx, double* __restrict__ out_ptotal) {
  __shared__ double ptot;
  if (threadIdx.x == 0) { ptot = rp_prefix_sum(node, cap, n_items - 2); *out_ptotal = ptot; }
  __syncthreads();
  const double p_total = ptot, stride = p_total / (double)batch;
  for (int64_t i = threadIdx.x; i < batch + n_redraw; i += 256) {
    double mass = i < batch ? u_strat[i] * stride + (double)i * stride : u_redraw[i - batch] * p_total;
    int64_t nd = 1;
    while (nd < cap) {
      const double left = node[2 * nd];
      if (left > mass) nd = 2 * nd;
      else { mass -= left; nd = 2 * nd + 1; }
    }
    out_idx[i] = nd - cap;
  }
}

extern "C" int ogl_replay_sample(const double* node, int64_t cap, int64_t n_items, int64_t batch, const double* u_strat,
                                 const double* u_redraw, int64_t n_redraw, int64_t* out_idx, double* out_ptotal,
                                 ogl_stream_t stream) {
  if (cap <= 0 || (cap & (cap - 1)) || n_items < 2 || n_items > cap || batch <= 0 || n_redraw < 0) return OGL_EINVAL;
  if (!node || !u_strat || (n_redraw > 0 && !u_redraw) || !out_idx || !out_ptotal) return OGL_EINVAL;
  hipLaunchKernelGGL(k_replay_sample, dim3(1), dim3(256), 0, (hipStream_t)stream, node, cap, n_items, batch, u_strat, u_redraw,
                     n_redraw, out_idx, out_ptotal);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// key -> storage index map maintenance: map[keys[i]] = start + i (keys outside [0, map_size) are ignored).
__global__ void __launch_bounds__(256) k_replay_note_keys(const int64_t* __restrict__ keys, int64_t n, int64_t start,
                                                          int64_t* __restrict__ map, int64_t map_size) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t k = keys[i];
  if (k >= 0 && k < map_size) map[k] = start + i;
}

extern "C" int ogl_replay_note_keys(const int64_t* keys, int64_t n, int64_t start, int64_t* map, int64_t map_size,
                                    ogl_stream_t stream) {
  if (n < 0 || start < 0 || map_size < 0) return OGL_EINVAL;
  if (n == 0) return OGL_OK;
  if (!keys || !map) return OGL_EINVAL;
  hipLaunchKernelGGL(k_replay_note_keys, dim3((unsigned)ogl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, keys, n, start, map,
                     map_size);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}


// ---- TrendPriority / HybridPriority on the device ----------------------------------------------------------------------------
// R/train/prioritized_replay/generate_priority.py:11-58: per vertex an exponentially smoothed RISE of its loss,
//   values[v] <- alpha values[v] + (1 - alpha) max(0, loss_v - prev_loss[v]),  a vertex scored for the first time starting from the
// running mean `avg` of the scored vertices' values; HybridPriority mixes loss and trend.  State in HBM (fp64, as the reference's
// np.float arrays): values[n_vertices], prev_loss[n_vertices], init[n_vertices] (1 until first scored), stats = {avg, n_items}.
// One workgroup, the reference's phase order; the two sums over the batch are strided per thread and then reduced in a fixed
// order (numpy sums pairwise: the mean agrees to fp64 rounding, not bit for bit).  ids must be distinct within a call.
__device__ __forceinline__ double rp_block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < RP_THREADS / 64; ++w) r += sh[w];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(RP_THREADS) k_priority_trend(const int64_t* __restrict__ ids, const float* __restrict__ loss32,
                                                               const double* __restrict__ loss64, int64_t n, int64_t n_vertices,
                                                               double* __restrict__ values, double* __restrict__ prev_loss,
                                                               unsigned char* __restrict__ init, double* __restrict__ stats,
                                                               double alpha, double loss_contrib, double* __restrict__ out,
                                                               int* __restrict__ err) {
  __shared__ double sh[RP_THREADS / 64];
  const int tid = threadIdx.x;
  const double avg0 = stats[0];
  double n_items = stats[1];
  __syncthreads();
  double fresh = 0.0;
  for (int64_t i = tid; i < n; i += RP_THREADS) {
    const int64_t v = ids[i];
    if (v < 0 || v >= n_vertices) { atomicExch(err, 1); continue; }
    if (init[v]) { init[v] = 0; values[v] = avg0; fresh += 1.0; }
  }
  n_items += rp_block_sum(fresh, sh);
  double s_old = 0.0, s_new = 0.0;
  for (int64_t i = tid; i < n; i += RP_THREADS) {
    const int64_t v = ids[i];
    if (v < 0 || v >= n_vertices) continue;
    const double l = loss32 ? (double)loss32[i] : loss64[i];
    const double old = values[v];
    double rise = l - prev_loss[v];
    rise = rise > 0.0 ? rise : 0.0;
    const double nv = alpha * old + (1.0 - alpha) * rise;       // (values *= alpha; values += update * (1 - alpha))
    values[v] = nv; prev_loss[v] = l;
    s_old += old; s_new += nv;
    // HybridPriority: trend * (1 - c) + loss * c, where `loss * c` is evaluated in the losses' own precision (numpy: a float32 array
    // times a Python float stays float32) and then added in float64
    const double lc = loss32 ? (double)(loss32[i] * (float)loss_contrib) : l * loss_contrib;
    out[i] = loss_contrib >= 0.0 ? nv * (1.0 - loss_contrib) + lc : nv;
  }
  s_old = rp_block_sum(s_old, sh);
  s_new = rp_block_sum(s_new, sh);
  if (tid == 0) {
    stats[0] = n_items > 0.0 ? (avg0 * n_items - s_old + s_new) / n_items : avg0;
    stats[1] = n_items;
  }
}

extern "C" int ogl_priority_trend(const int64_t* ids, const float* loss32, const double* loss64, int64_t n, int64_t n_vertices,
                                  double* values, double* prev_loss, unsigned char* init, double* stats, double alpha,
                                  double loss_contrib, double* out, int* err, ogl_stream_t stream) {
  if (n < 0 || n_vertices < 0 || !(alpha >= 0.0 && alpha <= 1.0) || loss_contrib > 1.0) return OGL_EINVAL;
  if (n == 0) return OGL_OK;
  if (!ids || (!loss32 == !loss64) || !values || !prev_loss || !init || !stats || !out || !err) return OGL_EINVAL;
  hipLaunchKernelGGL(k_priority_trend, dim3(1), dim3(RP_THREADS), 0, (hipStream_t)stream, ids, loss32, loss64, n, n_vertices, values,
                     prev_loss, init, stats, alpha, loss_contrib, out, err);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
