// Block construction (dgl.to_block semantics restated): dst-first, first-appearance relabelling.
// Replaces the relabel step inside dgl.sampling.NodeDataLoader
// (R/train/graphsage/pytorch/model.py:45-47,129-131,175-178,225-227,281-284,313-316).
//
// Deterministic on the GPU: a global open-addressing hash keyed by vertex id records the MINIMUM
// flat position p at which the id appears in [dst | picks row-major]; positions that own their
// id's minimum are "first appearances"; a 3-phase exclusive scan of those flags gives the local
// index.  Integer/HBM-bound work, no MFMA.
#include "ogl_common.h"
#include <cstdlib>
#include <algorithm>

#define BLK_SCAN 1024

struct block_ws {
  int32_t* tkey;   // [T]  vertex id or -1
  int32_t* tmin;   // [T]  min flat position
  int32_t* tlidx;  // [T]  local index of the id
  int32_t* slot;   // [P]  table slot of position p (or -1)
  int32_t* bsum;   // [NB + 1]
  int64_t T, P, NB;
};

static inline int64_t table_size(int64_t P) {
  int64_t T = 1024;
  while (T < 2 * P) T <<= 1;
  return T;
}

static inline int64_t ws_bytes(int64_t P) {
  int64_t T = table_size(P);
  int64_t NB = ogl_cdiv(P, BLK_SCAN);
  return 4 * (3 * T + ogl_round_up(P, 4) + ogl_round_up(NB + 1, 4));
}

extern "C" int64_t ogl_block_workspace_bytes(int64_t n_dst, int fanout) {
  if (n_dst < 0 || fanout < 0) return OGL_EINVAL;
  return ws_bytes(n_dst * (1 + (int64_t)fanout));
}

// table reset as a KERNEL (tkey = -1, tmin = INT_MAX-ish): the two arrays are adjacent, one launch fills both.  (A
// hipMemsetAsync node recorded into a captured hipGraph re-runs on only 1/16 of its range from the second replay on — ROCm
// 7.2, shown by tools/graph_probe.py —: stale keys filled the table and every insert probed all of it, 17-57 ms per build.)
// (optionally also fills `pad` [0, n_pad) with -1: the source list of a padded build, so that the fill is not a launch of its own)
__global__ void __launch_bounds__(256) k_block_reset(int32_t* __restrict__ tkey_tmin, int64_t T, int64_t* __restrict__ pad, int64_t n_pad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * T; i += (int64_t)gridDim.x * blockDim.x)
    tkey_tmin[i] = i < T ? -1 : 0x7F7F7F7F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) pad[i] = -1;
}

static inline void launch_block_reset(int32_t* tkey, int64_t T, hipStream_t stream, int64_t* pad = nullptr, int64_t n_pad = 0) {
  hipLaunchKernelGGL(k_block_reset, dim3((unsigned)std::min<int64_t>(ogl_cdiv(2 * T, 256), 2048)), dim3(256), 0, stream, tkey, T, pad,
                     n_pad);
}

__device__ __forceinline__ int64_t flat_id(const int64_t* __restrict__ dst, const int64_t* __restrict__ picks,
                                           int64_t n_dst, int64_t p) {
  return p < n_dst ? dst[p] : picks[p - n_dst];
}

__global__ void __launch_bounds__(256) k_block_insert(const int64_t* __restrict__ dst,
                                                      const int64_t* __restrict__ picks, int64_t n_dst,
                                                      int64_t P, int32_t* tkey, int32_t* tmin,
                                                      int32_t* __restrict__ slot, uint32_t mask, int shift) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int64_t id64 = flat_id(dst, picks, n_dst, p);
  if (id64 < 0) { slot[p] = -1; return; }
  int32_t id = (int32_t)id64;
  uint32_t h = ((uint32_t)id * 0x9E3779B1u) >> shift;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    int32_t old = atomicCAS(&tkey[h], -1, id);
    if (old == -1 || old == id) {
      atomicMin(&tmin[h], (int32_t)p);
      slot[p] = (int32_t)h;
      return;
    }
    h = (h + 1) & mask;
  }
  slot[p] = -1;  // unreachable: the table holds >= 2P slots
}

__device__ __forceinline__ int is_first(const int32_t* __restrict__ tmin, const int32_t* __restrict__ slot,
                                        int64_t n_dst, int64_t P, int64_t p) {
  if (p >= P) return 0;
  if (p < n_dst) return 1;  // every dst keeps its own source row
  int32_t s = slot[p];
  return (s >= 0 && tmin[s] == (int32_t)p) ? 1 : 0;
}

// exclusive scan of one flag per thread over a 1024-thread block; returns prefix, *total = sum
__device__ __forceinline__ int block_scan_1024(int f, int* total) {
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned long long b = __ballot(f);
  int pre = __popcll(b & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wid] = __popcll(b);
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    int s = wsum[w];
    if (w < wid) woff += s;
    tot += s;
  }
  *total = tot;
  __syncthreads();
  return pre + woff;
}

__global__ void __launch_bounds__(BLK_SCAN) k_block_count(const int32_t* __restrict__ tmin,
                                                          const int32_t* __restrict__ slot, int64_t n_dst,
                                                          int64_t P, int32_t* __restrict__ bsum) {
  int64_t p = (int64_t)blockIdx.x * BLK_SCAN + threadIdx.x;
  int tot;
  (void)block_scan_1024(is_first(tmin, slot, n_dst, P, p), &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: exclusive scan of bsum[0..NB) in place, bsum[NB] = total, n_src_out = total
__global__ void __launch_bounds__(BLK_SCAN) k_block_scan_sums(int32_t* bsum, int64_t NB, int64_t* n_src_out) {
  __shared__ int carry_s;
  __shared__ int part[BLK_SCAN];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < NB; base += BLK_SCAN) {
    int64_t i = base + threadIdx.x;
    int v = i < NB ? bsum[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    // Hillis-Steele inclusive scan in LDS (NB is a few hundred; this runs once per block build)
    for (int off = 1; off < BLK_SCAN; off <<= 1) {
      int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    int incl = part[threadIdx.x];
    int carry = carry_s;
    if (i < NB) bsum[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == BLK_SCAN - 1) carry_s = carry + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) { bsum[NB] = carry_s; *n_src_out = (int64_t)carry_s; }
}

__global__ void __launch_bounds__(BLK_SCAN) k_block_assign(const int64_t* __restrict__ dst,
                                                           const int64_t* __restrict__ picks,
                                                           const int32_t* __restrict__ tmin,
                                                           const int32_t* __restrict__ slot,
                                                           const int32_t* __restrict__ bsum, int64_t n_dst,
                                                           int64_t P, int64_t* __restrict__ src_ids,
                                                           int32_t* __restrict__ tlidx) {
  int64_t p = (int64_t)blockIdx.x * BLK_SCAN + threadIdx.x;
  int f = is_first(tmin, slot, n_dst, P, p);
  int tot;
  int pre = block_scan_1024(f, &tot);
  if (f) {
    int32_t li = bsum[blockIdx.x] + pre;
    src_ids[li] = flat_id(dst, picks, n_dst, p);
    int32_t s = slot[p];
    if (s >= 0 && tmin[s] == (int32_t)p) tlidx[s] = li;
  }
}

__global__ void __launch_bounds__(256) k_block_lookup(const int32_t* __restrict__ slot,
                                                      const int32_t* __restrict__ tlidx, int64_t n_dst,
                                                      int64_t P, int32_t* __restrict__ local_idx) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P - n_dst) return;
  int32_t s = slot[n_dst + e];
  local_idx[e] = s >= 0 ? tlidx[s] : -1;
}

// ---- small blocks: the whole relabelling in ONE workgroup ------------------------------------------------------------------
// A 32-seed batch relabels P = n_dst (1 + fanout) <= ~22 k positions: the six launches above are pure launch latency there
// (4-5 us each inside a replayed graph).  One 1024-thread block walks the same phases with __syncthreads() between them —
// table reset, insert (atomicMin of the flat position), first-appearance flags + scan (carried across 1024-position chunks),
// assign, lookup — and, optionally, fills src_ids past the source count with -1.  Same results as the multi-launch path.
#define BLK_SMALL_MAX_P 131072          // (B = 32, fanout 45 — R/settings/pubmed.json, bitcoin.json — is 67 712 positions)
#define BLK_SMALL_ONE_WG_P 4096      // up to here the whole build runs in one workgroup (ogl_build_block_padded)
__global__ void __launch_bounds__(BLK_SCAN) k_block_build_small(const int64_t* __restrict__ dst, const int64_t* __restrict__ picks,
                                                                int64_t n_dst, int64_t P, int32_t* tkey, int32_t* tmin,
                                                                int32_t* tlidx, int32_t* slot, uint32_t mask, int shift, int64_t T,
                                                                int64_t* __restrict__ src_ids, int64_t src_cap, int pad_tail,
                                                                int64_t* __restrict__ n_src_out, int32_t* __restrict__ local_idx) {
  __shared__ int carry_s;
  const int tid = threadIdx.x;
  for (int64_t i = tid; i < 2 * T; i += BLK_SCAN) tkey[i] = i < T ? -1 : 0x7F7F7F7F;     // tkey | tmin are adjacent
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t p = tid; p < P; p += BLK_SCAN) {
    const int64_t id64 = flat_id(dst, picks, n_dst, p);
    int32_t sl = -1;
    if (id64 >= 0) {
      const int32_t id = (int32_t)id64;
      uint32_t h = ((uint32_t)id * 0x9E3779B1u) >> shift;
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        const int32_t old = atomicCAS(&tkey[h], -1, id);
        if (old == -1 || old == id) { atomicMin(&tmin[h], (int32_t)p); sl = (int32_t)h; break; }
        h = (h + 1) & mask;
      }
    }
    slot[p] = sl;
  }
  __threadfence_block();
  __syncthreads();
  for (int64_t base = 0; base < P; base += BLK_SCAN) {
    const int64_t p = base + tid;
    // tmin was last written by device-scope atomics of OTHER threads in this launch: read it with a device-coherent load
    // (a plain load may be served from an L1 line fetched before those atomics; kernel boundaries did that job above)
    const int32_t sp = p < P ? slot[p] : -1;
    const bool own = sp >= 0 && __hip_atomic_load(&tmin[sp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)p;
    const int f = p < P && (p < n_dst || own) ? 1 : 0;
    int tot;
    const int pre = block_scan_1024(f, &tot);
    const int carry = carry_s;
    if (f) {
      const int32_t li = carry + pre;
      src_ids[li] = flat_id(dst, picks, n_dst, p);
      if (own) tlidx[sp] = li;
    }
    __syncthreads();
    if (tid == 0) carry_s = carry + tot;
    __syncthreads();
  }
  const int n_src = carry_s;
  if (tid == 0) *n_src_out = (int64_t)n_src;
  if (pad_tail)
    for (int64_t i = n_src + tid; i < src_cap; i += BLK_SCAN) src_ids[i] = -1;
  __threadfence_block();
  __syncthreads();
  for (int64_t e = tid; e < P - n_dst; e += BLK_SCAN) {
    const int32_t s = slot[n_dst + e];
    local_idx[e] = s >= 0 ? tlidx[s] : -1;
  }
}

// ogl_build_block with the tail of src_ids [n_src, src_cap) set to -1 ("no vertex"): what a captured step needs of a block
// whose source count stays on the device.  src_cap >= n_dst (1 + fanout).
extern "C" int ogl_build_block_padded(const int64_t* dst, int64_t n_dst, const int64_t* picks, int fanout, int64_t* src_ids,
                                      int64_t src_cap, int64_t* n_src_out, int32_t* local_idx, void* workspace,
                                      int64_t workspace_bytes, ogl_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_dst <= 0 || fanout <= 0 || !n_src_out || !dst || !src_ids || !picks || !local_idx) return OGL_EINVAL;
  const int64_t P = n_dst * (1 + (int64_t)fanout);
  if (src_cap < P || P >= ((int64_t)1 << 30)) return OGL_EINVAL;
  if (!workspace || workspace_bytes < ws_bytes(P)) return OGL_EWORKSPACE;
  const int64_t T = table_size(P);
  int32_t* base = (int32_t*)workspace;
  int logT = 0; while (((int64_t)1 << logT) < T) ++logT;
  if (P > BLK_SMALL_ONE_WG_P) {
    // larger blocks: the parallel phases of ogl_build_block (one workgroup's throughput loses to them from a few thousand
    // positions on: P = 21 632 is 35 us slower in one workgroup), with the -1 fill of src_ids folded into the table reset
    block_ws ws;
    ws.T = T; ws.P = P; ws.NB = ogl_cdiv(P, BLK_SCAN);
    ws.tkey = base; ws.tmin = base + T; ws.tlidx = base + 2 * T; ws.slot = base + 3 * T;
    ws.bsum = ws.slot + ogl_round_up(P, 4);
    launch_block_reset(ws.tkey, T, stream, src_ids, src_cap);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_insert, dim3((unsigned)ogl_cdiv(P, 256)), dim3(256), 0, stream, dst, picks, n_dst, P, ws.tkey, ws.tmin,
                       ws.slot, (uint32_t)(T - 1), 32 - logT);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_count, dim3((unsigned)ws.NB), dim3(BLK_SCAN), 0, stream, ws.tmin, ws.slot, n_dst, P, ws.bsum);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_scan_sums, dim3(1), dim3(BLK_SCAN), 0, stream, ws.bsum, ws.NB, n_src_out);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_assign, dim3((unsigned)ws.NB), dim3(BLK_SCAN), 0, stream, dst, picks, ws.tmin, ws.slot, ws.bsum, n_dst, P,
                       src_ids, ws.tlidx);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_lookup, dim3((unsigned)ogl_cdiv(P - n_dst, 256)), dim3(256), 0, stream, ws.slot, ws.tlidx, n_dst, P,
                       local_idx);
    OGL_CHECK_LAUNCH();
    return OGL_OK;
  }
  hipLaunchKernelGGL(k_block_build_small, dim3(1), dim3(BLK_SCAN), 0, stream, dst, picks, n_dst, P, base, base + T, base + 2 * T,
                     base + 3 * T, (uint32_t)(T - 1), 32 - logT, T, src_ids, src_cap, 1, n_src_out, local_idx);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

extern "C" int ogl_build_block(const int64_t* dst, int64_t n_dst, const int64_t* picks, int fanout,
                               int64_t* src_ids, int64_t* n_src_out, int32_t* local_idx,
                               void* workspace, int64_t workspace_bytes, ogl_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_dst < 0 || fanout < 0 || !n_src_out) return OGL_EINVAL;
  if (n_dst == 0) { OGL_CHECK_HIP(hipMemsetAsync(n_src_out, 0, sizeof(int64_t), stream)); return OGL_OK; }
  if (!dst || !src_ids || (fanout > 0 && (!picks || !local_idx))) return OGL_EINVAL;
  const int64_t P = n_dst * (1 + (int64_t)fanout);
  if (P >= ((int64_t)1 << 30)) return OGL_EINVAL;
  if (!workspace || workspace_bytes < ws_bytes(P)) return OGL_EWORKSPACE;
  block_ws ws;
  ws.T = table_size(P); ws.P = P; ws.NB = ogl_cdiv(P, BLK_SCAN);
  int32_t* base = (int32_t*)workspace;
  ws.tkey = base; ws.tmin = base + ws.T; ws.tlidx = base + 2 * ws.T; ws.slot = base + 3 * ws.T;
  ws.bsum = ws.slot + ogl_round_up(P, 4);
  int logT = 0; while (((int64_t)1 << logT) < ws.T) ++logT;
  launch_block_reset(ws.tkey, ws.T, stream);                                       // tkey = -1, tmin = 0x7F7F7F7F > P
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_block_insert, dim3((unsigned)ogl_cdiv(P, 256)), dim3(256), 0, stream, dst, picks,
                     n_dst, P, ws.tkey, ws.tmin, ws.slot, (uint32_t)(ws.T - 1), 32 - logT);
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_block_count, dim3((unsigned)ws.NB), dim3(BLK_SCAN), 0, stream, ws.tmin, ws.slot,
                     n_dst, P, ws.bsum);
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_block_scan_sums, dim3(1), dim3(BLK_SCAN), 0, stream, ws.bsum, ws.NB, n_src_out);
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_block_assign, dim3((unsigned)ws.NB), dim3(BLK_SCAN), 0, stream, dst, picks, ws.tmin,
                     ws.slot, ws.bsum, n_dst, P, src_ids, ws.tlidx);
  OGL_CHECK_LAUNCH();
  if (fanout > 0) {
    hipLaunchKernelGGL(k_block_lookup, dim3((unsigned)ogl_cdiv(P - n_dst, 256)), dim3(256), 0, stream,
                       ws.slot, ws.tlidx, n_dst, P, local_idx);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
}


// ---- batched build: every batch of a loader layer in one set of launches ------------------------------------------------
// Same algorithm, same results as ogl_build_block per batch (bit-exact); blockIdx.y = batch.  All batches of a chunk
// share one table size (that of the largest), so the workspace is nb x (3 T) + slots + per-batch scan sums.
struct block_batch {
  ogl_batch_desc bd;
  int fanout;
  int64_t T;          // table entries per batch
  int64_t NBmax;      // scan blocks of the largest batch
  int32_t* tkey; int32_t* tmin; int32_t* tlidx;   // [nb][T]
  int32_t* slot;      // packed like src_ids: batch b at row_off[b] * (1 + fanout)
  int32_t* bsum;      // [nb][NBmax + 1]
  uint32_t mask; int shift;
};

__global__ void __launch_bounds__(256) k_block_insert_b(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks,
                                                        block_batch w) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t P = n_dst * (1 + (int64_t)w.fanout);
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int32_t* tkey = w.tkey + (int64_t)b * w.T; int32_t* tmin = w.tmin + (int64_t)b * w.T;
  int32_t* slot = w.slot + w.bd.row_off[b] * (1 + (int64_t)w.fanout);
  const int64_t id64 = flat_id(dst_base + w.bd.dst_start[b], picks + w.bd.row_off[b] * w.fanout, n_dst, p);
  if (id64 < 0) { slot[p] = -1; return; }
  int32_t id = (int32_t)id64;
  uint32_t h = ((uint32_t)id * 0x9E3779B1u) >> w.shift;
  for (uint32_t probe = 0; probe <= w.mask; ++probe) {
    int32_t old = atomicCAS(&tkey[h], -1, id);
    if (old == -1 || old == id) {
      atomicMin(&tmin[h], (int32_t)p);
      slot[p] = (int32_t)h;
      return;
    }
    h = (h + 1) & w.mask;
  }
  slot[p] = -1;
}

__global__ void __launch_bounds__(BLK_SCAN) k_block_count_b(block_batch w) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t P = n_dst * (1 + (int64_t)w.fanout);
  if ((int64_t)blockIdx.x * BLK_SCAN >= P) return;               // block-uniform
  int64_t p = (int64_t)blockIdx.x * BLK_SCAN + threadIdx.x;
  int tot;
  (void)block_scan_1024(is_first(w.tmin + (int64_t)b * w.T, w.slot + w.bd.row_off[b] * (1 + (int64_t)w.fanout), n_dst, P, p), &tot);
  if (threadIdx.x == 0) w.bsum[(int64_t)b * (w.NBmax + 1) + blockIdx.x] = tot;
}

// one block per batch: exclusive scan of its bsum[0..NB) in place, bsum[NB] = total = n_src_out[b]
__global__ void __launch_bounds__(BLK_SCAN) k_block_scan_sums_b(block_batch w, int64_t* __restrict__ n_src_out) {
  __shared__ int carry_s;
  __shared__ int part[BLK_SCAN];
  const int b = blockIdx.x;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t NB = ogl_cdiv_dev(n_dst * (1 + (int64_t)w.fanout), BLK_SCAN);
  int32_t* bsum = w.bsum + (int64_t)b * (w.NBmax + 1);
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < NB; base += BLK_SCAN) {
    int64_t i = base + threadIdx.x;
    int v = i < NB ? bsum[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < BLK_SCAN; off <<= 1) {
      int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    int incl = part[threadIdx.x];
    int carry = carry_s;
    if (i < NB) bsum[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == BLK_SCAN - 1) carry_s = carry + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) { bsum[NB] = carry_s; n_src_out[b] = (int64_t)carry_s; }
}

__global__ void __launch_bounds__(BLK_SCAN) k_block_assign_b(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks,
                                                             block_batch w, int64_t* __restrict__ src_ids) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t P = n_dst * (1 + (int64_t)w.fanout);
  if ((int64_t)blockIdx.x * BLK_SCAN >= P) return;               // block-uniform
  const int32_t* tmin = w.tmin + (int64_t)b * w.T;
  const int32_t* slot = w.slot + w.bd.row_off[b] * (1 + (int64_t)w.fanout);
  int64_t p = (int64_t)blockIdx.x * BLK_SCAN + threadIdx.x;
  int f = is_first(tmin, slot, n_dst, P, p);
  int tot;
  int pre = block_scan_1024(f, &tot);
  if (f) {
    int32_t li = w.bsum[(int64_t)b * (w.NBmax + 1) + blockIdx.x] + pre;
    src_ids[w.bd.row_off[b] * (1 + (int64_t)w.fanout) + li] =
        flat_id(dst_base + w.bd.dst_start[b], picks + w.bd.row_off[b] * w.fanout, n_dst, p);
    int32_t s = slot[p];
    if (s >= 0 && tmin[s] == (int32_t)p) w.tlidx[(int64_t)b * w.T + s] = li;
  }
}

__global__ void __launch_bounds__(256) k_block_lookup_b(block_batch w, int32_t* __restrict__ local_idx) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_dst * w.fanout) return;
  int32_t s = w.slot[w.bd.row_off[b] * (1 + (int64_t)w.fanout) + n_dst + e];
  local_idx[w.bd.row_off[b] * w.fanout + e] = s >= 0 ? w.tlidx[(int64_t)b * w.T + s] : -1;
}

static int64_t batched_ws_bytes(const int64_t* dst_count, int nb, int fanout) {
  // chunks of OGL_MAX_BATCH reuse the same workspace: size it for the worst chunk
  int64_t worst = 16;
  for (int b0 = 0; b0 < nb; b0 += OGL_MAX_BATCH) {
    const int m = nb - b0 < OGL_MAX_BATCH ? nb - b0 : OGL_MAX_BATCH;
    int64_t mx = 0, rows = 0;
    for (int b = 0; b < m; ++b) { mx = mx > dst_count[b0 + b] ? mx : dst_count[b0 + b]; rows += dst_count[b0 + b]; }
    const int64_t Pmax = mx * (1 + (int64_t)fanout);
    const int64_t T = table_size(Pmax), NBmax = ogl_cdiv(Pmax, BLK_SCAN);
    const int64_t bytes = 4 * (3 * T * m + ogl_round_up(rows * (1 + (int64_t)fanout), 4) + ogl_round_up((NBmax + 1) * m, 4));
    worst = worst > bytes ? worst : bytes;
  }
  return worst;
}

static int64_t block_workspace_bytes_hash(const int64_t* dst_count, int nb, int fanout) {
  if (nb < 0 || fanout < 0 || (nb > 0 && !dst_count)) return OGL_EINVAL;
  for (int b = 0; b < nb; ++b) if (dst_count[b] < 0) return OGL_EINVAL;
  return batched_ws_bytes(dst_count, nb, fanout);
}

static int build_block_hash(const int64_t* dst_base, const int64_t* dst_start, const int64_t* dst_count, int nb,
                            const int64_t* picks, int fanout, int64_t* src_ids, int64_t* n_src_out,
                            int32_t* local_idx, void* workspace, int64_t workspace_bytes, ogl_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (nb < 0 || fanout < 0) return OGL_EINVAL;
  if (nb == 0) return OGL_OK;
  if (!dst_start || !dst_count || !n_src_out) return OGL_EINVAL;
  for (int b = 0; b < nb; ++b) if (dst_count[b] < 0 || dst_start[b] < 0) return OGL_EINVAL;
  if (!workspace || workspace_bytes < batched_ws_bytes(dst_count, nb, fanout)) return OGL_EWORKSPACE;
  int64_t row = 0;
  for (int b0 = 0; b0 < nb; b0 += OGL_MAX_BATCH) {
    const int m = nb - b0 < OGL_MAX_BATCH ? nb - b0 : OGL_MAX_BATCH;
    block_batch w;
    int64_t mx = 0, rows = 0;
    const int64_t row0 = row;
    for (int b = 0; b < m; ++b) {
      w.bd.dst_start[b] = dst_start[b0 + b]; w.bd.row_off[b] = row; w.bd.ctr[b] = 0;
      row += dst_count[b0 + b]; rows += dst_count[b0 + b];
      mx = mx > dst_count[b0 + b] ? mx : dst_count[b0 + b];
    }
    w.bd.row_off[m] = row;
    if (mx == 0) { OGL_CHECK_HIP(hipMemsetAsync(n_src_out + b0, 0, sizeof(int64_t) * m, stream)); continue; }
    if (!dst_base || !src_ids || (fanout > 0 && (!picks || !local_idx))) return OGL_EINVAL;
    const int64_t Pmax = mx * (1 + (int64_t)fanout);
    if (Pmax >= ((int64_t)1 << 30)) return OGL_EINVAL;
    w.fanout = fanout; w.T = table_size(Pmax); w.NBmax = ogl_cdiv(Pmax, BLK_SCAN);
    int32_t* base = (int32_t*)workspace;
    w.tkey = base; w.tmin = base + w.T * m; w.tlidx = base + 2 * w.T * m;
    // slots are addressed by the GLOBAL packed row offset: rebase so that batch b0's rows start at the array's head
    int32_t* slot0 = base + 3 * w.T * m;
    w.slot = slot0 - row0 * (1 + (int64_t)fanout);
    w.bsum = slot0 + ogl_round_up(rows * (1 + (int64_t)fanout), 4);
    int logT = 0; while (((int64_t)1 << logT) < w.T) ++logT;
    w.mask = (uint32_t)(w.T - 1); w.shift = 32 - logT;
    launch_block_reset(w.tkey, w.T * m, stream);                                       // tkey = -1, tmin = 0x7F7F7F7F > P
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_insert_b, dim3((unsigned)ogl_cdiv(Pmax, 256), (unsigned)m), dim3(256), 0, stream, dst_base, picks, w);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_count_b, dim3((unsigned)w.NBmax, (unsigned)m), dim3(BLK_SCAN), 0, stream, w);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_scan_sums_b, dim3((unsigned)m), dim3(BLK_SCAN), 0, stream, w, n_src_out + b0);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_assign_b, dim3((unsigned)w.NBmax, (unsigned)m), dim3(BLK_SCAN), 0, stream, dst_base, picks, w, src_ids);
    OGL_CHECK_LAUNCH();
    if (fanout > 0) {
      hipLaunchKernelGGL(k_block_lookup_b, dim3((unsigned)ogl_cdiv(mx * fanout, 256), (unsigned)m), dim3(256), 0, stream, w, local_idx);
      OGL_CHECK_LAUNCH();
    }
  }
  return OGL_OK;
}

// ---- the same build with a DIRECT-ADDRESS table (the caller knows that every id is < n_ids) --------------------------------
// Per batch one int32 per vertex id instead of a 2 P-slot hash: tmin[id] = the smallest flat position at which the id appears.
// One no-return atomicMin per position (the hash form needs an atomicCAS whose result it waits for, then the atomicMin, then
// probes), no slot array, and the first-appearance tests of the count / assign passes gather from a table of n_ids entries
// instead of 4 P.  Same first-appearance order, hence the same src_ids / local_idx bit for bit.  block_batch is reused with
// T = n_ids (tkey / slot unused).
__global__ void __launch_bounds__(256) k_block_fill_d(int4* __restrict__ t, int64_t n16) {
  const int4 v = make_int4(0x7F7F7F7F, 0x7F7F7F7F, 0x7F7F7F7F, 0x7F7F7F7F);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) t[i] = v;
}

__device__ __forceinline__ int32_t direct_id(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks, const block_batch& w,
                                             int b, int64_t n_dst, int64_t p) {
  const int64_t id = flat_id(dst_base + w.bd.dst_start[b], picks + w.bd.row_off[b] * w.fanout, n_dst, p);
  return (id >= 0 && id < w.T) ? (int32_t)id : -1;
}

__global__ void __launch_bounds__(256) k_block_min_d(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks, block_batch w) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t P = n_dst * (1 + (int64_t)w.fanout);
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int32_t id = direct_id(dst_base, picks, w, b, n_dst, p);
  // (a plain load in front — "skip the atomic when an earlier position already recorded a smaller one", two thirds of the positions —
  // makes the launch SLOWER: 0.41 -> 0.9-1.0 ms for 9.2 M positions, whether the load is plain, volatile or non-temporal: loads
  // and atomics to the same lines serialise in L2.  The launch runs at the part's random-atomic rate, ~22 G/s.)
  if (id >= 0) atomicMin(&w.tmin[(int64_t)b * w.T + id], (int32_t)p);       // (result unused: a no-return atomic)
}

// The same minima WITHOUT global atomics: the id range is cut into R pieces of RL ids that fit LDS; workgroup (batch, piece) streams
// ALL positions of its batch and keeps the minima of the ids in its piece in LDS (ds_min), then stores its piece of tmin — which
// also makes the fill launch unnecessary.  Global atomics execute at the memory side at ~22 G requests/s whatever their form
// (MI355X_MICROARCH.md, Global float atomics: the same holds for integer ones here): 9.2 M of them are 0.41 ms; R re-reads of
// the ids (8 pieces of a batch share an XCD under round-robin placement — speed only — so its L2 serves 7 of them) are ~0.06 ms.
#define BLK_LDS_MAX_ENTRIES 36864          // 144 KB of LDS
#define BLK_LDS_MAX_PIECES 16
__global__ void __launch_bounds__(1024) k_block_min_lds(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks, block_batch w,
                                                        int m, int R, int RL) {
  extern __shared__ int32_t tab[];
  const int L = blockIdx.x;
  const int q = L >> 3;
  const int r = q % R;
  const int b = (q / R) * 8 + (L & 7);
  if (b >= m) return;                                            // block-uniform
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t P = n_dst * (1 + (int64_t)w.fanout);
  const int64_t lo = (int64_t)r * RL;
  const int len = (int)(lo + RL <= w.T ? RL : (w.T > lo ? w.T - lo : 0));
  for (int i = threadIdx.x; i < len; i += 1024) tab[i] = 0x7F7F7F7F;
  __syncthreads();
  const int64_t* dst = dst_base + w.bd.dst_start[b];
  const int64_t* pk = picks + w.bd.row_off[b] * w.fanout;
  for (int64_t p0 = 0; p0 < P; p0 += 4 * 1024) {
    int64_t id[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0 + u * 1024 + threadIdx.x;
      id[u] = p < P ? flat_id(dst, pk, n_dst, p) : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t k = id[u] - lo;
      if (k >= 0 && k < len) atomicMin(&tab[k], (int32_t)(p0 + u * 1024 + threadIdx.x));
    }
  }
  __syncthreads();
  int32_t* out = w.tmin + (int64_t)b * w.T + lo;
  for (int i = threadIdx.x; i < len; i += 1024) out[i] = tab[i];
}

__device__ __forceinline__ int is_first_d(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks, const block_batch& w,
                                          int b, int64_t n_dst, int64_t P, int64_t p, int32_t* id_out) {
  *id_out = -1;
  if (p >= P) return 0;
  const int32_t id = direct_id(dst_base, picks, w, b, n_dst, p);
  *id_out = id;
  if (p < n_dst) return 1;                                                  // every dst keeps its own source row
  return (id >= 0 && w.tmin[(int64_t)b * w.T + id] == (int32_t)p) ? 1 : 0;
}

__global__ void __launch_bounds__(BLK_SCAN) k_block_count_d(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks,
                                                            block_batch w) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t P = n_dst * (1 + (int64_t)w.fanout);
  if ((int64_t)blockIdx.x * BLK_SCAN >= P) return;               // block-uniform
  const int64_t p = (int64_t)blockIdx.x * BLK_SCAN + threadIdx.x;
  int tot; int32_t id;
  (void)block_scan_1024(is_first_d(dst_base, picks, w, b, n_dst, P, p, &id), &tot);
  if (threadIdx.x == 0) w.bsum[(int64_t)b * (w.NBmax + 1) + blockIdx.x] = tot;
}

__global__ void __launch_bounds__(BLK_SCAN) k_block_assign_d(const int64_t* __restrict__ dst_base, const int64_t* __restrict__ picks,
                                                             block_batch w, int64_t* __restrict__ src_ids) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t P = n_dst * (1 + (int64_t)w.fanout);
  if ((int64_t)blockIdx.x * BLK_SCAN >= P) return;               // block-uniform
  const int64_t p = (int64_t)blockIdx.x * BLK_SCAN + threadIdx.x;
  int32_t id;
  const int f = is_first_d(dst_base, picks, w, b, n_dst, P, p, &id);
  int tot;
  const int pre = block_scan_1024(f, &tot);
  if (f) {
    const int32_t li = w.bsum[(int64_t)b * (w.NBmax + 1) + blockIdx.x] + pre;
    src_ids[w.bd.row_off[b] * (1 + (int64_t)w.fanout) + li] =
        flat_id(dst_base + w.bd.dst_start[b], picks + w.bd.row_off[b] * w.fanout, n_dst, p);
    if (id >= 0 && w.tmin[(int64_t)b * w.T + id] == (int32_t)p) w.tlidx[(int64_t)b * w.T + id] = li;
  }
}

__global__ void __launch_bounds__(256) k_block_lookup_d(const int64_t* __restrict__ picks, block_batch w, int32_t* __restrict__ local_idx) {
  const int b = blockIdx.y;
  const int64_t n_dst = w.bd.row_off[b + 1] - w.bd.row_off[b];
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_dst * w.fanout) return;
  const int64_t id = picks[w.bd.row_off[b] * w.fanout + e];
  local_idx[w.bd.row_off[b] * w.fanout + e] = (id >= 0 && id < w.T) ? w.tlidx[(int64_t)b * w.T + id] : -1;
}

// (diagnostic switch: 0 = the minima through global atomics even where the LDS form applies — tests compare the two)
static int g_block_min_lds = 1;
int oglx_knob_block_min_lds(int on, int* prev) { *prev = g_block_min_lds; g_block_min_lds = on ? 1 : 0; return OGL_OK; }   // (ogl_debug_set)

static int64_t batched_ws_bytes_ids(const int64_t* dst_count, int nb, int fanout, int64_t n_ids) {
  int64_t worst = 16;
  const int64_t Tn = ogl_round_up(n_ids, 4);
  for (int b0 = 0; b0 < nb; b0 += OGL_MAX_BATCH) {
    const int m = nb - b0 < OGL_MAX_BATCH ? nb - b0 : OGL_MAX_BATCH;
    int64_t mx = 0;
    for (int b = 0; b < m; ++b) mx = mx > dst_count[b0 + b] ? mx : dst_count[b0 + b];
    const int64_t NBmax = ogl_cdiv(mx * (1 + (int64_t)fanout), BLK_SCAN);
    const int64_t bytes = 4 * (2 * Tn * m + ogl_round_up((NBmax + 1) * m, 4));
    worst = worst > bytes ? worst : bytes;
  }
  return worst;
}

extern "C" int64_t ogl_block_workspace_bytes_batched(const int64_t* dst_count, int nb, int fanout, int64_t n_ids) {
  if (n_ids <= 0) return block_workspace_bytes_hash(dst_count, nb, fanout);       // (no bound on the ids: the hash-table form)
  if (nb < 0 || fanout < 0 || n_ids >= ((int64_t)1 << 31) || (nb > 0 && !dst_count)) return OGL_EINVAL;
  for (int b = 0; b < nb; ++b) if (dst_count[b] < 0) return OGL_EINVAL;
  return batched_ws_bytes_ids(dst_count, nb, fanout, n_ids);
}

extern "C" int ogl_build_block_batched(const int64_t* dst_base, const int64_t* dst_start, const int64_t* dst_count, int nb,
                                       const int64_t* picks, int fanout, int64_t n_ids, int64_t* src_ids, int64_t* n_src_out,
                                       int32_t* local_idx, void* workspace, int64_t workspace_bytes, ogl_stream_t stream_) {
  if (n_ids <= 0)                                                 // (no bound on the ids: the hash-table form)
    return build_block_hash(dst_base, dst_start, dst_count, nb, picks, fanout, src_ids, n_src_out, local_idx, workspace, workspace_bytes, stream_);
  hipStream_t stream = (hipStream_t)stream_;
  if (nb < 0 || fanout < 0 || n_ids >= ((int64_t)1 << 31)) return OGL_EINVAL;
  if (nb == 0) return OGL_OK;
  if (!dst_start || !dst_count || !n_src_out) return OGL_EINVAL;
  for (int b = 0; b < nb; ++b) if (dst_count[b] < 0 || dst_start[b] < 0) return OGL_EINVAL;
  if (!workspace || workspace_bytes < batched_ws_bytes_ids(dst_count, nb, fanout, n_ids)) return OGL_EWORKSPACE;
  const int64_t Tn = ogl_round_up(n_ids, 4);
  int64_t row = 0;
  for (int b0 = 0; b0 < nb; b0 += OGL_MAX_BATCH) {
    const int m = nb - b0 < OGL_MAX_BATCH ? nb - b0 : OGL_MAX_BATCH;
    block_batch w;
    int64_t mx = 0;
    for (int b = 0; b < m; ++b) {
      w.bd.dst_start[b] = dst_start[b0 + b]; w.bd.row_off[b] = row; w.bd.ctr[b] = 0;
      row += dst_count[b0 + b];
      mx = mx > dst_count[b0 + b] ? mx : dst_count[b0 + b];
    }
    w.bd.row_off[m] = row;
    if (mx == 0) { OGL_CHECK_HIP(hipMemsetAsync(n_src_out + b0, 0, sizeof(int64_t) * m, stream)); continue; }
    if (!dst_base || !src_ids || (fanout > 0 && (!picks || !local_idx))) return OGL_EINVAL;
    const int64_t Pmax = mx * (1 + (int64_t)fanout);
    if (Pmax >= ((int64_t)1 << 30)) return OGL_EINVAL;
    w.fanout = fanout; w.T = Tn; w.NBmax = ogl_cdiv(Pmax, BLK_SCAN);
    int32_t* base = (int32_t*)workspace;
    w.tkey = nullptr; w.slot = nullptr; w.mask = 0; w.shift = 0;
    w.tmin = base; w.tlidx = base + Tn * m; w.bsum = base + 2 * Tn * m;
    const int R = (int)ogl_cdiv(Tn, BLK_LDS_MAX_ENTRIES);
    if (R <= BLK_LDS_MAX_PIECES && g_block_min_lds) {
      const int RL = (int)ogl_round_up(ogl_cdiv(Tn, R), 4);
      static bool attr_set = false;
      if (!attr_set) {
        OGL_CHECK_HIP(hipFuncSetAttribute((const void*)k_block_min_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
      }
      hipLaunchKernelGGL(k_block_min_lds, dim3((unsigned)(8 * ogl_cdiv(m, 8) * R)), dim3(1024), (size_t)RL * 4, stream, dst_base, picks, w, m, R, RL);
      OGL_CHECK_LAUNCH();
    } else {
      const int64_t n16 = Tn * m / 4;
      hipLaunchKernelGGL(k_block_fill_d, dim3((unsigned)std::min<int64_t>(ogl_cdiv(n16, 256), 4096)), dim3(256), 0, stream, (int4*)w.tmin, n16);
      OGL_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_block_min_d, dim3((unsigned)ogl_cdiv(Pmax, 256), (unsigned)m), dim3(256), 0, stream, dst_base, picks, w);
      OGL_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_block_count_d, dim3((unsigned)w.NBmax, (unsigned)m), dim3(BLK_SCAN), 0, stream, dst_base, picks, w);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_scan_sums_b, dim3((unsigned)m), dim3(BLK_SCAN), 0, stream, w, n_src_out + b0);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_block_assign_d, dim3((unsigned)w.NBmax, (unsigned)m), dim3(BLK_SCAN), 0, stream, dst_base, picks, w, src_ids);
    OGL_CHECK_LAUNCH();
    if (fanout > 0) {
      hipLaunchKernelGGL(k_block_lookup_d, dim3((unsigned)ogl_cdiv(mx * fanout, 256), (unsigned)m), dim3(256), 0, stream, picks, w, local_idx);
      OGL_CHECK_LAUNCH();
    }
  }
  return OGL_OK;
}


// ======================================================================================================================
// The WHOLE sampling phase of a small batch in ONE launch (round 5): what a captured 32-seed step ran as eleven 4-5 us graph nodes —
// stage [counter | seeds] from mapped host memory, sample the output block, relabel it, sample the input block for the sources found,
// relabel it, publish the two source counts to the host — is one 1024-thread workgroup walking the same phases with
// __syncthreads() between them (60 -> ~25 us of a 216 us pubmed-rung step; R/train/graphsage/pytorch/model.py:76-117 runs this
// as DGL's NodeDataLoader + two to_block calls per batch on the host).
// Same draws as ogl_sample_layer_dev (Philox key = seed, counter = (quad | layer << 16, vertex id, batch counter)), same blocks as
// ogl_build_block_padded on the same shapes: the output block over the B seeds (sources padded with -1 up to n1_cap = B (1 + S)), the
// input block over ALL n1_cap rows of that source list (padded destinations sample nothing: their index rows are -1, their own rows
// stay in the source list), sources padded with -1 up to n0_cap = n1_cap (1 + S).  Positions that cannot hold a vertex (the picks of
// padded destinations) are never visited, the hash table is sized by the positions that can: first-appearance order is unchanged.
struct SmallSampleArgs {
  const int64_t* indptr; const int32_t* indices; const int32_t* deg; int64_t n;
  const int64_t* head_host;               // mapped pinned memory [1 + B]: Philox batch counter | seeds
  int64_t* head_dev;                      // [1 + B]: the static buffer the train graph reads
  int B, S;
  uint32_t seed_lo, seed_hi;
  int64_t* picks1; int64_t* picks0;       // workspace: [B, S], [n1_cap, S]
  int64_t* src1; int32_t* lidx1;          // [n1_cap], [B, S]
  int64_t* src0; int32_t* lidx0;          // [n0_cap], [n1_cap, S]
  int32_t* table;                         // workspace: tkey | tmin | tlidx (T_max each) | slot (n1_cap (1 + S))
  int64_t T_max;
  int64_t* counts;                        // device [2]: n1, n0
  int64_t* seq_dev; volatile int64_t* counts_host;   // host-mapped [3]: n1, n0, sequence number (ogl_publish_i64's protocol)
  int64_t fill0;                          // 0: src0 is padded with -1 up to its capacity; m > 0: up to round_up(n0, m) only
  int reg_build;                          // 1: the relabelling with its positions in registers where it applies (wg_build_reg)
};

// one layer's picks for rows [0, n_live) of dst (one thread per (row, 4 slots): k_sample_layer_dev's arithmetic).  TWO items per
// thread and trip, their loads issued together: an item is a chain of two dependent cold loads (degree / row offset, then the picked
// neighbours), and the input block of a 32-seed batch has more items than the workgroup has threads (228 live rows x 7 quads = 1 596):
// walked one per trip that is two chains in a row.
__device__ __forceinline__ void wg_sample(const SmallSampleArgs& a, const int64_t* __restrict__ dst, int64_t n_live, uint64_t ctr,
                                          uint32_t layer_bits, int64_t* __restrict__ picks) {
  const int quads = (a.S + 3) / 4;
  const int64_t total = n_live * quads;
  for (int64_t t0 = threadIdx.x; t0 < total; t0 += 2 * BLK_SCAN) {
    int64_t ti[2], d[2], base[2];
    uint32_t deg[2];
    int q[2];
    bool on[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t t = t0 + (int64_t)u * BLK_SCAN;
      on[u] = t < total;
      ti[u] = on[u] ? t / quads : 0;
      q[u] = on[u] ? (int)(t - ti[u] * quads) : 0;
      d[u] = on[u] ? dst[ti[u]] : -1;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      deg[u] = 0; base[u] = 0;
      if (on[u] && d[u] >= 0 && d[u] < a.n) { deg[u] = (uint32_t)a.deg[d[u]]; base[u] = a.indptr[d[u]]; }
    }
    int64_t pk[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j0 = q[u] * 4, cnt = min(4, a.S - j0);
      if (on[u] && deg[u] != 0) {
        const philox4 r = philox4x32_10((uint32_t)q[u] | layer_bits, (uint32_t)((uint64_t)d[u] & 0xFFFFFFFFu), (uint32_t)((uint64_t)d[u] >> 32),
                                        (uint32_t)(ctr & 0xFFFFFFFFu), a.seed_lo, a.seed_hi ^ (uint32_t)(ctr >> 32));
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pk[u][j] = j < cnt ? (int64_t)a.indices[base[u] + (uint32_t)(((uint64_t)w[j] * (uint64_t)deg[u]) >> 32)] : -1;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[u][j] = -1;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!on[u]) continue;
      const int j0 = q[u] * 4, cnt = min(4, a.S - j0);
      int64_t* out = picks + ti[u] * a.S + j0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < cnt) out[j] = pk[u][j];
    }
  }
}

// relabel one block inside the workgroup: destinations dst[0 .. n_dst) first (rows >= n_live hold -1 and have no picks), then the
// picks of rows [0, n_live) in row-major order; returns the source count.  Compact position q: q < n_dst the destination q, else pick
// q - n_dst — the same ORDER as the flat positions of ogl_build_block, which is all first-appearance relabelling depends on.
// The hash table lives in LDS while it fits (keys + minima, 2 x 64 KB for 16 384 entries: positions <= 10 922, i.e. up to ~400 real
// destinations at fanout 25): a global-memory table costs one L2 round trip per atomic, ~1.5 us per 1 024 positions and phase — the
// one-workgroup kernel then took the 60 us of the eleven launches it replaced.  Larger blocks keep the global table.
#define SS_LDS_T 16384
template <bool LDS_TABLE>
__device__ __forceinline__ int wg_build_t(const SmallSampleArgs& a, const int64_t* __restrict__ dst, int64_t n_dst, int64_t n_live,
                                          const int64_t* __restrict__ picks, int64_t* __restrict__ src_ids, int64_t src_cap,
                                          int32_t* __restrict__ local_idx, int* carry_s, int32_t* lkey, int32_t* lmin, int64_t T,
                                          int64_t fill_mult = 0) {
  const int tid = threadIdx.x;
  const int64_t Q = n_dst + n_live * a.S;
  int logT = 0;
  while (((int64_t)1 << logT) < T) ++logT;
  const uint32_t mask = (uint32_t)(T - 1);
  const int shift = 32 - logT;
  // (LDS form: `tmin` doubles as `tlidx` once the scan has read a slot's minimum — one thread owns each slot's minimum)
  int32_t* tkey = LDS_TABLE ? lkey : a.table;
  int32_t* tmin = LDS_TABLE ? lmin : a.table + a.T_max;
  int32_t* tlidx = LDS_TABLE ? lmin : a.table + 2 * a.T_max;
  int32_t* slot = a.table + 3 * a.T_max;
  for (int64_t i = tid; i < T; i += BLK_SCAN) { tkey[i] = -1; tmin[i] = 0x7F7F7F7F; }
  if (tid == 0) *carry_s = 0;
  __threadfence_block();
  __syncthreads();
  for (int64_t q = tid; q < Q; q += BLK_SCAN) {
    const int64_t id64 = q < n_dst ? dst[q] : picks[q - n_dst];
    int32_t sl = -1;
    if (id64 >= 0) {
      const int32_t id = (int32_t)id64;
      uint32_t h = ((uint32_t)id * 0x9E3779B1u) >> shift;
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        const int32_t old = atomicCAS(&tkey[h], -1, id);
        if (old == -1 || old == id) { atomicMin(&tmin[h], (int32_t)q); sl = (int32_t)h; break; }
        h = (h + 1) & mask;
      }
    }
    slot[q] = sl;
  }
  __threadfence_block();
  __syncthreads();
  for (int64_t base = 0; base < Q; base += BLK_SCAN) {
    const int64_t q = base + tid;
    const int32_t sp = q < Q ? slot[q] : -1;
    int32_t mn = -1;
    if (sp >= 0) mn = LDS_TABLE ? tmin[sp] : __hip_atomic_load(&tmin[sp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool own = sp >= 0 && mn == (int32_t)q;
    const int f = q < Q && (q < n_dst || own) ? 1 : 0;
    int tot;
    const int pre = block_scan_1024(f, &tot);               // (its barriers order every read of tmin above before the writes below)
    const int carry = *carry_s;
    if (f) {
      const int32_t li = carry + pre;
      src_ids[li] = q < n_dst ? dst[q] : picks[q - n_dst];
      // (LDS form: minima and local indices share one array, and later chunks still compare positions against the minima)
      if (own && !LDS_TABLE) tlidx[sp] = li;
      if (own && LDS_TABLE) slot[q] = -2 - li;               // (remembered per POSITION; applied to the table after the scan)
    }
    __syncthreads();
    if (tid == 0) *carry_s = carry + tot;
    __syncthreads();
  }
  const int n_src = *carry_s;
  if (LDS_TABLE) {
    // second pass of the owners: slot[q] = -2 - li marks "q owns its table slot, local index li"; re-probe for the slot (read-only)
    __syncthreads();
    for (int64_t q = tid; q < Q; q += BLK_SCAN) {
      const int32_t sv = slot[q];
      if (sv <= -2) {
        const int64_t id64 = q < n_dst ? dst[q] : picks[q - n_dst];
        const int32_t id = (int32_t)id64;
        uint32_t h = ((uint32_t)id * 0x9E3779B1u) >> shift;
        while (tkey[h] != id) h = (h + 1) & mask;
        tlidx[h] = -2 - sv;
        slot[q] = (int32_t)h;
      }
    }
  }
  // (fill_mult: the caller reads the source list up to round_up(n_src, fill_mult) only — the size bucket of the train graph it replays —,
  // so the -1 padding stops there: 21 632 entries per 32-seed step otherwise, for ~2 600 sources)
  const int64_t fill_end = fill_mult > 0 ? min(src_cap, (n_src + fill_mult - 1) / fill_mult * fill_mult) : src_cap;
  for (int64_t i = n_src + tid; i < fill_end; i += BLK_SCAN) src_ids[i] = -1;
  __threadfence_block();
  __syncthreads();
  for (int64_t e = tid; e < n_dst * a.S; e += BLK_SCAN) {
    int32_t v = -1;
    if (e < n_live * a.S) { const int32_t s = slot[n_dst + e]; v = s >= 0 ? tlidx[s] : -1; }
    local_idx[e] = v;
  }
  __syncthreads();
  return n_src;
}

// The same relabelling with every position in REGISTERS (LDS table, at most SS_KMAX positions per thread: every block of the 32-seed rungs):
// thread t owns the CONTIGUOUS positions [t K, (t + 1) K), K = ceil(Q / 1024) — so ONE block scan of the per-thread counts orders all
// first appearances (the chunked form above scans once per 1 024 positions, and because later chunks still compare positions against the
// table's minima it parks local ids in the global `slot` array and re-probes the table in a second pass: 2 writes + 3 reads of global
// memory per position and ceil(Q / 1024) x 4 barriers).  Here a position's id and table slot never leave registers, the minima are
// overwritten by the local ids right after the one scan (its barriers separate the last read of a minimum from the first write), and the
// index rows are written from the registers.  Same order, same arrays, bit for bit (tests/test_gpu_round5.py, test_gpu_graphs.py).
#define SS_KMAX 11
__device__ __forceinline__ int block_scan_counts_1024(int v, int* total) {
  __shared__ int wsum2[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum2[wid] = x;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int sv = wsum2[w];
    if (w < wid) woff += sv;
    tot += sv;
  }
  *total = tot;
  __syncthreads();
  return x - v + woff;
}

__device__ __forceinline__ int wg_build_reg(const SmallSampleArgs& a, const int64_t* __restrict__ dst, int64_t n_dst, int64_t n_live,
                                            const int64_t* __restrict__ picks, int64_t* __restrict__ src_ids, int64_t src_cap,
                                            int32_t* __restrict__ local_idx, int32_t* lkey, int32_t* lmin, int64_t T, int64_t fill_mult) {
  const int tid = threadIdx.x;
  const int Q = (int)(n_dst + n_live * a.S);
  const int K = (Q + BLK_SCAN - 1) / BLK_SCAN;               // <= SS_KMAX (the caller checks)
  const int q0 = tid * K;
  int logT = 0;
  while (((int64_t)1 << logT) < T) ++logT;
  const uint32_t mask = (uint32_t)(T - 1);
  const int shift = 32 - logT;
  for (int64_t i = tid; i < T; i += BLK_SCAN) { lkey[i] = -1; lmin[i] = 0x7F7F7F7F; }
  int64_t id[SS_KMAX];
  int sl[SS_KMAX];
#pragma unroll
  for (int r = 0; r < SS_KMAX; ++r) {
    const int q = q0 + r;
    id[r] = (r < K && q < Q) ? (q < n_dst ? dst[q] : picks[q - n_dst]) : -1;
    sl[r] = -1;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SS_KMAX; ++r) {
    if (r < K && id[r] >= 0) {
      const int32_t v = (int32_t)id[r];
      uint32_t h = ((uint32_t)v * 0x9E3779B1u) >> shift;
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        const int32_t old = atomicCAS(&lkey[h], -1, v);
        if (old == -1 || old == v) { atomicMin(&lmin[h], (int32_t)(q0 + r)); sl[r] = (int)h; break; }
        h = (h + 1) & mask;
      }
    }
  }
  __syncthreads();
  unsigned fl = 0, ow = 0;
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < SS_KMAX; ++r) {
    const int q = q0 + r;
    const bool valid = r < K && q < Q;
    const bool own = valid && sl[r] >= 0 && lmin[sl[r]] == q;
    const bool f = valid && (q < n_dst || own);
    fl |= (f ? 1u : 0u) << r;
    ow |= (own ? 1u : 0u) << r;
    cnt += f ? 1 : 0;
  }
  int tot;
  int li = block_scan_counts_1024(cnt, &tot);               // (its barriers: every read of a minimum above, before any write below)
#pragma unroll
  for (int r = 0; r < SS_KMAX; ++r) {
    if ((fl >> r) & 1u) {
      src_ids[li] = id[r];                                   // (a padded destination keeps its row in the source list: id -1)
      if ((ow >> r) & 1u) lmin[sl[r]] = li;                  // the table's minimum becomes the local id of its key
      ++li;
    }
  }
  const int n_src = tot;
  const int64_t fill_end = fill_mult > 0 ? min(src_cap, ((int64_t)n_src + fill_mult - 1) / fill_mult * fill_mult) : src_cap;
  for (int64_t i = n_src + tid; i < fill_end; i += BLK_SCAN) src_ids[i] = -1;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SS_KMAX; ++r) {
    const int q = q0 + r;
    if (r < K && q < Q && q >= n_dst) local_idx[q - n_dst] = sl[r] >= 0 ? lmin[sl[r]] : -1;
  }
  for (int64_t e = n_live * a.S + tid; e < n_dst * a.S; e += BLK_SCAN) local_idx[e] = -1;     // the padded destinations' rows
  __syncthreads();
  return n_src;
}

__device__ __forceinline__ int wg_build(const SmallSampleArgs& a, const int64_t* __restrict__ dst, int64_t n_dst, int64_t n_live,
                                        const int64_t* __restrict__ picks, int64_t* __restrict__ src_ids, int64_t src_cap,
                                        int32_t* __restrict__ local_idx, int* carry_s, int32_t* lkey, int32_t* lmin, int64_t fill_mult = 0) {
  const int64_t Q = n_dst + n_live * a.S;
  int64_t T = 1024;
  while (2 * T < 3 * Q && T < a.T_max) T <<= 1;            // load factor <= 2/3
  if (T <= SS_LDS_T && Q <= (int64_t)SS_KMAX * BLK_SCAN && a.reg_build)
    return wg_build_reg(a, dst, n_dst, n_live, picks, src_ids, src_cap, local_idx, lkey, lmin, T, fill_mult);
  if (T <= SS_LDS_T) return wg_build_t<true>(a, dst, n_dst, n_live, picks, src_ids, src_cap, local_idx, carry_s, lkey, lmin, T, fill_mult);
  while (T < 2 * Q && T < a.T_max) T <<= 1;
  return wg_build_t<false>(a, dst, n_dst, n_live, picks, src_ids, src_cap, local_idx, carry_s, lkey, lmin, T, fill_mult);
}

__global__ void __launch_bounds__(BLK_SCAN) k_sample_blocks_small(SmallSampleArgs a) {
  __shared__ int carry_s;
  __shared__ int32_t lkey[SS_LDS_T], lmin[SS_LDS_T];
  const int tid = threadIdx.x;
  // [counter | seeds] from the mapped host buffer into the static device buffer (the train graph gathers its labels by these seeds)
  if (tid < 1 + a.B) a.head_dev[tid] = a.head_host[tid];
  __threadfence_block();
  __syncthreads();
  const uint64_t ctr = (uint64_t)a.head_dev[0];
  const int64_t* seeds = a.head_dev + 1;
  const int64_t n1_cap = (int64_t)a.B * (1 + a.S), n0_cap = n1_cap * (1 + a.S);
  // ---- the output block: every seed is live
  wg_sample(a, seeds, a.B, ctr, 1u << 16, a.picks1);
  __threadfence_block();
  __syncthreads();
  const int n1 = wg_build(a, seeds, a.B, a.B, a.picks1, a.src1, n1_cap, a.lidx1, &carry_s, lkey, lmin);
  // ---- the input block: ALL n1_cap rows of src1 are its destinations, the first n1 of them real
  wg_sample(a, a.src1, n1, ctr, 0u, a.picks0);
  __threadfence_block();
  __syncthreads();
  const int n0 = wg_build(a, a.src1, n1_cap, n1, a.picks0, a.src0, n0_cap, a.lidx0, &carry_s, lkey, lmin, a.fill0);
  if (tid == 0) {
    a.counts[0] = n1; a.counts[1] = n0;
    a.counts_host[0] = n1; a.counts_host[1] = n0;
    const int64_t s = *a.seq_dev + 1;
    *a.seq_dev = s;
    __threadfence_system();
    a.counts_host[2] = s;
    __threadfence_system();
  }
}

extern "C" int64_t ogl_sample_blocks_small_workspace_bytes(int B, int fanout) {
  if (B <= 0 || fanout <= 0) return OGL_EINVAL;
  const int64_t n1_cap = (int64_t)B * (1 + fanout), P0 = n1_cap * (1 + fanout);
  if (P0 > BLK_SMALL_MAX_P) return OGL_EINVAL;
  const int64_t T = table_size(P0);
  return 8 * ((int64_t)B * fanout + n1_cap * fanout) + 4 * (3 * T + ogl_round_up(P0, 4)) + 64;
}

extern "C" int ogl_sample_blocks_small_fill(const ogl_graph_t* g, const int64_t* head_host_mapped, int64_t* head_dev, int B, int fanout,
                                            uint64_t seed, int64_t* src1, int32_t* lidx1, int64_t* src0, int32_t* lidx0, int64_t* counts,
                                            int64_t* seq_dev, int64_t* counts_host_mapped, void* workspace, int64_t workspace_bytes,
                                            int64_t src0_fill_multiple, ogl_stream_t stream) {
  if (!g || B <= 0 || fanout <= 0 || B > 1023 || src0_fill_multiple < 0) return OGL_EINVAL;
  const int64_t need = ogl_sample_blocks_small_workspace_bytes(B, fanout);
  if (need < 0) return OGL_EINVAL;
  if (!head_host_mapped || !head_dev || !src1 || !lidx1 || !src0 || !lidx0 || !counts || !seq_dev || !counts_host_mapped) return OGL_EINVAL;
  if (!workspace || ((uintptr_t)workspace & 7) || workspace_bytes < need) return OGL_EWORKSPACE;
  const int64_t n1_cap = (int64_t)B * (1 + fanout), P0 = n1_cap * (1 + fanout);
  SmallSampleArgs a;
  a.indptr = g->indptr; a.indices = g->indices; a.deg = g->deg; a.n = g->n;
  a.head_host = head_host_mapped; a.head_dev = head_dev; a.B = B; a.S = fanout;
  a.seed_lo = (uint32_t)(seed & 0xFFFFFFFFu); a.seed_hi = (uint32_t)(seed >> 32);
  a.picks1 = (int64_t*)workspace; a.picks0 = a.picks1 + (int64_t)B * fanout;
  a.table = (int32_t*)(a.picks0 + n1_cap * fanout);
  a.T_max = table_size(P0);
  a.src1 = src1; a.lidx1 = lidx1; a.src0 = src0; a.lidx0 = lidx0;
  a.counts = counts; a.seq_dev = seq_dev; a.counts_host = (volatile int64_t*)counts_host_mapped;
  a.fill0 = src0_fill_multiple;
  a.reg_build = 1;
  hipLaunchKernelGGL(k_sample_blocks_small, dim3(1), dim3(BLK_SCAN), 0, (hipStream_t)stream, a);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
