// Segmented backward of the fixed-fanout mean / sum neighbour reduction.
//
//     out[d, :] = (1 / S) sum_j src[idx[d, j], :]          (mailbox.mean(axis=1) / .sum(axis=1), R/train/graphsage/pytorch/
//                                                            aggregator_dgl.py:158,165,185; the autograd of it is what this replaces)
//     dsrc[s, :] = (1 / S) sum_{(d, j): idx[d, j] = s} dout[d, :]
//
// Round 1 scattered dout along the edges with E x D float atomics (Reddit meanpool layer 0: 176 k edges x 600 = 106 M atomics at the
// ~1.3 TB/s atomic rate: 0.3 ms, behind a 150 MB zero fill, in an order that changes from run to run).  Here the edges are sorted by
// SOURCE once per block (a plan built from the indices alone: it can run beside the forward pass) and the backward is a gather:
//
//   plan   k_seg_count     cnt[s] = #edges into s                         (integer atomics: counts do not depend on their order)
//          k_seg_bsum / k_seg_scan_sums / k_seg_apply   start[s] = exclusive prefix of cnt (three-phase scan), cursors zeroed
//          k_seg_place     every edge takes a place in its source's range (atomic cursor: the order INSIDE a range is arbitrary)
//          k_seg_sort_chunks / k_seg_rank   ... and is moved to its rank among the range's edge ids (ranges of more than 64 entries —
//                          hub sources — sorted in chunks through LDS, shorter ones by a walk): sorted[] lists every source's edges in edge
//                          order — the same list whatever order the atomics ran in, so the sums below are reproducible
//   apply  k_seg_reduce    one block per TILE of 64 consecutive entries of the sorted list (load-balanced: a hub referenced by a
//                          thousand destinations is spread over many tiles, no serial tail): the 64 gradient rows are gathered 8 at
//                          a time (a thread owns one 16-byte column chunk), runs of equal source are summed in list order; a run
//                          that lies inside the tile is finished on the spot, the (at most two) runs that cross a tile boundary
//                          leave partial rows
//          k_seg_fixup     one wave per source: a source whose range spans several tiles sums its partial rows in tile order; a
//                          source nobody sampled gets its zero row
// A finished row is scaled by 1 / S (mean), optionally masked by [relu_out[s, :] > 0] (the ReLU in front of a pooling mean:
// 'meanpool', aggregator_dgl.py:181-185) and written as fp32 and / or as the row-major bf16x3 image the weight-gradient product reads
// (ogl_linear_bwd_weight_x3k, dy_rows): for the first layer (no input gradient) dsrc itself is never materialised.
// HBM-bound: E rows of 4 D bytes gathered (from an [n_dst, D] matrix that mostly stays in L2 / MALL) + n_src rows written.
#include <algorithm>
#include "ogl_common.h"
#include "x6_arith.h"

#define SG_TILE 64
#define SG_TILE_SMALL 16     // blocks of at most SG_SMALL_E entries: a 64-entry tile is eight dependent batches of row loads in 200 blocks
#define SG_SMALL_E 32768     //   (31 us for the 12 800 edges of a 512-seed output block); 16-entry tiles are two batches in 800 blocks
static inline int seg_tile(int64_t E) { return E <= SG_SMALL_E ? SG_TILE_SMALL : SG_TILE; }
#define SG_THREADS 256
#define SG_U 8
#define SG_SCAN 1024

__global__ void __launch_bounds__(256) k_seg_count(const int32_t* __restrict__ idx, int64_t E, int64_t n_src, int* __restrict__ cnt) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int s = idx[e];
    if (s >= 0 && s < n_src) atomicAdd(&cnt[s], 1);
  }
}

// block b: sum of cnt[b * SG_SCAN ..)
__global__ void __launch_bounds__(SG_SCAN) k_seg_bsum(const int* __restrict__ cnt, int64_t n, int* __restrict__ bsum) {
  __shared__ int part[SG_SCAN / 64];
  const int64_t i = (int64_t)blockIdx.x * SG_SCAN + threadIdx.x;
  int v = i < n ? cnt[i] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < SG_SCAN / 64; ++w) t += part[w];
    bsum[blockIdx.x] = t;
  }
}

// exclusive scan of the block sums in place (one block; NB <= a few thousand)
#define SR_LONG 64        // ranges up to this many entries are ranked by k_seg_rank's quadratic walk, longer ones sorted through LDS
#define SR_CHUNK 4096

__global__ void __launch_bounds__(SG_SCAN) k_seg_scan_sums(int* __restrict__ bsum, int NB, int* __restrict__ nlong) {
  __shared__ int buf[SG_SCAN];
  __shared__ int carry;
  if (threadIdx.x == 0) { carry = 0; *nlong = 0; }            // (the long-source list k_seg_apply fills next)
  __syncthreads();
  for (int base = 0; base < NB; base += SG_SCAN) {
    const int i = base + threadIdx.x;
    const int v = i < NB ? bsum[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < SG_SCAN; o <<= 1) {                 // Hillis-Steele inclusive scan
      const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < NB) bsum[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == SG_SCAN - 1) carry += buf[threadIdx.x];
    __syncthreads();
  }
}

// start[i] = bsum[block] + exclusive prefix inside the block (i in [0, n]: n = n_src, start[n_src] = the number of valid edges);
// cursors zeroed
__global__ void __launch_bounds__(SG_SCAN) k_seg_apply(const int* __restrict__ cnt, int64_t n, const int* __restrict__ bsum,
                                                       int* __restrict__ start, int* __restrict__ cur, int* __restrict__ nlong,
                                                       int* __restrict__ longs) {
  __shared__ int buf[SG_SCAN];
  const int64_t i = (int64_t)blockIdx.x * SG_SCAN + threadIdx.x;
  const int v = i < n ? cnt[i] : 0;
  buf[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < SG_SCAN; o <<= 1) {
    const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
    __syncthreads();
    buf[threadIdx.x] += t;
    __syncthreads();
  }
  if (i <= n) start[i] = bsum[blockIdx.x] + buf[threadIdx.x] - v;
  if (i < n) cur[i] = 0;
  if (i < n && v > SR_LONG) longs[atomicAdd(nlong, 1)] = (int)i;          // (at most E / SR_LONG of them)
}

__global__ void __launch_bounds__(256) k_seg_place(const int32_t* __restrict__ idx, int64_t E, int64_t n_src, const int* __restrict__ start,
                                                   int* __restrict__ cur, int* __restrict__ U) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int s = idx[e];
    if (s >= 0 && s < n_src) U[start[s] + atomicAdd(&cur[s], 1)] = (int)e;
  }
}

// Ranges of more than SR_LONG entries — a source sampled by hundreds or thousands of destinations of the block: a hub of a power-law
// graph under replace = True sampling — would cost k_seg_rank's walk count^2 loads in one wave.  They are sorted in chunks of SR_CHUNK
// edge ids through LDS (bitonic), one workgroup per (source, chunk): a range of one chunk goes straight to sorted[]; a longer one is
// sorted chunk by chunk IN PLACE in U, and k_seg_rank then gives every entry its place in its own chunk + its lower bounds in the
// others (edge ids are distinct).  The long sources come as a list k_seg_apply collected (any order: every range is sorted by itself).
__global__ void __launch_bounds__(256) k_seg_sort_chunks(int* __restrict__ U, const int* __restrict__ start, const int* __restrict__ nlong,
                                                         const int* __restrict__ longs, int* __restrict__ sorted) {
  __shared__ int key[SR_CHUNK];
  const int tid = threadIdx.x, nl = *nlong;
  for (int q = 0; q < nl; ++q) {
    const int src = longs[q];
    const int a = start[src], c = start[src + 1] - a;
    const int nch = (c + SR_CHUNK - 1) / SR_CHUNK;
    for (int j = 0; j < nch; ++j) {
      if ((unsigned)(q + j) % gridDim.x != blockIdx.x) continue;            // (block-uniform)
      const int c0 = j * SR_CHUNK, cn = min(SR_CHUNK, c - c0);
      int P = 64;
      while (P < cn) P <<= 1;                                 // the chunk padded to a power of two with +infinity keys
      for (int i = tid; i < P; i += 256) key[i] = i < cn ? U[a + c0 + i] : 0x7FFFFFFF;
      __syncthreads();
      for (int k = 2; k <= P; k <<= 1)
        for (int h = k >> 1; h > 0; h >>= 1) {
          for (int i = tid; i < (P >> 1); i += 256) {
            const int lo = ((i & ~(h - 1)) << 1) | (i & (h - 1)), hi = lo | h;
            const bool up = (lo & k) == 0;
            const int x = key[lo], y = key[hi];
            if ((x > y) == up) { key[lo] = y; key[hi] = x; }
          }
          __syncthreads();
        }
      int* const dst = nch == 1 ? sorted : U;
      for (int i = tid; i < cn; i += 256) dst[a + c0 + i] = key[i];
      __syncthreads();
    }
  }
}

// thread per placed position: its entry moves to its rank among the entries of the same source (edge ids are distinct)
__global__ void __launch_bounds__(256) k_seg_rank(const int32_t* __restrict__ idx, const int* __restrict__ U, const int* __restrict__ start,
                                                  int64_t n_src, int* __restrict__ sorted) {
  const int total = start[n_src];
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = U[t];
    const int s = idx[e];
    const int a = start[s], b = start[s + 1], c = b - a;
    int rank = 0;
    if (c <= SR_LONG) {
      for (int k = a; k < b; ++k) rank += U[k] < e ? 1 : 0;   // (neighbouring threads walk the same range: broadcast loads)
    } else if (c <= SR_CHUNK) {
      continue;                                               // (one sorted chunk: k_seg_sort_chunks wrote it to sorted[] itself)
    } else {
      const int mine = ((int)t - a) / SR_CHUNK;               // U holds the range as sorted chunks: the place in its own chunk ...
      rank = (int)t - a - mine * SR_CHUNK;
      for (int j = 0; j * SR_CHUNK < c; ++j) {                // ... + the entries below e in every other chunk
        if (j == mine) continue;
        const int* ch = U + a + j * SR_CHUNK;
        int lo = 0, hi = min(SR_CHUNK, c - j * SR_CHUNK);
        while (lo < hi) { const int m = (lo + hi) >> 1; if (ch[m] < e) lo = m + 1; else hi = m; }
        rank += lo;
      }
    }
    sorted[a + rank] = e;
  }
}

struct SegOut {
  float* out; int64_t ldo;                    // fp32 rows [n_src, D] (nullable)
  unsigned char* img; int64_t img_row_bytes;  // row-major bf16x3 image, n_src + 1 rows (nullable)
  const float* mask; int64_t ldm;             // optional: row s is multiplied by [mask[s, :] > 0]
  float divisor;                              // S for the mean, 1 for the sum
  const float* add; int64_t lda; int64_t n_add;   // optional: rows s < n_add of the fp32 output get add[s, :] on top (after scale and mask)
};

// the mask of this thread's 4 columns of source s as {> 0 : keep}
__device__ __forceinline__ float4 seg_mask4(const SegOut& o, int64_t s, int ch) {
  return *(const float4*)(o.mask + s * o.ldm + 4 * ch);
}

// one finished row chunk (this thread's 4 columns of source s): scale, mask, store
__device__ __forceinline__ void seg_store(const SegOut& o, int64_t s, int ch, int D, float4 a) {
  const int b4 = ch * 4;
  if (o.divisor != 1.f) { a.x /= o.divisor; a.y /= o.divisor; a.z /= o.divisor; a.w /= o.divisor; }
  if (o.mask && b4 < D) {
    const float4 m = seg_mask4(o, s, ch);
    a.x = m.x > 0.f ? a.x : 0.f; a.y = m.y > 0.f ? a.y : 0.f; a.z = m.z > 0.f ? a.z : 0.f; a.w = m.w > 0.f ? a.w : 0.f;
  }
  if (o.add && s < o.n_add && b4 < D) {       // (the head rows' own gradient joins here: no add launch behind the backward)
    const float4 t = *(const float4*)(o.add + s * o.lda + b4);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
  const float e0 = b4 < D ? a.x : 0.f, e1 = b4 + 1 < D ? a.y : 0.f, e2 = b4 + 2 < D ? a.z : 0.f, e3 = b4 + 3 < D ? a.w : 0.f;
  if (o.out && b4 < D) {
    float* p = o.out + s * o.ldo + b4;
    if (b4 + 4 <= D) *(float4*)p = make_float4(e0, e1, e2, e3);
    else { p[0] = e0; if (b4 + 1 < D) p[1] = e1; if (b4 + 2 < D) p[2] = e2; }
  }
  if (o.img) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3(e0, e1, h0, m0, l0); split3(e2, e3, h1, m1, l1);
    unsigned char* rp = o.img + s * o.img_row_bytes + (int64_t)(ch >> 3) * 192 + (ch & 1) * 8;
    const int pc = (ch & 7) >> 1;
    *(uint2*)(rp + x3_piece(pc, 0) * 16) = make_uint2(h0, h1);
    *(uint2*)(rp + x3_piece(pc, 1) * 16) = make_uint2(m0, m1);
    *(uint2*)(rp + x3_piece(pc, 2) * 16) = make_uint2(l0, l1);
  }
}

template <int TILE>
__global__ void __launch_bounds__(SG_THREADS) k_seg_reduce(const float* __restrict__ dout, int64_t ldd, int64_t n_dst, int S, int D,
                                                           const int32_t* __restrict__ idx, const int* __restrict__ sorted,
                                                           const int* __restrict__ start, int64_t n_src, SegOut o,
                                                           float* __restrict__ partial, int64_t ldpart) {
  // per entry of the tile: its destination row, its source, and whether the source's whole range lies inside the tile (then the
  // run is finished here) — fetched once, in parallel, so that the walk below never waits for a dependent scalar load
  __shared__ int se[TILE], ss[TILE];
  __shared__ unsigned char whole[TILE];
  const int total = start[n_src];
  const int t0 = blockIdx.x * TILE;
  if (t0 >= total) return;
  const int t1 = min(total, t0 + TILE), cntk = t1 - t0;
  const int tid = threadIdx.x;
  if (tid < TILE) {
    const int e = tid < cntk ? sorted[t0 + tid] : 0;
    const int s = tid < cntk ? idx[e] : -1;
    se[tid] = e / S;                                          // the destination row of the entry
    ss[tid] = s;
    whole[tid] = (s >= 0 && start[s] >= t0 && start[s + 1] <= t1) ? 1 : 0;
  }
  __syncthreads();
  const int D4 = (D + 3) >> 2;
  const int Kp4 = o.img ? (int)(o.img_row_bytes / 192) * 8 : D4;   // chunks of the padded image row
  if (tid >= max(D4, Kp4)) return;
  const bool cin = tid < D4;
  const int ch = cin ? tid : D4 - 1;
  SegOut plain = o;                                           // (the mask is applied here, from a prefetched value)
  plain.mask = nullptr;
  const bool masked = o.mask != nullptr;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int run_k0 = 0;
  for (int k0 = 0; k0 < cntk; k0 += SG_U) {
    float4 v[SG_U], mv[SG_U];
    bool ends[SG_U];
#pragma unroll
    for (int u = 0; u < SG_U; ++u) {                          // unconditional loads (a slot past the tile re-reads its last row)
      const int k = k0 + u < cntk ? k0 + u : cntk - 1;
      v[u] = *(const float4*)(dout + (int64_t)se[k] * ldd + 4 * ch);
      ends[u] = k0 + u < cntk && (k0 + u + 1 == cntk || ss[k + 1] != ss[k]);
      // the ReLU mask of a run that ENDS at this entry, requested with the batch's rows (any other entry re-reads row 0: cached)
      if (masked) mv[u] = seg_mask4(o, (int64_t)((ends[u] && whole[k]) ? ss[k] : 0), ch);
    }
#pragma unroll
    for (int u = 0; u < SG_U; ++u) {
      const int k = k0 + u;
      if (k >= cntk) break;                                   // (block-uniform)
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      if (ends[u]) {                                          // (block-uniform) entries [run_k0, k] of source ss[k]
        float4 r = cin ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
        if (whole[k]) {
          if (masked && cin) {
            if (o.divisor != 1.f) { r.x /= o.divisor; r.y /= o.divisor; r.z /= o.divisor; r.w /= o.divisor; }
            r.x = mv[u].x > 0.f ? r.x : 0.f; r.y = mv[u].y > 0.f ? r.y : 0.f; r.z = mv[u].z > 0.f ? r.z : 0.f; r.w = mv[u].w > 0.f ? r.w : 0.f;
            SegOut q = plain; q.divisor = 1.f;
            seg_store(q, ss[k], tid, D, r);
          } else seg_store(plain, ss[k], tid, D, r);
        } else if (cin) {
          *(float4*)(partial + ((int64_t)blockIdx.x * 2 + (run_k0 == 0 ? 0 : 1)) * ldpart + 4 * tid) = r;
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        run_k0 = k + 1;
      }
    }
  }
}

// Blocks of at most SG_SMALL_E entries (the 512-seed output block's 12 800 edges): ONE launch, a block per SGR_SRC consecutive sources —
// their planned lists are one contiguous range of sorted[] (block-uniform: scalar loads of the range bounds and of the entries), a thread
// per 16-byte column chunk, SGR_U row loads in flight, a source's row finished and stored (scale, mask, addend, fp32 / image: seg_store)
// at its last entry; sources nobody sampled get their zero row; the block that holds source n_src writes the image's zero row.  No
// partial rows, no fix-up launch, no add launch behind it: in the traced 'mean' step at the Reddit rung 16.8 us where k_seg_reduce<16>,
// k_seg_fixup and the add of the head rows' gradient took 15.0 + 6.4 + 5.0 (8 sources and 8 loads per trip measured the same: the launch is
// three dependent round trips — bounds, entries, rows — long, not bandwidth- or trip-bound).
#define SGR_SRC 4
#define SGR_U 16
__global__ void __launch_bounds__(256) k_seg_rows(const float* __restrict__ dout, int64_t ldd, int S, int D, const int* __restrict__ sorted,
                                                  const int* __restrict__ start, int64_t n_src, SegOut o) {
  __shared__ int bnd[SGR_SRC + 1];                                // start[s0 .. s0 + SGR_SRC] (clamped: sources past n_src are empty)
  const int tid = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * SGR_SRC;
  if (tid <= SGR_SRC) bnd[tid] = start[s0 + tid < n_src ? s0 + tid : n_src];
  __syncthreads();
  const int D4 = (D + 3) >> 2;
  const int Kp4 = o.img ? (int)(o.img_row_bytes / 192) * 8 : D4;
  if (tid >= max(D4, Kp4)) return;
  const bool cin = tid < D4;
  const float* src = dout + 4 * (cin ? tid : D4 - 1);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc = zero;
  int cur = 0, lo = bnd[0], nxt = bnd[1];                         // the source the walk is in, its range [lo, nxt)
  const int k_end = bnd[SGR_SRC];
  // (block-uniform control flow throughout: every thread sees the same bounds and entries)
  while (cur < SGR_SRC && nxt == lo) {                            // sources without an edge in front of the first entry
    if (s0 + cur < n_src) seg_store(o, s0 + cur, tid, D, zero);
    ++cur; nxt = cur < SGR_SRC ? bnd[cur + 1] : -1;
  }
  for (int k0 = bnd[0]; k0 < k_end; k0 += SGR_U) {
    float4 v[SGR_U];
#pragma unroll
    for (int u = 0; u < SGR_U; ++u) {                             // (a slot past the range re-reads its last entry's row)
      const int e = __builtin_amdgcn_readfirstlane(sorted[k0 + u < k_end ? k0 + u : k_end - 1]);
      v[u] = *(const float4*)(src + (int64_t)(e / S) * ldd);
    }
#pragma unroll
    for (int u = 0; u < SGR_U; ++u) {
      const int k = k0 + u;
      if (k < k_end) {
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        if (k + 1 == nxt) {                                       // the source's last entry
          seg_store(o, s0 + cur, tid, D, cin ? acc : zero);
          acc = zero;
          ++cur; lo = nxt; nxt = cur < SGR_SRC ? bnd[cur + 1] : -1;
          while (cur < SGR_SRC && nxt == lo) {                    // sources without an edge behind it
            if (s0 + cur < n_src) seg_store(o, s0 + cur, tid, D, zero);
            ++cur; nxt = cur < SGR_SRC ? bnd[cur + 1] : -1;
          }
        }
      }
    }
  }
  if (o.img && s0 <= n_src && n_src < s0 + SGR_SRC) {             // the image's all-zero row (row n_src)
    SegOut z = o; z.out = nullptr; z.mask = nullptr; z.add = nullptr;
    seg_store(z, n_src, tid, D, zero);
  }
}

// wave per source (+ one wave for the image's zero row): sources spanning several tiles, sources without an edge
__global__ void __launch_bounds__(256) k_seg_fixup(const int* __restrict__ start, int64_t n_src, int D, SegOut o,
                                                   const float* __restrict__ partial, int64_t ldpart, int TILE) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s > n_src) return;
  const int D4 = (D + 3) >> 2;
  const int Kp4 = o.img ? (int)(o.img_row_bytes / 192) * 8 : D4;
  const int nch = max(D4, Kp4);
  if (s == n_src) {                                           // the image's all-zero row
    if (o.img) {
      SegOut z = o; z.out = nullptr; z.mask = nullptr;
      for (int ch = lane; ch < nch; ch += 64) seg_store(z, s, ch, D, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    return;
  }
  const int a = start[s], b = start[s + 1];
  if (a == b) {
    SegOut z = o; z.mask = nullptr;
    for (int ch = lane; ch < nch; ch += 64) seg_store(z, s, ch, D, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  const int tf = a / TILE, tl = (b - 1) / TILE;
  if (tf == tl) return;                                       // finished by its tile
  for (int ch = lane; ch < nch; ch += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < D4) {
      for (int t = tf; t <= tl; ++t) {                        // tile order: the order of the sorted list
        const int slot = (t == tf && a > t * TILE) ? 1 : 0;
        const float4 v = *(const float4*)(partial + ((int64_t)t * 2 + slot) * ldpart + 4 * ch);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    seg_store(o, s, ch, D, acc);
  }
}


// ---- the same backward, written as the TRANSPOSED group-major image (x6_arith.h) ------------------------------------------------
// The first layer of 'meanpool' has ONE consumer for dP — fc_pool's weight gradient dW = dP^T . X — and that product is fastest on the
// 256 x 128 tile whose A operand is the image of dP^T (what the 'pool' mode's backward writes, pool_bwd_x3.hip): sources dealt round-robin
// over G groups of 32 (source = lane * G + group), group b one contiguous slab of (D + 1) x 192 bytes.
//
//   plan   k_seg_gscan     gbase[g] = first entry of group g in a GROUP-MAJOR copy of the sorted lists (one block: totals + exclusive scan)
//          k_seg_gplace    one wave per group: its 32 sources' lists, lane after lane, as (destination row, lane | last-of-source) entries
//   apply  k_seg_groups    one block per group, thread = two columns: the group's entries are ONE contiguous run read with scalar
//                          loads (block-uniform: no per-entry vector instruction besides the row load and the add), SGG_U row loads in
//                          flight; a finished source is scaled, masked and parked in the group's 32 x D slab in LDS (which starts out
//                          holding the ReLU mask's rows), and the slab leaves as the group's 192-byte pieces of every image row.
// What the first two versions of this kernel taught (tools/seg_t_probe.py; both 157-230 us alone, like the row-wise launch): with a
// thread per column, TEN waves repeat every entry's bookkeeping (list entry from LDS, end-of-source test, branch) — 100 us of issue
// slots for 176 k edges with every memory access ablated; and cutting the columns into eight XCD-local slices made the gathers L2 hits
// (74 % hit rate) but the image writes 16-byte islands.  Hence: entries in SGPRs, few waves per row, whole 192-byte pieces per store run:
// 121 us alone (the row-wise launch: 163), of which the mask rows 32, the gathers 18 and the image 47 with the others ablated — the launch
// moves what the fabric delivers (FETCH + WRITE ~ 6 TB/s).  Column parts of a group as separate blocks (2, 3, 4 per group; 8 = one per
// XCD, the L2-resident slices again) measured 128 / 130 / 146 / 138 us: the parts repeat the walk, the walk is what is not free.
#define SGG_THREADS 320              // thread = two columns (D <= 640)
#define SGG_U 16

// one block: group totals and their exclusive scan (gbase[G] = all entries)
__global__ void __launch_bounds__(1024) k_seg_gscan(const int* __restrict__ start, int64_t n_src, int G, int* __restrict__ gbase,
                                                    int2* __restrict__ gent) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int g0 = 0; g0 < G; g0 += 1024) {
    const int g = g0 + tid;
    int c = 0;
    if (g < G)
      for (int l = 0; l < 32; ++l) {
        const int64_t s = (int64_t)l * G + g;
        if (s < n_src) c += start[s + 1] - start[s];
      }
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o);
      if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int before = carry;
    for (int k = 0; k < wv; ++k) before += wsum[k];
    if (g < G) gbase[g] = before + inc - c;
    __syncthreads();
    if (tid == 1023) carry = before + inc;
    __syncthreads();
  }
  if (tid == 0) gbase[G] = carry;
  if (tid < SGG_U) gent[carry + tid] = make_int2(0, 0);          // the padding behind the last group's run (row 0, no flag)
}

// one wave per group: entry = (destination row, lane | 0x100 on the source's last edge), sources in lane order, edges in list order
__global__ void __launch_bounds__(256) k_seg_gplace(const int* __restrict__ start, const int* __restrict__ sorted, int64_t n_src, int G, int S,
                                                    const int* __restrict__ gbase, int2* __restrict__ gent) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  const int64_t s = (int64_t)lane * G + g;
  int a = 0, c = 0;
  if (lane < 32 && s < n_src) { a = start[s]; c = start[s + 1] - a; }
  int inc = c;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
  }
  int2* out = gent + gbase[g] + inc - c;
  for (int j = 0; j < c; ++j) out[j] = make_int2(sorted[a + j] / S, lane | (j + 1 == c ? 0x100 : 0));
}

template <bool OFF32>                // (n_dst * ldd < 2^31: 32-bit row offsets, one scalar multiply per entry)
__global__ void __launch_bounds__(SGG_THREADS) k_seg_groups(const float* __restrict__ dout, int64_t ldd, int D, int DP,
                                                            const int* __restrict__ gbase, const int2* __restrict__ gent, int64_t n_src,
                                                            int G, const float* __restrict__ mask, int64_t ldm, float scale,
                                                            unsigned char* __restrict__ img, int64_t gstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sgg_smem[];
  float* T = (float*)sgg_smem;                                    // [32][DP] (DP even)
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int c2 = 2 * tid;
  const bool col = c2 < D;
  const int r0 = gbase[b], n = gbase[b + 1] - r0;                 // (block-uniform: scalar loads)
  if (col) {                                                      // the slab starts as the mask rows (1 without a mask, 0 past n_src)
    float2 m[32];
#pragma unroll
    for (int l = 0; l < 32; ++l) {
      const int64_t s = (int64_t)l * G + b;
      m[l] = s < n_src ? (mask ? *(const float2*)(mask + s * ldm + c2) : make_float2(1.f, 1.f)) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int l = 0; l < 32; ++l) *(float2*)(T + l * DP + c2) = m[l];
  }
  // (every thread touches its own two columns of the slab until the emit: no barrier before it)
  const int2* __restrict__ ge = gent + r0;
  const float* src = dout + (col ? c2 : 0);                       // (threads past the last column load column 0 and drop it)
  float2 acc = make_float2(0.f, 0.f);
  unsigned seen = 0u;                                             // lanes of the group that had an edge (block-uniform)
  for (int k0 = 0; k0 < n; k0 += SGG_U) {
    int fl[SGG_U];
    float2 v[SGG_U];
#pragma unroll
    for (int u = 0; u < SGG_U; ++u) {                             // (a slot past the group's run reads the next group's entry — or the
      const int2 e = ge[k0 + u];                                  //  list's SGG_U entries of padding —: a valid row, dropped)
      const int d = __builtin_amdgcn_readfirstlane(e.x);
      fl[u] = __builtin_amdgcn_readfirstlane(e.y);
      v[u] = OFF32 ? *(const float2*)(src + (unsigned)d * (unsigned)ldd) : *(const float2*)(src + (int64_t)d * ldd);
    }
#pragma unroll
    for (int u = 0; u < SGG_U; ++u) {
      if (k0 + u < n) {                                           // (block-uniform)
        acc.x += v[u].x; acc.y += v[u].y;
        if (fl[u] & 0x100) {                                      // (block-uniform) the source's last edge: scale, mask, park
          const int l = fl[u] & 31;
          seen |= 1u << l;
          if (col) {
            float2* t = (float2*)(T + l * DP + c2);
            const float2 mk = *t;
            *t = make_float2(mk.x > 0.f ? acc.x * scale : 0.f, mk.y > 0.f ? acc.y * scale : 0.f);
          }
          acc = make_float2(0.f, 0.f);
        }
      }
    }
  }
  if (col) {
#pragma unroll
    for (int l = 0; l < 32; ++l)
      if (!((seen >> l) & 1u)) *(float2*)(T + l * DP + c2) = make_float2(0.f, 0.f);   // a source nobody sampled (still holds the mask)
  }
  __syncthreads();
  pb_emit(T, D, DP, b, img, gstride, tid, SGG_THREADS);
}

static int64_t seg_ints_lists(int64_t E, int64_t n_src) {
  const int64_t nb = ogl_cdiv(n_src + 1, SG_SCAN);
  // cnt [n_src + 1] | start [n_src + 1] | cur [n_src] | bsum [nb] | U [E] | sorted [E] | nlong [4] | longs [E / 64 + 1]
  return (n_src + 1) * 2 + n_src + nb + 2 * E + 4 + (E / 64 + 1) + 16;
}
// ... | gbase [G + 1] | gent [E] x int2 (8-byte aligned): the group-major lists (ogl_reduce_bwd_seg_plan with group_lists = 1)
static int64_t seg_gbase_off(int64_t E, int64_t n_src) { return ogl_round_up(seg_ints_lists(E, n_src), 2); }
static int64_t seg_gent_off(int64_t E, int64_t n_src) { return ogl_round_up(seg_gbase_off(E, n_src) + ogl_cdiv(n_src, 32) + 1, 2); }
static int64_t seg_ints(int64_t E, int64_t n_src) { return seg_gent_off(E, n_src) + 2 * (E + SGG_U); }   // (+ SGG_U entries of padding)

extern "C" int64_t ogl_reduce_bwd_seg_workspace_bytes(int64_t n_dst, int fanout, int d, int64_t n_src) {
  if (n_dst < 0 || fanout < 0 || d < 0 || n_src < 0) return OGL_EINVAL;
  const int64_t E = n_dst * fanout;
  const int64_t tiles = ogl_cdiv(E, seg_tile(E)) + 1;
  return ogl_round_up(seg_ints(E, n_src) * 4, 256) + tiles * 2 * ogl_round_up(d, 4) * 4;
}

// Plan: needs the indices only (run it beside the forward pass).  workspace: ogl_reduce_bwd_seg_workspace_bytes, 16-byte aligned.
static int seg_plan_groups(int64_t n_dst, int fanout, int64_t n_src, void* workspace, int64_t workspace_bytes, ogl_stream_t stream);

extern "C" int ogl_reduce_bwd_seg_plan(const int32_t* idx, int64_t n_dst, int fanout, int64_t n_src, int group_lists, void* workspace,
                                       int64_t workspace_bytes, ogl_stream_t stream) {
  if (n_dst < 0 || fanout < 0 || n_src <= 0 || n_src >= (1ll << 31) || n_dst * (int64_t)fanout >= (1ll << 31)) return OGL_EINVAL;
  const int64_t E = n_dst * fanout;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < ogl_round_up(seg_ints(E, n_src) * 4, 256)) return OGL_EWORKSPACE;
  if (E > 0 && !idx) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int* cnt = (int*)workspace;
  int* start = cnt + (n_src + 1);
  int* cur = start + (n_src + 1);
  const int64_t nb = ogl_cdiv(n_src + 1, SG_SCAN);
  int* bsum = cur + n_src;
  int* U = bsum + nb;
  int* sorted = U + E;
  int* nlong = sorted + E;
  int* longs = nlong + 4;
  // (a kernel, not hipMemsetAsync: a memset node of a captured graph re-runs on 1/16 of its range on ROCm 7.2, tools/graph_probe.py)
  const int rc = ogl_fill_zero(cnt, (n_src + 1) * 4, stream);
  if (rc != OGL_OK) return rc;
  if (E > 0) {
    hipLaunchKernelGGL(k_seg_count, dim3((unsigned)std::min<int64_t>(2048, ogl_cdiv(E, 256))), dim3(256), 0, st, idx, E, n_src, cnt);
    OGL_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_seg_bsum, dim3((unsigned)nb), dim3(SG_SCAN), 0, st, (const int*)cnt, n_src + 1, bsum);
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_seg_scan_sums, dim3(1), dim3(SG_SCAN), 0, st, bsum, (int)nb, nlong);
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_seg_apply, dim3((unsigned)nb), dim3(SG_SCAN), 0, st, (const int*)cnt, n_src, (const int*)bsum, start, cur, nlong, longs);
  OGL_CHECK_LAUNCH();
  if (E > 0) {
    hipLaunchKernelGGL(k_seg_place, dim3((unsigned)std::min<int64_t>(2048, ogl_cdiv(E, 256))), dim3(256), 0, st, idx, E, n_src,
                       (const int*)start, cur, U);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_seg_sort_chunks, dim3(64), dim3(256), 0, st, U, (const int*)start, (const int*)nlong, (const int*)longs, sorted);
    OGL_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_seg_rank, dim3((unsigned)std::min<int64_t>(4096, ogl_cdiv(E, 256))), dim3(256), 0, st, idx, (const int*)U,
                       (const int*)start, n_src, sorted);
    OGL_CHECK_LAUNCH();
  }
  return group_lists ? seg_plan_groups(n_dst, fanout, n_src, workspace, workspace_bytes, stream) : OGL_OK;
}

// Apply: dsrc = (sum over the planned edge lists of dout rows) / divisor, masked by [mask > 0] when given; written as fp32 rows
// (`out`, nullable) and / or as the row-major bf16x3 image of [n_src, d] (`image`: ogl_x3_image_bytes(n_src, d), nullable).
// op: OGL_REDUCE_MEAN (divisor fanout) or OGL_REDUCE_SUM.  d a multiple of 4 <= 1024; 16-byte aligned rows.
static int g_seg_rows = 1;          // (ogl_debug_set(OGL_KNOB_SEG_ROWS): 0 = small blocks through k_seg_reduce<16> + k_seg_fixup as before)
int oglx_knob_seg_rows(int on, int* prev) { *prev = g_seg_rows; g_seg_rows = on ? 1 : 0; return OGL_OK; }

static int seg_apply(const float* dout, int64_t ldd, const int32_t* idx, int64_t n_dst, int fanout, int d, int op, int64_t n_src,
                     const float* mask, int64_t ldm, const float* add, int64_t lda, int64_t n_add, float* out, int64_t ldo, void* image,
                     void* workspace, int64_t workspace_bytes, ogl_stream_t stream) {
  if (add && (!out || n_add < 0 || n_add > n_src || lda < d || (lda & 3) || ((uintptr_t)add & 15))) return OGL_EINVAL;
  if (n_dst < 0 || fanout <= 0 || d <= 0 || d > 1024 || (d & 3) || n_src <= 0 || ldd < d || (ldd & 3)) return OGL_EINVAL;
  if (op != OGL_REDUCE_MEAN && op != OGL_REDUCE_SUM) return OGL_EINVAL;
  if ((!out && !image) || (out && (ldo < d || (ldo & 3) || ((uintptr_t)out & 15))) || (mask && (ldm < d || (ldm & 3) || ((uintptr_t)mask & 15))))
    return OGL_EINVAL;
  if ((image && ((uintptr_t)image & 15)) || ((uintptr_t)dout & 15)) return OGL_EINVAL;
  const int64_t E = n_dst * fanout;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < ogl_reduce_bwd_seg_workspace_bytes(n_dst, fanout, d, n_src)) return OGL_EWORKSPACE;
  if (n_dst > 0 && (!dout || !idx)) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int* cnt = (int*)workspace;
  int* start = cnt + (n_src + 1);
  const int64_t nb = ogl_cdiv(n_src + 1, SG_SCAN);
  int* sorted = start + (n_src + 1) + n_src + nb + E;
  float* partial = (float*)((unsigned char*)workspace + ogl_round_up(seg_ints(E, n_src) * 4, 256));
  const int64_t ldpart = ogl_round_up(d, 4);
  SegOut o;
  o.out = out; o.ldo = ldo; o.img = (unsigned char*)image; o.img_row_bytes = ogl_cdiv(d, 32) * 192; o.mask = mask; o.ldm = ldm;
  o.divisor = op == OGL_REDUCE_MEAN ? (float)fanout : 1.f;
  o.add = add; o.lda = lda; o.n_add = add ? n_add : 0;
  const int tile = seg_tile(E);
  if (tile == SG_TILE_SMALL && g_seg_rows) {                      // small blocks: one launch, a block per SGR_SRC sources
    hipLaunchKernelGGL(k_seg_rows, dim3((unsigned)ogl_cdiv(n_src + 1, SGR_SRC)), dim3(256), 0, st, dout, ldd, fanout, d, (const int*)sorted,
                       (const int*)start, n_src, o);
    OGL_CHECK_LAUNCH();
    return OGL_OK;
  }
  if (E > 0) {
    if (tile == SG_TILE_SMALL)
      hipLaunchKernelGGL(k_seg_reduce<SG_TILE_SMALL>, dim3((unsigned)ogl_cdiv(E, SG_TILE_SMALL)), dim3(SG_THREADS), 0, st, dout, ldd, n_dst, fanout,
                         d, idx, (const int*)sorted, (const int*)start, n_src, o, partial, ldpart);
    else
      hipLaunchKernelGGL(k_seg_reduce<SG_TILE>, dim3((unsigned)ogl_cdiv(E, SG_TILE)), dim3(SG_THREADS), 0, st, dout, ldd, n_dst, fanout, d, idx,
                         (const int*)sorted, (const int*)start, n_src, o, partial, ldpart);
    OGL_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_seg_fixup, dim3((unsigned)ogl_cdiv(n_src + 1, 4)), dim3(256), 0, st, (const int*)start, n_src, d, o,
                     (const float*)partial, ldpart, tile);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

extern "C" int ogl_reduce_bwd_seg_apply(const float* dout, int64_t ldd, const int32_t* idx, int64_t n_dst, int fanout, int d, int op,
                                        int64_t n_src, const float* mask, int64_t ldm, const float* add, int64_t lda, int64_t n_add,
                                        float* out, int64_t ldo, void* image, void* workspace, int64_t workspace_bytes,
                                        ogl_stream_t stream) {
  return seg_apply(dout, ldd, idx, n_dst, fanout, d, op, n_src, mask, ldm, add, lda, n_add, out, ldo, image, workspace, workspace_bytes, stream);
}

// The group-major copy of a planned workspace's lists (ogl_reduce_bwd_seg_plan with group_lists = 1): what ogl_reduce_bwd_seg_apply_t
// walks.  Gradient-free like the plan itself.
static int seg_plan_groups(int64_t n_dst, int fanout, int64_t n_src, void* workspace, int64_t workspace_bytes, ogl_stream_t stream) {
  if (n_dst < 0 || fanout <= 0 || n_src <= 0 || n_src >= (1ll << 31) || n_dst * (int64_t)fanout >= (1ll << 31)) return OGL_EINVAL;
  const int64_t E = n_dst * fanout;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < ogl_round_up(seg_ints(E, n_src) * 4, 256)) return OGL_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* cnt = (int*)workspace;
  const int* start = cnt + (n_src + 1);
  const int64_t nb = ogl_cdiv(n_src + 1, SG_SCAN);
  const int* sorted = start + (n_src + 1) + n_src + nb + E;
  int* gbase = cnt + seg_gbase_off(E, n_src);
  int2* gent = (int2*)(cnt + seg_gent_off(E, n_src));
  const int64_t G = ogl_cdiv(n_src, 32);
  hipLaunchKernelGGL(k_seg_gscan, dim3(1), dim3(1024), 0, st, start, n_src, (int)G, gbase, gent);
  OGL_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_seg_gplace, dim3((unsigned)ogl_cdiv(G, 4)), dim3(256), 0, st, start, sorted, n_src, (int)G, fanout, (const int*)gbase, gent);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// The masked mean / sum backward as the transposed group-major image of [d rows, 32 * ceil(n_src / 32)] (ogl_x3_image_bytes(d, 32 G)):
// what ogl_linear_bwd_weight_x3k reads as its dy operand with interleave = G.  d <= 640; rows 8-byte aligned (even leading dimensions
// that cover an even number of columns); mask nullable.  The workspace holds the plan AND its group lists.
extern "C" int ogl_reduce_bwd_seg_apply_t(const float* dout, int64_t ldd, int64_t n_dst, int fanout, int d, int op, int64_t n_src,
                                          const float* mask, int64_t ldm, void* image, const void* workspace, int64_t workspace_bytes,
                                          ogl_stream_t stream) {
  const int d2 = (d + 1) & ~1;
  if (n_dst < 0 || fanout <= 0 || d <= 0 || d > 2 * SGG_THREADS || n_src <= 0 || n_src >= (1ll << 31) || ldd < d2 || (ldd & 1)) return OGL_EINVAL;
  if (op != OGL_REDUCE_MEAN && op != OGL_REDUCE_SUM) return OGL_EINVAL;
  if (!image || ((uintptr_t)image & 15) || ((uintptr_t)dout & 7) || (mask && (ldm < d2 || (ldm & 1) || ((uintptr_t)mask & 7)))) return OGL_EINVAL;
  const int64_t E = n_dst * fanout;
  if (E >= (1ll << 31)) return OGL_EINVAL;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < ogl_round_up(seg_ints(E, n_src) * 4, 256)) return OGL_EWORKSPACE;
  if (n_dst > 0 && !dout) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int* cnt = (const int*)workspace;
  const int* gbase = cnt + seg_gbase_off(E, n_src);
  const int2* gent = (const int2*)(cnt + seg_gent_off(E, n_src));
  const int64_t G = ogl_cdiv(n_src, 32);
  const int DP = d2 + 2;
  const size_t lds = (size_t)32 * DP * 4;
  const bool off32 = n_dst * ldd < (1ll << 31);
  static bool attr_set = false;
  if (!attr_set) {
    OGL_CHECK_HIP(hipFuncSetAttribute((const void*)k_seg_groups<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OGL_CHECK_HIP(hipFuncSetAttribute((const void*)k_seg_groups<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const dim3 grid((unsigned)G), threads(SGG_THREADS);
  const float scale = op == OGL_REDUCE_MEAN ? 1.f / (float)fanout : 1.f;
  if (off32)
    hipLaunchKernelGGL(k_seg_groups<true>, grid, threads, lds, st, dout, ldd, d, DP, gbase, gent, n_src, (int)G, mask, ldm, scale,
                       (unsigned char*)image, ((int64_t)d + 1) * 192);
  else
    hipLaunchKernelGGL(k_seg_groups<false>, grid, threads, lds, st, dout, ldd, d, DP, gbase, gent, n_src, (int)G, mask, ldm, scale,
                       (unsigned char*)image, ((int64_t)d + 1) * 192);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
