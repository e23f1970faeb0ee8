// Backward of the layer-0 'pool' aggregator, fused down to the operand the weight-gradient GEMM consumes.
//
// Forward (aggregate.hip / linear*.hip): P = relu(fc_pool(X[input_nodes])), out[d, f] = max_j P[idx[d, j], f] with
// argmax[d, f] = the winning block-local source row (R/train/graphsage/pytorch/aggregator_dgl.py:85-94,171: DGL's
// 'pool' SAGEConv = fc_pool -> ReLU -> max).  Backward: dP[s, f] = sum over {d : argmax[d, f] == s} of
// dout[d, f] * [out[d, f] > 0], and for layer 0 dP has ONE consumer — the weight gradient dW = dP^T . X, which
// wants dP^T as a bf16x3 image (linear_x3.hip).  The unfused path is three passes over the [n_src, D] matrix (zero
// fill, float-atomic scatter at the L2 atomic rate, transpose + split); here dP is never materialised:
//
//   k_pool_bucket    one wave per destination: finds the sampling slot of every winner, counting-sorts the row's D
//                    (column, gradient) entries by slot (LDS), writes them with the slot offsets, and marks
//                    bitmap[group][d] for every source group the destination samples (atomicOr, order-free).
//   k_pool_bwd_x3    one block per source group (32 sources = one 32-deep reduction group of the image): 16-lane
//                    teams take the group's destinations, read ONLY the entry segments whose slot points into the
//                    group and add them into the group's 32 x D slab in LDS (ds_add_f32); then the slab is split and
//                    written as the group's 192-byte pieces of every image row.
//
// Sources are dealt to groups ROUND-ROBIN (source s sits in group s % G, lane s / G; G = number of groups): block
// builders number frequently sampled vertices first, and contiguous groups would put all the hubs — and a long serial
// tail — into the first few blocks.  The image's reduction index m therefore stands for source (m & 31) * G + (m >> 5);
// the other operand of the product is built with the same dealing (ogl_x3_split_t, interleave = G).
//
// HBM-bound integer/float streaming: 3 D floats read + 8 D bytes written per destination by the bucket pass, then
// 8 bytes read per entry (each entry is read once) and 6 bytes written per slab element.
//
// PLAN / APPLY (ogl_pool_bwd_x3_plan / _apply): the same backward re-cut so that the part on the backward's critical path reads
// memory in streams.  Everything but the gradient VALUES depends on the forward pass only (argmax, the ReLU sign of the pooled
// output, the sampled indices), so the forward pass can run, beside its own products:
//   k_pool_bucket<true>   the bucket pass without values: the slot offsets, the row's columns in slot order (10 bits column + 6 bits
//                         slot per entry) and, per (destination, slot) segment, its length added to the segment's source group;
//   k_pool_plan_scan      exclusive scan of the group totals -> where every group's records start in a GROUP-MAJOR array;
//   k_pool_plan_place     every segment takes its place inside its group's range (atomic cursor): seginfo[d][slot] = position + lane.
// The backward then runs
//   k_pool_values         one wave per destination: its gradient row through LDS into the planned places — (column, lane, value)
//                         records, a destination's segment = one contiguous run of ~24 records;
//   k_pool_groups         one block per source group: ONE contiguous stream of records -> ds_add_f32 into the slab -> the image.
// Measured (round 3): the one-call form's group pass spends 115-124 us although its reads alone take 29 us and its image writes
// alone 31-37 us — scattered small reads (bitmap -> list -> indices -> offsets -> entries) collapse beside 512 blocks' streaming
// writes; here the group pass has no dependent read at all.
#include <cstdlib>
#include "x6_arith.h"

#define PB_THREADS 640              // 10 waves = 40 teams of 16 lanes
#define PB_TEAMS (PB_THREADS / 16)
#define PB_LIST 512                 // destinations per pass (uint16 offsets inside a super-chunk of 640 words)
#define PB_MAX_D 640                // columns per destination row the bucket pass keeps in registers (10 per lane)
#define PB_MAX_S 63                 // sampling slots (one lane each, plus one lane for the end offset)

struct PbDiv { unsigned mul; unsigned shift; unsigned G; };   // floor(x / G) = (x * mul) >> shift for 0 <= x < 2^31
__device__ __forceinline__ unsigned pb_div(unsigned x, PbDiv v) { return (unsigned)(((uint64_t)x * v.mul) >> v.shift); }

template <bool PLAN>
__global__ void __launch_bounds__(256) k_pool_bucket(const float* __restrict__ dout, int64_t ldo, const int32_t* __restrict__ argmax,
                                                     const float* __restrict__ relu_out, int64_t ldr,
                                                     const int32_t* __restrict__ idx, int64_t n_dst, int S, int D, int64_t n_src,
                                                     PbDiv dv, unsigned* __restrict__ bitmap, int64_t words,
                                                     unsigned short* __restrict__ off, uint2* __restrict__ ent) {
  __shared__ int cnt[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t d = (int64_t)blockIdx.x * 4 + wv;
  if (d >= n_dst) return;
  const int my = lane < S ? idx[d * S + lane] : -1;
  if (!PLAN && my >= 0 && my < n_src) {
    const unsigned q = pb_div((unsigned)my, dv);
    atomicOr(&bitmap[(int64_t)((unsigned)my - q * dv.G) * words + (d >> 5)], 1u << (d & 31));
  }
  cnt[wv][lane] = 0;
  constexpr int NI = PB_MAX_D / 64;
  int slot[NI], pos[NI];
  float gv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = lane + 64 * i;
    const bool in = f < D;
    const int fc = in ? f : 0;
    const int a = argmax[d * D + fc];
    gv[i] = PLAN ? 0.f : dout[d * ldo + fc];
    const float m = relu_out ? relu_out[d * ldr + fc] : 1.f;
    const bool ok = in && a >= 0 && a < n_src && m > 0.f;
    slot[i] = ok ? a : -1;                        // winner's source id for now; the slot search follows
  }
  // the FIRST slot holding each winner (duplicate samples own nothing): slots from the last to the first, so the
  // lowest match is the one that sticks; idx values are broadcast through an SGPR (v_readlane), not through LDS
  int sl[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) sl[i] = -1;
  for (int j = S - 1; j >= 0; --j) {
    const int sj = __builtin_amdgcn_readlane(my, j);
#pragma unroll
    for (int i = 0; i < NI; ++i) sl[i] = (slot[i] == sj && sj >= 0) ? j : sl[i];
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    slot[i] = slot[i] >= 0 ? sl[i] : -1;
    pos[i] = slot[i] >= 0 ? atomicAdd(&cnt[wv][slot[i]], 1) : 0;
  }
  // exclusive scan of the S bucket sizes (lane j = bucket j, lane S = end)
  const int c = cnt[wv][lane];
  int s = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(s, o);
    if (lane >= o) s += v;
  }
  const int start = s - c;
  if (lane <= S) off[d * (S + 1) + lane] = (unsigned short)start;
  if (PLAN && lane < S && c > 0) {                                // (PLAN: `bitmap` is the array of group totals)
    const unsigned q = pb_div((unsigned)my, dv);                  // c > 0: `my` won somewhere, so it is a valid source
    atomicAdd(&bitmap[(unsigned)my - q * dv.G], (unsigned)c);
  }
  cnt[wv][lane] = start;
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (slot[i] >= 0) {
      if constexpr (PLAN) ((unsigned short*)ent)[d * D + cnt[wv][slot[i]] + pos[i]] = (unsigned short)((lane + 64 * i) | (slot[i] << 10));
      else ent[d * D + cnt[wv][slot[i]] + pos[i]] = make_uint2((unsigned)(lane + 64 * i), __float_as_uint(gv[i]));
    }
}

__global__ void __launch_bounds__(PB_THREADS) k_pool_bwd_x3(const int32_t* __restrict__ idx, int S, const unsigned short* __restrict__ off,
                                                            const uint2* __restrict__ ent, const unsigned* __restrict__ bitmap,
                                                            int64_t words, int64_t n_src, PbDiv dv, int D, int DP,
                                                            unsigned char* __restrict__ img, int64_t gstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
  float* T = (float*)pb_smem;                                     // [32][DP] slab of dP for this source group
  unsigned short* list = (unsigned short*)(T + 32 * DP);          // up to PB_LIST destinations of the current pass
  int* wsum = (int*)(list + PB_LIST);                             // per-wave popcount totals of the current super-chunk
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x;
  for (int i = tid; i < 8 * DP; i += PB_THREADS) ((float4*)T)[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // 32 * DP floats
  const unsigned* bm = bitmap + (int64_t)b * words;
  const int team = tid >> 4, tl = tid & 15;
  const int tshift = tid & 48;                                    // first lane of this team inside its wave

  // super-chunks of PB_THREADS bitmap words (one word per thread: 20480 destinations); a typical group references
  // ~100 destinations, so the whole list is built and consumed in one pass
  for (int64_t w0 = 0; w0 < words; w0 += PB_THREADS) {
    unsigned word = 0;
    if (w0 + tid < words) word = bm[w0 + tid];
    const int c = __popc(word);
    int s = c;                                                    // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(s, o);
      if (lane >= o) s += v;
    }
    __syncthreads();                                              // previous super-chunk fully consumed (and T zeroed)
    if (lane == 63) wsum[wv] = s;
    __syncthreads();
    int before = 0, L = 0;
#pragma unroll
    for (int k = 0; k < PB_THREADS / 64; ++k) {
      const int t = wsum[k];
      if (k < wv) before += t;
      L += t;
    }
    const int rank0 = before + s - c;                             // rank of this thread's first destination
    const int64_t dbase = w0 * 32;
    for (int p0 = 0; p0 < L; p0 += PB_LIST) {
      if (p0 > 0) __syncthreads();                                // the previous pass's list is consumed
      {
        int rk = rank0;
        unsigned wbits = word;
        while (wbits) {
          const int bit = __builtin_ctz(wbits);
          wbits &= wbits - 1;
          if (rk >= p0 && rk < p0 + PB_LIST) list[rk - p0] = (unsigned short)(tid * 32 + bit);
          ++rk;
        }
      }
      __syncthreads();
      const int Lp = min(PB_LIST, L - p0);
      // ---- one destination per 16-lane team: slots whose source belongs to this group -> their entry segments -------
      const int rounds = (Lp + PB_TEAMS - 1) / PB_TEAMS;          // wave-uniform trip count (the ballots below need whole waves)
      for (int rd = 0; rd < rounds; ++rd) {
        const int i = rd * PB_TEAMS + team;
        const bool live = i < Lp;
        const int64_t d = dbase + list[live ? i : 0];
        for (int r = 0; 16 * r < S; ++r) {                        // lane tl looks at slots tl, tl + 16, ...
          const int j = tl + 16 * r;
          int a = -1, lo = 0, hi = 0;
          if (live && j < S) {
            a = idx[d * S + j];
            lo = off[d * (S + 1) + j];
            hi = off[d * (S + 1) + j + 1];
          }
          int sloc = -1;
          if (a >= 0 && a < n_src && hi > lo) {
            const unsigned q = pb_div((unsigned)a, dv);
            if ((unsigned)a - q * dv.G == (unsigned)b) sloc = (int)q;
          }
          unsigned mask = (unsigned)((__ballot(sloc >= 0) >> tshift) & 0xFFFFull);   // this team's matching slots
          // teams of one wave run the loop together: it ends when the slowest team has no slot left
          while (__any(mask != 0)) {
            const int src_lane = mask ? __builtin_ctz(mask) : 0;
            const bool act = mask != 0;
            mask &= mask - 1;
            const int sl = __shfl(sloc, tshift + src_lane);
            const int e0 = __shfl(lo, tshift + src_lane), e1 = __shfl(hi, tshift + src_lane);
            if (act)
              for (int e = e0 + tl; e < e1; e += 16) {
                const uint2 en = ent[d * D + e];
                __hip_atomic_fetch_add(&T[sl * DP + (int)en.x], __uint_as_float(en.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
          }
        }
      }
    }
  }
  __syncthreads();
  pb_emit(T, D, DP, b, img, gstride, tid, PB_THREADS);
}

// ---- the planned form -------------------------------------------------------------------------------------------------
#define PB_POS_BITS 27              // a record's place in the group-major array (n_dst * D < 2^27); the source's lane above it

// gbase[g] = first record of group g (exclusive scan of the totals k_pool_bucket<true> accumulated), gbase[G] = all records;
// the cursors the place pass advances start at zero.  One block.
__global__ void __launch_bounds__(1024) k_pool_plan_scan(const unsigned* __restrict__ gcount, unsigned* __restrict__ gbase,
                                                         unsigned* __restrict__ gcur, int G) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int g0 = 0; g0 < G; g0 += 1024) {
    const int g = g0 + tid;
    const unsigned c = g < G ? gcount[g] : 0u;
    unsigned s = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = __shfl_up(s, o);
      if (lane >= o) s += v;
    }
    if (lane == 63) wsum[wv] = s;
    __syncthreads();
    unsigned before = carry;
    for (int k = 0; k < wv; ++k) before += wsum[k];
    if (g < G) { gbase[g] = before + s - c; gcur[g] = 0u; }
    __syncthreads();
    if (tid == 1023) carry = before + s;
    __syncthreads();
  }
  if (tid == 0) gbase[G] = carry;
}

// every non-empty (destination, slot) segment takes `len` places of its group's range (atomic cursor; four segments per thread in
// flight).  A FEW blocks on purpose (PB_PLACE_BLOCKS): this runs beside the forward products, whose blocks need whole CUs — a grid
// of 689 short blocks kept cycling through every CU and the 185-tile product beside it took 52 us instead of 36 (round 3).
#define PB_PLACE_BLOCKS 16
__global__ void __launch_bounds__(1024) k_pool_plan_place(const int32_t* __restrict__ idx, int64_t n_dst, int S,
                                                          const unsigned short* __restrict__ off, PbDiv dv, const unsigned* __restrict__ gbase,
                                                          unsigned* __restrict__ gcur, unsigned* __restrict__ seginfo) {
  const int tid = threadIdx.x;
  const int64_t total = n_dst * S;
  constexpr int U = 4;
  for (int64_t i0 = (int64_t)blockIdx.x * 1024 * U; i0 < total; i0 += (int64_t)gridDim.x * 1024 * U) {
    unsigned len[U], a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 1024 + tid;
      len[u] = 0; a[u] = 0;
      if (i < total) {
        const int64_t d = i / S;
        const int j = (int)(i - d * S);
        len[u] = (unsigned)off[d * (S + 1) + j + 1] - (unsigned)off[d * (S + 1) + j];
        a[u] = (unsigned)idx[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 1024 + tid;
      if (i >= total) continue;
      unsigned info = 0xFFFFFFFFu;
      if (len[u] > 0 && a[u] < 0x80000000u) {                      // a winner's slot: a[u] is a valid source id
        const unsigned q = pb_div(a[u], dv), g = a[u] - q * dv.G;
        // (g < G by construction; the guard keeps a re-read index that is no longer what the bucket pass saw — inputs recycled
        // under a plan whose backward never ran — from writing outside the group arrays)
        if (g < dv.G) info = (gbase[g] + atomicAdd(&gcur[g], len[u])) | (q << PB_POS_BITS);
      }
      seginfo[i] = info;
    }
  }
}

// backward, one wave per destination: the gradient row goes through LDS into the planned places
__global__ void __launch_bounds__(256) k_pool_values(const float* __restrict__ dout, int64_t ldo, int64_t n_dst, int S, int D,
                                                     const unsigned short* __restrict__ off, const unsigned short* __restrict__ colperm,
                                                     const unsigned* __restrict__ seginfo, uint2* __restrict__ gent) {
  __shared__ float row[4][PB_MAX_D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t d = (int64_t)blockIdx.x * 4 + wv;
  if (d >= n_dst) return;
#pragma unroll
  for (int i = 0; i < PB_MAX_D / 64; ++i) {
    const int f = lane + 64 * i;
    if (f < D) row[wv][f] = dout[d * ldo + f];
  }
  const unsigned si = lane < S ? seginfo[d * S + lane] : 0u;
  const int oj = lane <= S ? (int)off[d * (S + 1) + lane] : 0;
  const int n = __builtin_amdgcn_readlane(oj, S);                  // S <= 63: lane S holds the end offset
  // (this wave's LDS row is written and read by the same wave: no barrier, the compiler's lgkmcnt wait orders them)
  // HOIST: every trip's column entry is requested before the first one is used — with the load inside the loop a wave paid one memory
  // latency per 64 entries, ten in a row for a 602-column row, and the kernel is one wave's dependent chain long (all 7 038 waves are
  // resident at once): -7 us per replayed Reddit step, same box, three alternations (0.9821 -> 0.9754 ms)
  unsigned cpv[PB_MAX_D / 64];
#pragma unroll
  for (int i = 0; i < PB_MAX_D / 64; ++i) {
    const int e = 64 * i + lane;
    cpv[i] = colperm[d * D + (e < n ? e : 0)];
  }
#pragma unroll
  for (int i = 0; i < PB_MAX_D / 64; ++i) {                        // whole waves: the shuffles below need every lane
    const int e = 64 * i + lane;
    if (64 * i >= n) break;                                        // (wave-uniform)
    const bool live = e < n;
    const unsigned cp = live ? cpv[i] : 0u;
    const int col = cp & 1023, j = cp >> 10;
    const unsigned sij = __shfl(si, j);
    const int o = __shfl(oj, j);
    if (live) gent[(sij & ((1u << PB_POS_BITS) - 1u)) + (unsigned)(e - o)] = make_uint2((unsigned)col | ((sij >> PB_POS_BITS) << 16), __float_as_uint(row[wv][col]));
  }
}

// backward, one block per source group: its records are one contiguous run
__global__ void __launch_bounds__(PB_THREADS) k_pool_groups(const uint2* __restrict__ gent, const unsigned* __restrict__ gbase, int D, int DP,
                                                            unsigned char* __restrict__ img, int64_t gstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
  float* T = (float*)pb_smem;                                     // [32][DP] slab of dP for this source group
  const int tid = threadIdx.x, b = blockIdx.x;
  const unsigned r0 = gbase[b], r1 = gbase[b + 1];
  uint2 first = make_uint2(0u, 0u);
  if (r0 + tid < r1) first = gent[r0 + tid];                      // in flight while the slab is cleared
  // (the later trips' records — a group holds ~2 150 = 3.4 trips of 640 threads — requested ahead as well: 0.9371 against 0.9376 ms
  // per step, no change: this pass is paced by its image writes)
  for (int i = tid; i < 8 * DP; i += PB_THREADS) ((float4*)T)[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // 32 * DP floats
  __syncthreads();
  if (r0 + tid < r1)
    __hip_atomic_fetch_add(&T[(first.x >> 16) * DP + (int)(first.x & 0xFFFFu)], __uint_as_float(first.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (unsigned r = r0 + PB_THREADS + tid; r < r1; r += PB_THREADS) {
    const uint2 en = gent[r];
    __hip_atomic_fetch_add(&T[(en.x >> 16) * DP + (int)(en.x & 0xFFFFu)], __uint_as_float(en.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  pb_emit(T, D, DP, b, img, gstride, tid, PB_THREADS);
}

static inline int64_t pb_words(int64_t n_dst) { return ogl_cdiv(n_dst, 32); }
static inline int64_t pb_bitmap_bytes(int64_t n_dst, int64_t n_src) { return ogl_round_up(ogl_cdiv(n_src, 32) * pb_words(n_dst) * 4 + 16, 256); }
static inline int64_t pb_off_bytes(int64_t n_dst, int fanout) { return ogl_round_up(n_dst * (fanout + 1) * 2 + 16, 256); }

static PbDiv pb_make_div(unsigned G) {
  unsigned l = 0;
  while ((1ull << l) < G) ++l;                                    // l = ceil(log2 G)
  PbDiv v;
  v.G = G;
  v.mul = (unsigned)((1ull << (31 + l)) / G + 1);                 // < 2^32; exact quotients for every x < 2^31
  v.shift = 31 + l;
  return v;
}

__global__ void __launch_bounds__(256) k_pb_zero16(uint4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// workspace of the planned form: [group totals | group starts (G + 1) | group cursors] [slot offsets] [columns in slot order, 2 B]
// [segment places, 4 B per (destination, slot)] [records, 8 B per (destination, column)]
struct PbPlanLayout { int64_t counts, off, colperm, seginfo, gent, total; };
static PbPlanLayout pb_plan_layout(int64_t n_dst, int fanout, int d, int64_t n_src) {
  const int64_t G = ogl_cdiv(n_src, 32);
  PbPlanLayout L;
  L.counts = 0;
  L.off = ogl_round_up((3 * G + 1) * 4 + 16, 256);
  L.colperm = L.off + pb_off_bytes(n_dst, fanout);
  L.seginfo = L.colperm + ogl_round_up(n_dst * (int64_t)d * 2 + 16, 256);
  L.gent = L.seginfo + ogl_round_up(n_dst * (int64_t)fanout * 4 + 16, 256);
  L.total = L.gent + ogl_round_up(n_dst * (int64_t)d * 8 + 16, 256);
  return L;
}

extern "C" int64_t ogl_pool_bwd_x3_workspace_bytes(int64_t n_dst, int fanout, int d, int64_t n_src) {
  if (n_dst < 0 || n_src < 0 || fanout < 0 || d < 0) return OGL_EINVAL;
  const int64_t one_call = pb_bitmap_bytes(n_dst, n_src) + pb_off_bytes(n_dst, fanout) + ogl_round_up(n_dst * (int64_t)d * 8 + 16, 256);
  return std::max(one_call, pb_plan_layout(n_dst, fanout, d, n_src).total);
}

static int pb_check(int64_t n_dst, int fanout, int d, int64_t n_src, const void* workspace, int64_t workspace_bytes) {
  if (n_dst < 0 || fanout < 0 || d <= 0 || n_src <= 0) return OGL_EINVAL;
  if (d > PB_MAX_D || fanout > PB_MAX_S || n_src >= (1ll << 31) || n_dst * (int64_t)d >= (1ll << 31)) return OGL_EINVAL;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < ogl_pool_bwd_x3_workspace_bytes(n_dst, fanout, d, n_src))
    return OGL_EWORKSPACE;
  return OGL_OK;
}

extern "C" int ogl_pool_bwd_x3(const float* dout, int64_t ldo, const int32_t* argmax, const float* relu_out, int64_t ldr,
                               const int32_t* idx32, int64_t n_dst, int fanout, int d, int64_t n_src, void* image,
                               void* workspace, int64_t workspace_bytes, ogl_stream_t stream) {
  if (ldo < d || (relu_out && ldr < d)) return OGL_EINVAL;
  const int rc = pb_check(n_dst, fanout, d, n_src, workspace, workspace_bytes);
  if (rc != OGL_OK) return rc;
  if (!image || ((uintptr_t)image & 15)) return OGL_EINVAL;
  if (n_dst > 0 && fanout > 0 && (!dout || !argmax || !idx32)) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t groups = ogl_cdiv(n_src, 32), words = pb_words(n_dst);
  unsigned* bitmap = (unsigned*)workspace;
  unsigned short* off = (unsigned short*)((unsigned char*)workspace + pb_bitmap_bytes(n_dst, n_src));
  uint2* ent = (uint2*)((unsigned char*)off + pb_off_bytes(n_dst, fanout));
  // zeroed by a KERNEL, not hipMemsetAsync: a memset node recorded into a captured hipGraph re-runs on only 1/16 of its
  // range from the second replay on (ROCm 7.2; tools/graph_probe.py shows it), and the step is replayed as a graph
  {
    const int64_t n16 = pb_bitmap_bytes(n_dst, n_src) / 16;       // the region is 256-B rounded and 16-B aligned
    hipLaunchKernelGGL(k_pb_zero16, dim3((unsigned)std::min<int64_t>(ogl_cdiv(n16, 256), 2048)), dim3(256), 0, st, (uint4*)bitmap, n16);
    OGL_CHECK_LAUNCH();
  }
  const PbDiv dv = pb_make_div((unsigned)groups);
  if (n_dst > 0 && fanout > 0) {
    hipLaunchKernelGGL((k_pool_bucket<false>), dim3((unsigned)ogl_cdiv(n_dst, 4)), dim3(256), 0, st, dout, ldo, argmax, relu_out, ldr, idx32,
                       n_dst, fanout, d, n_src, dv, bitmap, words, off, ent);
    OGL_CHECK_LAUNCH();
  }
  const int64_t gstride = ((int64_t)d + 1) * 192;                 // GROUP-MAJOR image: [group][d rows + zero row][192 B]
  const int DP = d | 1;                                            // odd slab stride: conflict-free column reads in the emit phase
  const size_t lds = (size_t)32 * DP * 4 + PB_LIST * 2 + (PB_THREADS / 64) * 4 + 16;
  static bool attr_set = false;
  if (!attr_set) {
    OGL_CHECK_HIP(hipFuncSetAttribute((const void*)k_pool_bwd_x3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_pool_bwd_x3, dim3((unsigned)groups), dim3(PB_THREADS), lds, st, idx32, fanout, off, ent, bitmap, words, n_src,
                     dv, d, DP, (unsigned char*)image, gstride);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

extern "C" int ogl_pool_bwd_x3_plan(const int32_t* argmax, const float* relu_out, int64_t ldr, const int32_t* idx32, int64_t n_dst,
                                    int fanout, int d, int64_t n_src, void* workspace, int64_t workspace_bytes, ogl_stream_t stream) {
  if (relu_out && ldr < d) return OGL_EINVAL;
  const int rc = pb_check(n_dst, fanout, d, n_src, workspace, workspace_bytes);
  if (rc != OGL_OK) return rc;
  if (n_dst * (int64_t)d >= (1ll << PB_POS_BITS)) return OGL_EINVAL;
  if (n_dst > 0 && fanout > 0 && (!argmax || !idx32)) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t G = ogl_cdiv(n_src, 32);
  const PbPlanLayout L = pb_plan_layout(n_dst, fanout, d, n_src);
  unsigned char* w = (unsigned char*)workspace;
  unsigned* gcount = (unsigned*)(w + L.counts);
  unsigned* gbase = gcount + G;
  unsigned* gcur = gbase + G + 1;
  unsigned short* off = (unsigned short*)(w + L.off);
  unsigned short* colperm = (unsigned short*)(w + L.colperm);
  unsigned* seginfo = (unsigned*)(w + L.seginfo);
  {
    const int64_t n16 = L.off / 16;                               // totals, starts, cursors (a kernel, not a memset node: see above)
    hipLaunchKernelGGL(k_pb_zero16, dim3((unsigned)std::min<int64_t>(ogl_cdiv(n16, 256), 2048)), dim3(256), 0, st, (uint4*)w, n16);
    OGL_CHECK_LAUNCH();
  }
  const PbDiv dv = pb_make_div((unsigned)G);
  if (n_dst > 0 && fanout > 0) {
    hipLaunchKernelGGL((k_pool_bucket<true>), dim3((unsigned)ogl_cdiv(n_dst, 4)), dim3(256), 0, st, (const float*)nullptr, (int64_t)0, argmax,
                       relu_out, ldr, idx32, n_dst, fanout, d, n_src, dv, gcount, (int64_t)0, off, (uint2*)colperm);
    OGL_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_pool_plan_scan, dim3(1), dim3(1024), 0, st, (const unsigned*)gcount, gbase, gcur, (int)G);
  OGL_CHECK_LAUNCH();
  if (n_dst > 0 && fanout > 0) {
    hipLaunchKernelGGL(k_pool_plan_place, dim3((unsigned)std::min<int64_t>(PB_PLACE_BLOCKS, ogl_cdiv(n_dst * fanout, 4096))), dim3(1024), 0, st,
                       idx32, n_dst, fanout, (const unsigned short*)off, dv, (const unsigned*)gbase, gcur, seginfo);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
}

extern "C" int ogl_pool_bwd_x3_apply(const float* dout, int64_t ldo, const int32_t* idx32, int64_t n_dst, int fanout, int d,
                                     int64_t n_src, void* image, const void* workspace, int64_t workspace_bytes, ogl_stream_t stream) {
  (void)idx32;                                                    // (the plan already holds what the indices say)
  if (ldo < d) return OGL_EINVAL;
  const int rc = pb_check(n_dst, fanout, d, n_src, workspace, workspace_bytes);
  if (rc != OGL_OK) return rc;
  if (n_dst * (int64_t)d >= (1ll << PB_POS_BITS)) return OGL_EINVAL;
  if (!image || ((uintptr_t)image & 15)) return OGL_EINVAL;
  if (n_dst > 0 && fanout > 0 && !dout) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t G = ogl_cdiv(n_src, 32);
  const PbPlanLayout L = pb_plan_layout(n_dst, fanout, d, n_src);
  unsigned char* w = (unsigned char*)workspace;
  const unsigned* gbase = (const unsigned*)(w + L.counts) + G;
  uint2* gent = (uint2*)(w + L.gent);
  if (n_dst > 0 && fanout > 0) {
    hipLaunchKernelGGL(k_pool_values, dim3((unsigned)ogl_cdiv(n_dst, 4)), dim3(256), 0, st, dout, ldo, n_dst, fanout, d,
                       (const unsigned short*)(w + L.off), (const unsigned short*)(w + L.colperm), (const unsigned*)(w + L.seginfo), gent);
    OGL_CHECK_LAUNCH();
  }
  const int64_t gstride = ((int64_t)d + 1) * 192;
  const int DP = d | 1;
  const size_t lds = (size_t)32 * DP * 4 + 16;
  static bool attr_set = false;
  if (!attr_set) {
    OGL_CHECK_HIP(hipFuncSetAttribute((const void*)k_pool_groups, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_pool_groups, dim3((unsigned)G), dim3(PB_THREADS), lds, st, (const uint2*)gent, gbase, d, DP, (unsigned char*)image, gstride);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
