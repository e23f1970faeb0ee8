// Dense projections on PRE-SPLIT operands ("bf16x3 images"): the same split-bf16 x6 arithmetic as linear.hip
// (x6_arith.h), but the exact 3-term bf16 split of every fp32 operand is done ONCE by a streaming kernel instead of
// by every GEMM block that touches the element.  What that buys on gfx950: the GEMM's staging becomes a pure byte
// copy, so tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no VALU, no ds_write)
// and the matrix pipe only waits for ds_read_b128 fragments.  Replaces the same torch.nn.Linear call sites as
// linear.hip (R/train/graphsage/pytorch/aggregator_dgl.py:85-94,171,181,206) for the large layer-0 products.
//
// Image of an fp32 matrix X[R, K] (reduction index contiguous): R rows + ONE all-zero row (index R), each row
// G = ceil(K/32) groups of 192 bytes; group g holds k = 32g .. 32g+31 as two 96-byte HALVES of 16 elements, each half
// three 32-byte planes (hi, mid, lo terms of the split): element e of plane p sits at byte (e / 16) * 96 + p * 32 +
// (e % 16) * 2 (x3_piece, x6_arith.h); pad k >= K is zero.  One (row, group) is 192 contiguous bytes = the 12 pieces of a
// 32-deep GEMM step, one (row, half) 96 contiguous bytes = the 6 pieces of a 16-deep step.
//
// k_gemm_x3: C[i, j] = epilogue(sum_k A[i, k] B[j, k]), both operands images (A optionally gathered by an int64 row
// list: rows outside the table read the zero row).  8 waves per block, one block per CU (144 KB LDS: two stages of
// (BM + BN) x 192 B), 2 waves per SIMD.  LDS image of a stage = the 12 pieces of row r at pieces 12r .. 12r+11 with
// the chunk index XOR-ed inside each plane by swz(r) = {0, 2, 3, 1}[(r >> 2) & 3]: the DMA is lane-linear (piece i of
// the stage lands at byte 16 i), the swizzle is applied on the per-lane SOURCE address, and every MFMA fragment is one
// conflict-free ds_read_b128 (16-lane groups {0-3,12-15,20-27}, ... hit 16 distinct 16-byte slots of the 256-byte
// bank row: slot = 4 ((plane - r) mod 4) + (chunk ^ swz(r)); with the 16x16x32 fragment a group holds rows 0-3 and
// 12-15 at chunk c and rows 4-11 at chunk c ^ 1, which is what the table — not the plain (r >> 2) & 3 — separates).
// The matrix instruction is v_mfma_f32_16x16x32_bf16 (one MFMA = one product term over the whole 32-deep step).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "x6_arith.h"

#define X3_GROUP_BYTES 192

struct X3Operand {
  const unsigned char* img;
  int64_t row_bytes;     // distance between two image rows
  int64_t step_bytes;    // distance between two consecutive 32-deep groups of one row
  const int64_t* rows;   // optional gather (A only)
  int64_t nrows;         // valid row ids are [0, nrows)
  int64_t zero_row;      // index of the image's all-zero row (>= nrows): where invalid ids and tile padding point
};

struct X3Args {
  X3Operand a, b;
  int64_t M, N;          // output rows / columns (N includes the ones column when ones_col)
  int nsteps;            // 32-deep reduction groups
  int ones_col;          // column N-1 = row sums of A (the B image carries an all-ones row there): bias gradient
  float* C; int64_t ldc;
  int relu;
  float* db;
  int nsplit, steps_per_split;
  float* ws; int64_t ws_ld;
  int NI, NJ;
  unsigned long long* stamps;   // diagnostics only (ogl_x3_debug_stamps): per block {s_memtime, s_memrealtime} at entry and exit
};

// PROBE builds (diagnostics, tools/gemm_x3_bench.py clock): s_memtime stamps around the segments of every step, summed per wave
#define X3_STAMP(t)                                                                      \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  } while (0)

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) — indices into register arrays stay constants
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>());
    static_for<B + 1, E>(f);
  }
}

__device__ float4 g_x3_trash[64];   // where epilogue lanes with nothing to store aim their (statically counted) stores

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// TM, TN: the wave tile in units of 32 rows / columns; SPREAD: the next stage's DMA pieces go out between the MFMA groups
// of the first half of a step instead of in one burst at its top.
template <int WAVES_M, int WAVES_N, int TM, int TN, bool SPREAD>
__global__ void __launch_bounds__(WAVES_M * WAVES_N * 64) k_gemm_x3(X3Args g) {
  constexpr int RB = TM * 2, CB = TN * 2;   // 16 x 16 MFMA row / column blocks of a wave tile
  constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32, ROWS = BM + BN;
  constexpr int PIECES = ROWS * 12;
  static_assert(PIECES % NT == 0, "every thread issues the same number of DMA pieces");
  constexpr int NLOAD = PIECES / NT;
  constexpr int STAGE = PIECES * 16;
  // the first BM * 12 / NT pieces a thread moves are A pieces, the rest B pieces (static)
  constexpr int NLOAD_A = BM * 12 / NT;
  static_assert(BM * 12 % NT == 0, "A / B pieces split on a load boundary");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the LDS base of every DMA piece is scalar
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int l15 = lane & 15, quad = lane >> 4;
  // row swizzle of the 16-byte chunk index inside a plane: a bijection of (r >> 2) & 3 chosen so that the 16x16x32
  // fragment (lane -> row l & 15, chunk l >> 4) reads conflict-free (the plain (r >> 2) & 3 does not: a 16-lane
  // ds_read_b128 group holds rows 0-3, 12-15 at chunk c and rows 4-11 at chunk c ^ 1)
  auto swz = [](int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; };   // 0, 2, 3, 1

  // PERSISTENT blocks, XCD-aware and bijective.  Hardware deals block L to XCD L % 8; the logical tile space (split,
  // row panel, column tile — column tile fastest) is cut into 8 contiguous chunks and the blocks of XCD c walk chunk
  // c with stride gridDim / 8: tiles that run at the same time on one XCD share the A row panel (and, for split-K,
  // the same slice of both operands), so each XCD's L2 fetches a panel once.  One block per CU (144 KB LDS).
  const int T = g.NI * g.NJ * g.nsplit;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk_begin = xcd * (T >> 3) + min(xcd, T & 7), chunk_len = (T >> 3) + (xcd < (T & 7) ? 1 : 0);
  if (slot >= chunk_len) return;
  if (g.stamps && tid == 0) {
    g.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
    g.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }

  struct Tile { int64_t i0, j0; int split, ks_begin, ks_end; };
  auto decode = [&](int logical) {
    Tile t;
    t.split = logical / (g.NI * g.NJ);
    const int tile = logical - t.split * (g.NI * g.NJ);
    const int ti = tile / g.NJ, tj = tile - ti * g.NJ;
    t.i0 = (int64_t)ti * BM; t.j0 = (int64_t)tj * BN;
    t.ks_begin = 0; t.ks_end = g.nsteps;
    if (g.nsplit > 1) {
      t.ks_begin = t.split * g.steps_per_split;
      t.ks_end = min(g.nsteps, t.ks_begin + g.steps_per_split);
    }
    return t;
  };

  // gathered row ids of a tile's A pieces (independent loads, clamped index; validity is applied in make_src)
  auto load_rids = [&](const Tile& t, int64_t (&rid)[NLOAD_A]) {
    if (g.a.rows) {
#pragma unroll
      for (int u = 0; u < NLOAD_A; ++u) {
        const int64_t gi = t.i0 + (u * NT + tid) / 12;
        rid[u] = g.a.rows[gi < g.M ? gi : g.M - 1];   // raw: nothing here consumes the loaded value
      }
    } else {
#pragma unroll
      for (int u = 0; u < NLOAD_A; ++u) rid[u] = t.i0 + (u * NT + tid) / 12;
    }
  };
  // per-lane DMA sources: piece i = u * NT + tid of a stage = (row r = i / 12, plane p, swizzled chunk c)
  const unsigned char* src[NLOAD];
  auto make_src = [&](const Tile& t, const int64_t (&rid)[NLOAD_A]) {
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      const int i = u * NT + tid;
      const int r = i / 12, jp = i - r * 12;                 // physical piece jp of LDS row r holds the row's logical piece
      const int j = (jp & ~3) | ((jp & 3) ^ swz(r));         // j (its low two bits swizzled): byte offset 16 j in the group
      const unsigned char* rowp;
      if (u < NLOAD_A) {
        const int64_t id = rid[u < NLOAD_A ? u : 0];
        const bool ok = t.i0 + r < g.M && id >= 0 && id < g.a.nrows;
        rowp = g.a.img + (ok ? id : g.a.zero_row) * g.a.row_bytes + (int64_t)t.ks_begin * g.a.step_bytes;
      } else {
        const int64_t gj = t.j0 + (r - BM);
        rowp = g.b.img + (gj < g.N ? gj : g.b.zero_row) * g.b.row_bytes + (int64_t)t.ks_begin * g.b.step_bytes;
      }
      src[u] = rowp + j * 16;
    }
  };

  // per-lane fragment offsets inside a stage (bytes): row base + the (swizzled) piece of the lane's 8-element chunk
  // `quad` in plane p.  The swizzle depends on (r >> 2) & 3, which is the same for every 16-row block of a lane (block
  // bases are multiples of 16), so the three plane offsets are per-lane constants shared by all A and B row blocks.
  int offa[RB], offb[CB], offp[3];
  {
    const int q = swz(l15);
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      const int j = x3_piece(quad, sp);
      offp[sp] = ((j & ~3) | ((j & 3) ^ q)) * 16;
    }
  }
#pragma unroll
  for (int t = 0; t < RB; ++t) offa[t] = (wm * TM * 32 + t * 16 + l15) * 192;
#pragma unroll
  for (int t = 0; t < CB; ++t) offb[t] = (BM + wn * TN * 32 + t * 16 + l15) * 192;

  // accumulators: C^T tiles (the weight-side fragment is the MFMA's first operand), so a lane holds a 4-column group
  // of ONE output row: row l & 15, columns 4 (l >> 4) + (0..3) of the 16 x 16 block
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[RB][CB];

  auto issue = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      __builtin_amdgcn_global_load_lds((gptr_t)src[u], (lptr_t)(smem + buf * STAGE + (u * NT + wid * 64) * 16), 16, 0, 0);
      src[u] += u < NLOAD_A ? g.a.step_bytes : g.b.step_bytes;
    }
  };

  // One 32-deep step = RB x CB x 6 v_mfma_f32_16x16x32_bf16 (one MFMA = one product term over the whole step; same
  // cycles per flop as 32x32x16, but the chip holds a higher clock under it: +8 %).  A fragments first, B blocks streamed
  // one ahead (two-deep ring).  Per accumulator the six terms go smallest first (i + j = 4, then 3, then 2) — as in
  // k_gemm's x6 path.  fetch_buf >= 0 (SPREAD): the DMA pieces of the NEXT stage go out behind the MFMA groups of the
  // first half of the step — a DMA instruction holds its wave for 40-300 cycles (the CU's address path takes one per
  // ~38 cycles when all eight waves feed it); issued in one burst at the top of the step they idle the matrix pipe,
  // spread out the SIMD's other wave multiplies meanwhile.
  auto compute = [&](int buf, int fetch_buf) __attribute__((always_inline)) {
    const unsigned char* st = smem + buf * STAGE;
    bf16x8 a[RB][3], b[2][3];
#pragma unroll
    for (int t = 0; t < RB; ++t)
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) a[t][sp] = *(const bf16x8*)(st + offa[t] + offp[sp]);
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) b[0][sp] = *(const bf16x8*)(st + offb[0] + offp[sp]);
    constexpr int NG = RB * CB, NGI = NG / 2;   // DMA slots: the MFMA groups of the first half of the step
    static_for<0, CB>([&](auto yc) __attribute__((always_inline)) {
      constexpr int y = decltype(yc)::value;
      if constexpr (y + 1 < CB) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) b[(y + 1) & 1][sp] = *(const bf16x8*)(st + offb[y + 1] + offp[sp]);
      }
      static_for<0, RB>([&](auto xc) __attribute__((always_inline)) {
        constexpr int x = decltype(xc)::value, gi = y * RB + x;
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][2], a[x][0], acc[x][y], 0, 0, 0);
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][1], a[x][1], acc[x][y], 0, 0, 0);
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][0], a[x][2], acc[x][y], 0, 0, 0);
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][1], a[x][0], acc[x][y], 0, 0, 0);
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][0], a[x][1], acc[x][y], 0, 0, 0);
        acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][0], a[x][0], acc[x][y], 0, 0, 0);
        // the pieces u with u * NGI / NLOAD == gi go out behind this group
        static_for<0, NLOAD>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value;
          if constexpr (u * NGI / NLOAD == gi) {
            if (fetch_buf >= 0) {
              __builtin_amdgcn_sched_barrier(0);
              __builtin_amdgcn_global_load_lds((gptr_t)src[u], (lptr_t)(smem + fetch_buf * STAGE + (u * NT + wid * 64) * 16), 16, 0, 0);
              src[u] += u < NLOAD_A ? g.a.step_bytes : g.b.step_bytes;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      });
    });
  };

  // Epilogue of a finished tile: every lane owns 4-column groups of one output row (see the accumulator layout).
  // Every thread issues EXACTLY NSTORE = TM * TN * 4 16-byte stores, whatever the tile's position: lanes with nothing
  // to store (rows >= M, columns >= N, unaligned destinations) aim theirs at a scratch line.  The static count is what
  // lets the next tile's first barrier wait for its DMA only (s_waitcnt vmcnt(NSTORE)) while these stores drain.
  auto epilogue = [&](const Tile& t) {
    float* const dst = g.nsplit > 1 ? g.ws + (int64_t)t.split * g.M * g.ws_ld : g.C;
    const int64_t ldd = g.nsplit > 1 ? g.ws_ld : g.ldc;
    const bool vec_ok = (ldd & 3) == 0 && ((uintptr_t)dst & 15) == 0;
    const bool fin = g.nsplit == 1;
    float* const trash = (float*)&g_x3_trash[lane];
    auto store_group = [&](int64_t row, int64_t col, float v0, float v1, float v2, float v3) {
      float v[4] = {v0, v1, v2, v3};
      const bool rok = row < g.M;
      const bool has_oc = fin && g.ones_col && col + 3 >= g.N - 1 && col < g.N;   // this group holds the ones column
      if (fin && g.relu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      // the 16-byte store: whole groups, and partial groups whose tail falls into the row's pad columns
      const bool vec = rok && vec_ok && !has_oc && col < g.N && col + 4 <= ldd;
      *(float4*)(vec ? dst + row * ldd + col : trash) = make_float4(v[0], v[1], v[2], v[3]);
      if (rok && !vec && col < g.N) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (col + c >= g.N) continue;
          if (fin && g.ones_col && col + c == g.N - 1) { if (g.db) g.db[row] = v[c]; }
          else dst[row * ldd + col + c] = v[c];
        }
      }
    };
#pragma unroll
    for (int x = 0; x < RB; ++x)
#pragma unroll
      for (int y = 0; y < CB; ++y)
        store_group(t.i0 + wm * TM * 32 + x * 16 + l15, t.j0 + wn * TN * 32 + y * 16 + 4 * quad,
                    acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]);
  };

  // Two-stage ring that runs ACROSS tiles.  The barrier at the top of a step (a) retires this wave's DMA of the
  // stage about to be multiplied (vmcnt(0) before s_barrier) and makes every wave's pieces visible, (b) guarantees
  // all waves finished reading the other buffer, which the next DMA overwrites.  The last step of a tile already
  // fetches the first stage of the block's next tile, so the pipeline fill (row-id gather, first DMA) of every
  // tile but the first is covered by matrix work, and the epilogue stores drain under the fill's tail.
  int cur = chunk_begin + slot;
  Tile tc = decode(cur);
  int buf = 0;
  bool first_tile = true;
  constexpr int NSTORE = RB * CB;
  {
    int64_t rid[NLOAD_A];
    load_rids(tc, rid);
    make_src(tc, rid);
  }
  issue(0);
  while (true) {
    const int nxt = cur + nslots;
    const bool has_next = nxt - chunk_begin < chunk_len;
    Tile tn = tc;
    int64_t rid_next[NLOAD_A];
    if (has_next) tn = decode(nxt);
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int b = 0; b < CB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;
    for (int ks = tc.ks_begin; ks < tc.ks_end; ++ks) {
      if (ks == tc.ks_begin && !first_tile) {
        // the previous tile's NSTORE epilogue stores were issued AFTER this stage's DMA: leave them in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
        __builtin_amdgcn_s_barrier();
      } else {
        __syncthreads();
      }
      // row ids of the NEXT tile: fetched after this tile's first barrier (behind the previous epilogue's stores, which
      // nothing waits for any more), consumed at the last step
      if (ks == tc.ks_begin && has_next) load_rids(tn, rid_next);
      const bool more = ks + 1 < tc.ks_end;
      if (!more && has_next) make_src(tn, rid_next);
      if constexpr (!SPREAD) {
        if (more || has_next) issue(buf ^ 1);
        compute(buf, -1);
      } else {
        compute(buf, more || has_next ? buf ^ 1 : -1);
      }
      buf ^= 1;
    }
    epilogue(tc);
    if (!has_next) {
      if (g.stamps && tid == 0) {
        g.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        g.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
      }
      break;
    }
    cur = nxt; tc = tn; first_tile = false;
  }
}

// ---- the WIDE tile: 256 x 320 outputs on 16-deep half-steps, halves of the block alternating between multiplying and
// loading ------------------------------------------------------------------------------------------------------------
// What bounds k_gemm_x3 is the ISSUE of its stage DMA: a wave gets one LDS-DMA instruction
// accepted per 60-300 cycles and multiplies nothing meanwhile.  This kernel attacks both factors:
//  * 1.67 x fewer DMA pieces per flop: a 256 x 320 tile; to double-buffer 576 rows in 160 KB a stage is one 96-byte HALF
//    of a group (16 reduction elements): 2 x 57 KB.  The matrix instruction is therefore v_mfma_f32_32x32x16_bf16.
//  * the block's waves form two HALVES (waves 0-3 and 4-7: waves w and w + 4 share a SIMD) that alternate per segment —
//    two segments per half-step, a barrier after each: one half MULTIPLIES (60 MFMAs per wave; its A fragments sit in
//    registers, the ten 32-column B blocks are streamed from LDS one ahead), the other half LOADS: issues its DMA pieces
//    of a later stage (buffer_load_dwordx4 ... lds: a third cheaper to issue than global_load_lds with 64-bit lane
//    addresses), reads its next A fragments, and stores a finished tile.  With n = the block's running half-step and
//    stage n in LDS buffer n & 1:
//      half 0 multiplies step n in segment 2n,  loads in segment 2n+1: DMA(A rows of stage n+2), fragments of step n+1
//      half 1 loads in segment 2n: DMA(B rows of stage n+1), fragments of step n;  multiplies step n in segment 2n+1
//    Half 0 moves ONLY A rows and half 1 ONLY B rows: the B rows of buffer n & 1 are read (streamed) during both multiply
//    segments of step n and rewritten by half 1 in segment 2n+2 at the earliest; the A rows are read in the load segments
//    2n-1 / 2n and rewritten by half 0 from segment 2n+1 on.  A wave waits for its own DMA of stage n+1 before the
//    barrier that ends segment 2n (half 0 after multiplying — its pieces have had a whole segment — half 1 at the end of
//    the load segment that issued them).
// One wave = 32 rows x 320 columns of the tile (accumulators: 160 registers).  Images must be < 4 GB (32-bit offsets).
template <bool PROBE>
__global__ void __launch_bounds__(512) k_gemm_x3w(X3Args g) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the body uses device-only buffer builtins)
  constexpr int BM = 256, BN = 320, NYB = BN / 32;
  constexpr int NLA = BM * 6 / 256;                       // 6 A pieces per thread of half 0 per stage
  constexpr int NLB = (BN * 6 + 255) / 256;               // 8 B pieces per thread of half 1 (the last load half dead)
  constexpr int A_STAGE = BM * 96, B_STAGE = NLB * 256 * 16;   // bytes of one stage's A rows / B rows (+ dead pieces)
  constexpr int NA = 3, NB = 2;                           // ring depths: A rows come from HBM (2-3 us), B rows from L2
  constexpr int B_REGION = NA * A_STAGE;
  constexpr int NLH = NLA > NLB ? NLA : NLB;
  constexpr int NSTORE = NYB * 4;
  static_assert(NA * A_STAGE + NB * B_STAGE <= 160 * 1024, "the rings fit one CU");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NA * A_STAGE + NB * B_STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wid >> 2, ht = tid - h * 256;
  const int l31 = lane & 31, hi = lane >> 5;

  const int T = g.NI * g.NJ * g.nsplit;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int chunk_begin = xcd * (T >> 3) + min(xcd, T & 7), chunk_len = (T >> 3) + (xcd < (T & 7) ? 1 : 0);
  if (slot >= chunk_len) return;
  if (g.stamps && tid == 0) {
    g.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
    g.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  const int first = chunk_begin + slot, last_logical = chunk_begin + chunk_len;

  struct Tile { int ti, tj, split, hs_begin, hs_end; };   // hs_*: half-steps (two per 32-deep group)
  auto decode = [&](int logical) __attribute__((always_inline)) {
    Tile t;
    t.split = logical / (g.NI * g.NJ);
    const int tile = logical - t.split * (g.NI * g.NJ);
    t.ti = tile / g.NJ; t.tj = tile - t.ti * g.NJ;
    int kb = 0, ke = g.nsteps;
    if (g.nsplit > 1) {
      kb = t.split * g.steps_per_split;
      ke = min(g.nsteps, kb + g.steps_per_split);
    }
    t.hs_begin = 2 * kb; t.hs_end = 2 * ke;
    return t;
  };
  int total = 0;
  for (int l = first; l < last_logical; l += nslots) { const Tile t = decode(l); total += t.hs_end - t.hs_begin; }

  // ---- fetch side ----------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(h == 0 ? g.a.img : g.b.img), 0, 0xFFFFFFFF, 0x00020000);
  const unsigned step_bytes = (unsigned)(h == 0 ? g.a.step_bytes : g.b.step_bytes);
  unsigned src[NLH];
  auto load_rids = [&](const Tile& t, int64_t (&rid)[NLA]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NLA; ++u) {
      const int64_t gi = (int64_t)t.ti * BM + (u * 256 + ht) / 6;
      rid[u] = gi;
      if (g.a.rows) rid[u] = g.a.rows[gi < g.M ? gi : g.M - 1];       // raw: nothing here consumes the value
    }
  };
  auto make_src = [&](const Tile& t, const int64_t (&rid)[NLA]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NLH; ++u) {
      int i = u * 256 + ht;
      asm volatile("" : "+v"(i));                          // keep the piece geometry out of the always-live set
      const int r = i / 6, jp = i - r * 6;
      const int j = jp ^ ((r >> 3) & 1);                   // physical piece jp of LDS row r holds logical piece j of the half
      int64_t off;
      if (h == 0) {
        const int64_t id = rid[u < NLA ? u : 0];
        const bool ok = u < NLA && (int64_t)t.ti * BM + r < g.M && id >= 0 && id < g.a.nrows;
        off = (ok ? id : g.a.zero_row) * g.a.row_bytes + (int64_t)(t.hs_begin >> 1) * g.a.step_bytes;
      } else {
        const int64_t gj = (int64_t)t.tj * BN + r;
        off = (r < BN && gj < g.N ? gj : g.b.zero_row) * g.b.row_bytes + (int64_t)(t.hs_begin >> 1) * g.b.step_bytes;
      }
      src[u] = (unsigned)(off + j * 16);
    }
  };
  int f_logical = first, fhs, fhs_end, fstage = 0;
  bool f_more = true;
  {
    const Tile t = decode(first);
    int64_t rid[NLA];
    if (h == 0) load_rids(t, rid);
    make_src(t, rid);
    fhs = t.hs_begin; fhs_end = t.hs_end;
  }

  // ---- multiply side ---------------------------------------------------------------------------------------------------
  // fragment offsets inside a stage: row (96 bytes) + the swizzled piece 2 p + (lane >> 5) of plane p; the swizzle bit
  // (r >> 3) & 1 of a lane is the same in every 32-row block
  int offp[3];
#pragma unroll
  for (int sp = 0; sp < 3; ++sp) offp[sp] = ((2 * sp + hi) ^ ((l31 >> 3) & 1)) * 16;
  const int rowa = (wid * 32 + l31) * 96, rowb = l31 * 96;
  f32x16 acc[NYB];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int y = 0; y < NYB; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[y][e] = 0.f;
  };
  bf16x8 fa[3], fb[2][3];
  // the A fragments of step n (a load segment reads them ahead; the B blocks are all read inside the multiply)
  auto read_one = [&](int n, auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k < 3) fa[k] = *(const bf16x8*)(smem + (n % NA) * A_STAGE + rowa + offp[k]);
    else fb[0][k - 3] = *(const bf16x8*)(smem + B_REGION + (n & 1) * B_STAGE + rowb + offp[k - 3]);
  };
  auto read_frags = [&](int n) __attribute__((always_inline)) {
    static_for<0, 6>([&](auto kc) __attribute__((always_inline)) { read_one(n, kc); });
  };
  // fetch(): this thread's pieces of the next stage of ITS operand (half 0: A rows, half 1: B rows) into its ring slot;
  // read_n >= 0: the A fragment reads of step read_n are spread between the DMA instructions
  auto fetch = [&](int read_n) __attribute__((always_inline)) {
    if (!f_more) {
      if (read_n >= 0) read_frags(read_n);
      return;
    }
    const bool tile_ends = fhs + 1 == fhs_end, has_next = f_logical + nslots < last_logical;
    Tile tn = Tile();
    int64_t rid[NLA];
    if (tile_ends && has_next) {
      tn = decode(f_logical + nslots);
      if (h == 0) load_rids(tn, rid);
    }
    const unsigned adv = (fhs & 1) ? step_bytes - 96 : 96;  // after the second half of a group: on to the next group
    static_for<0, NLH>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if (u < NLA || h == 1) {
        const int slot_off = h == 0 ? (fstage % NA) * A_STAGE : B_REGION + (fstage & 1) * B_STAGE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(smem + slot_off + (u * 256 + (wid & 3) * 64) * 16), 16, src[u], 0, 0, 0);
        src[u] += adv;
      }
      if constexpr (u < 6) {
        if (read_n >= 0) read_one(read_n, uc);
      }
    });
    ++fstage; ++fhs;
    if (tile_ends) {
      if (!has_next) { f_more = false; return; }
      make_src(tn, rid);
      f_logical += nslots; fhs = tn.hs_begin; fhs_end = tn.hs_end;
    }
  };
  // one half-step: 10 column blocks x 6 v_mfma_f32_32x32x16_bf16, B block y + 1 read while block y multiplies.  The
  // weight-side fragment is the first operand: the accumulator holds C^T, a lane owns 4-column groups of ONE output row.
  auto multiply = [&](int n) __attribute__((always_inline)) {
    const unsigned char* st = smem + B_REGION + (n & 1) * B_STAGE + rowb;
    static_for<0, NYB>([&](auto yc) __attribute__((always_inline)) {
      constexpr int y = decltype(yc)::value;
      if constexpr (y + 1 < NYB) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) fb[(y + 1) & 1][sp] = *(const bf16x8*)(st + (y + 1) * 32 * 96 + offp[sp]);
      }
      acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[y & 1][2], fa[0], acc[y], 0, 0, 0);
      acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[y & 1][1], fa[1], acc[y], 0, 0, 0);
      acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[y & 1][0], fa[2], acc[y], 0, 0, 0);
      acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[y & 1][1], fa[0], acc[y], 0, 0, 0);
      acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[y & 1][0], fa[1], acc[y], 0, 0, 0);
      acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[y & 1][0], fa[0], acc[y], 0, 0, 0);
    });
  };
  // epilogue: lane -> output row l & 31, register e -> column (e & 3) + 8 (e >> 2) + 4 (lane >> 5); exactly NSTORE
  // 16-byte stores per thread (dead lanes aim at a scratch line), so that a wave can wait for the DMA it issued BEFORE them
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
    float* const dst = g.nsplit > 1 ? g.ws + (int64_t)t.split * g.M * g.ws_ld : g.C;
    const int64_t ldd = g.nsplit > 1 ? g.ws_ld : g.ldc;
    const bool vec_ok = (ldd & 3) == 0 && ((uintptr_t)dst & 15) == 0;
    const bool fin = g.nsplit == 1;
    float* const trash = (float*)&g_x3_trash[lane];
    const int64_t row = (int64_t)t.ti * BM + wid * 32 + l31;
    const bool rok = row < g.M;
#pragma unroll
    for (int y = 0; y < NYB; ++y)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t col = (int64_t)t.tj * BN + y * 32 + 8 * q + 4 * hi;
        float v[4] = {acc[y][4 * q], acc[y][4 * q + 1], acc[y][4 * q + 2], acc[y][4 * q + 3]};
        const bool has_oc = fin && g.ones_col && col + 3 >= g.N - 1 && col < g.N;
        if (fin && g.relu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
        }
        const bool vec = rok && vec_ok && !has_oc && col < g.N && col + 4 <= ldd;
        *(float4*)(vec ? dst + row * ldd + col : trash) = make_float4(v[0], v[1], v[2], v[3]);
        if (rok && !vec && col < g.N) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (col + c >= g.N) continue;
            if (fin && g.ones_col && col + c == g.N - 1) { if (g.db) g.db[row] = v[c]; }
            else dst[row * ldd + col + c] = v[c];
          }
        }
      }
  };
  // segment boundary: nothing — MFMAs included, which a memory clobber alone does not pin — may be scheduled across it
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };

  unsigned long long p_wait = 0, p_load = 0, p_mul = 0, p_vm = 0, ta = 0, tb = 0, tc = 0;   // PROBE only
  auto wait_keep = [&](bool fetch_behind, bool stores) __attribute__((always_inline)) {
    // wait for this wave's OLDER DMA; left in flight: the pieces of the one fetch issued after it (fetch_behind) and the
    // finished tile's stores issued after that (stores) — both static counts
    if (fetch_behind && stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA + NSTORE) : "memory");
    else if (fetch_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA) : "memory");
    else if (stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  zero_acc();
  // One loop per half; the step counter starts at -3: the pipeline fill runs through the same code, multiply switched off.
  //   half 0, step n: multiply(n) | wait A(n+1), leaving A(n+2) in flight | barrier | fetch A(n+3), A fragments of n+1 | barrier
  //   half 1, step n: fetch B(n+1), fragments of n, wait B(n+1) | barrier | multiply(n) | barrier
  // A ring of three (HBM latency: a stage has 3.5 segments to land), B ring of two (L2 hits: 1.5 segments).
  auto run = [&](auto half_tag) __attribute__((always_inline)) {
    constexpr int H = decltype(half_tag)::value;
    bool stores_pending = false;
    int m_logical = first, mhs = 0, mhs_end = 0, mhs_begin = 0;
    { const Tile t = decode(first); mhs = mhs_begin = t.hs_begin; mhs_end = t.hs_end; }
    for (int n = -3; n < total; ++n) {
      const bool live = n >= 0;
      const bool last = live && mhs + 1 == mhs_end;
      if constexpr (PROBE) X3_STAMP(ta);
      if (H == 0) {
        if (live) multiply(n);
        if constexpr (PROBE) { X3_STAMP(tb); p_mul += tb - ta; }
        // A(n+1) must have landed; A(n+2) (fetched one load segment ago, if it exists) may stay in flight
        wait_keep(n + 2 >= 0 && n + 2 < total, stores_pending);
        stores_pending = false;
        if constexpr (PROBE) { X3_STAMP(tc); p_vm += tc - tb; }
        barrier();
        if constexpr (PROBE) { X3_STAMP(ta); p_wait += ta - tb; }
        const bool rd = n + 1 >= 0 && n + 1 < total;
        fetch(rd && !last ? n + 1 : -1);                   // A rows of stage n + 3, A fragments of step n + 1
        if (last) {                                        // the finished tile goes out before the fragment registers fill
          epilogue(decode(m_logical));
          zero_acc();
          stores_pending = true;
          if (rd) read_frags(n + 1);
        }
        if constexpr (PROBE) { X3_STAMP(tb); p_load += tb - ta; }
        barrier();
        if constexpr (PROBE) { X3_STAMP(ta); p_wait += ta - tb; }
      } else {
        const bool prev_tile = live && mhs == mhs_begin && m_logical != first;
        if (n >= -1) fetch(live && !prev_tile ? n : -1);   // B rows of stage n + 1, A fragments of step n
        if (prev_tile) {                                   // the tile BEFORE step n's tile: its stores go out behind the DMA
          epilogue(decode(m_logical - nslots));
          zero_acc();
          read_frags(n);
          stores_pending = true;
        }
        if constexpr (PROBE) X3_STAMP(tc);
        wait_keep(false, stores_pending);                  // B(n+1), issued at the top of this segment: half 0 reads it next
        stores_pending = false;
        if constexpr (PROBE) { unsigned long long td; X3_STAMP(td); p_vm += td - tc; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // half 0 refills this stage's A slot two segments on
        if constexpr (PROBE) { X3_STAMP(tb); p_load += tb - ta; }
        barrier();
        if constexpr (PROBE) { X3_STAMP(ta); p_wait += ta - tb; }
        if (live) multiply(n);
        if constexpr (PROBE) { X3_STAMP(tb); p_mul += tb - ta; }
        barrier();
        if constexpr (PROBE) { X3_STAMP(ta); p_wait += ta - tb; }
      }
      if (live && ++mhs == mhs_end && n + 1 < total) {
        m_logical += nslots;
        const Tile t = decode(m_logical);
        mhs = mhs_begin = t.hs_begin; mhs_end = t.hs_end;
      }
    }
    if (H == 1) epilogue(decode(m_logical));               // the block's last tile
  };
  if (h == 0) run(std::integral_constant<int, 0>());
  else run(std::integral_constant<int, 1>());
  if (g.stamps && tid == 0) {
    g.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    g.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (PROBE) {
    if (g.stamps && lane == 0) {
      unsigned long long* o = g.stamps + 1024 + (blockIdx.x * 8 + wid) * 4;
      o[0] = p_wait; o[1] = p_load; o[2] = p_mul; o[3] = p_vm;
    }
  }
#endif
}

__global__ void __launch_bounds__(256) k_x3_splitk_reduce(X3Args g) {
  const int64_t total = g.M * g.N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / g.N, col = t - row * g.N;
    float v = 0.f;
    for (int s = 0; s < g.nsplit; ++s) v += g.ws[((int64_t)s * g.M + row) * g.ws_ld + col];  // fixed order
    const bool oc = g.ones_col && col == g.N - 1;
    if (g.relu) v = fmaxf(v, 0.f);
    if (oc) { if (g.db) g.db[row] = v; }
    else g.C[row * g.ldc + col] = v;
  }
}

// ---- image builders ---------------------------------------------------------------------------------------------------
// image[r][k/32][plane][k%32] = split(src[row(r), k]); one thread per (row, 8 consecutive k).  Row R (all zeros) is
// written too.  `append`: the image carries ONE extra reduction element k = K — 1.0 in every row including the zero
// row (append = 1: the activations side) or append_vec[r] (append = 2: the weights side, i.e. the bias) — so that a
// product of two such images is x . w^T + bias with the bias added by the matrix pipe (no bias pass in the epilogue).
__global__ void __launch_bounds__(256) k_x3_split(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ rows,
                                                  int64_t nrows_src, int64_t R, int K, int append,
                                                  const float* __restrict__ append_vec, unsigned char* __restrict__ img,
                                                  int64_t row_bytes) {
  const int cpr = (int)(row_bytes / X3_GROUP_BYTES) * 4;   // 8-element chunks per row
  const int64_t total = (R + 1) * cpr;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / cpr;
    const int ch = (int)(t - r * cpr);
    const int k = ch * 8;
    float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool ok = r < R;
    int64_t row = r;
    if (ok && rows) { row = rows[r]; ok = row >= 0 && row < nrows_src; }
    if (ok && k < K) {
      const float* p = src + row * ld + k;
      if (k + 8 <= K) {
        const float4 lo = ld16(p), hi = ld16(p + 4);
        e[0] = lo.x; e[1] = lo.y; e[2] = lo.z; e[3] = lo.w; e[4] = hi.x; e[5] = hi.y; e[6] = hi.z; e[7] = hi.w;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) if (k + q < K) e[q] = p[q];
      }
    }
    if (append && K >= k && K < k + 8) {
      const float av = append == 1 ? 1.f : (r < R ? append_vec[r] : 0.f);
#pragma unroll
      for (int q = 0; q < 8; ++q) if (k + q == K) e[q] = av;
    }
    uint4 o[3];
    split3(e[0], e[1], o[0].x, o[1].x, o[2].x);
    split3(e[2], e[3], o[0].y, o[1].y, o[2].y);
    split3(e[4], e[5], o[0].z, o[1].z, o[2].z);
    split3(e[6], e[7], o[0].w, o[1].w, o[2].w);
    unsigned char* d = img + r * row_bytes + (int64_t)(ch >> 2) * X3_GROUP_BYTES;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) *(uint4*)(d + x3_piece(ch & 3, sp) * 16) = o[sp];
  }
}

// image of the TRANSPOSE: image row n, reduction index m, stored GROUP-MAJOR (the rows of one 32-deep reduction step are
// contiguous: one tile of this kernel writes two 12 KB runs, and a GEMM stage reads one run):
// image[m/32][n][plane][m%32] = split(src[row(m), n]), 64 x 64 tiles through LDS.  ones_row: image row N is 1.0 for m < M (bias gradient operand).  Pad m >= M is zero.
__global__ void __launch_bounds__(256) k_x3_split_t(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ rows,
                                                    int64_t nrows_src, int64_t M, int N, int ones_row, int64_t G_il, int64_t Mi,
                                                    unsigned char* __restrict__ img, int64_t row_bytes) {
  __shared__ float tile[64][65];
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  const int tid = threadIdx.x, ty = tid >> 4, tx = (tid & 15) * 4;
  const int G = (int)(row_bytes / X3_GROUP_BYTES);
  const int64_t zero_row = (int64_t)N + (ones_row ? 1 : 0);
  const int64_t gstride = (zero_row + 1) * X3_GROUP_BYTES;       // GROUP-MAJOR image: [group][image row][192 B]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t ip = m0 + ty + 16 * k;                          // position in the image's reduction index
    const int64_t i = G_il ? (ip & 31) * G_il + (ip >> 5) : ip;   // the source row it stands for (round-robin dealing)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ip < Mi && i < M) {
      const int64_t row = rows ? rows[i] : i;
      if (!rows || (row >= 0 && row < nrows_src)) {
        const float* p = src + row * ld + n0 + tx;
        if (n0 + tx + 3 < N) v = ld16(p);
        else {
          if (n0 + tx < N) v.x = p[0];
          if (n0 + tx + 1 < N) v.y = p[1];
          if (n0 + tx + 2 < N) v.z = p[2];
        }
      }
    }
    tile[ty + 16 * k][tx] = v.x; tile[ty + 16 * k][tx + 1] = v.y; tile[ty + 16 * k][tx + 2] = v.z; tile[ty + 16 * k][tx + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u = tid + 256 * k;                 // unit = (n local, group of the tile, 8-element chunk)
    const int c = u & 3, gl = (u >> 2) & 1, nl = u >> 3;
    const int n = n0 + nl;
    const int64_t grp = m0 / 32 + gl;
    if (n >= N || grp >= G) continue;
    const int ml = gl * 32 + c * 8;
    uint4 o[3];
    split3(tile[ml + 0][nl], tile[ml + 1][nl], o[0].x, o[1].x, o[2].x);
    split3(tile[ml + 2][nl], tile[ml + 3][nl], o[0].y, o[1].y, o[2].y);
    split3(tile[ml + 4][nl], tile[ml + 5][nl], o[0].z, o[1].z, o[2].z);
    split3(tile[ml + 6][nl], tile[ml + 7][nl], o[0].w, o[1].w, o[2].w);
    unsigned char* d = img + grp * gstride + (int64_t)n * X3_GROUP_BYTES;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) *(uint4*)(d + x3_piece(c, sp) * 16) = o[sp];
  }
  if (blockIdx.y == 0 && tid >= 64 && tid < 64 + 24) {            // the zero row (last row of every group slab)
    const int t = tid - 64, gl = t / 12;
    const int64_t grp = m0 / 32 + gl;
    if (grp < G) *(uint4*)(img + grp * gstride + zero_row * X3_GROUP_BYTES + (t - gl * 12) * 16) = make_uint4(0, 0, 0, 0);
  }
  if (ones_row && blockIdx.y == 0 && tid < 8) {   // 8 chunks of 8 m: row N of the image
    const int c = tid & 3, gl = tid >> 2;
    const int64_t grp = m0 / 32 + gl;
    if (grp < G) {
      const int64_t mb = m0 + gl * 32 + c * 8;
      unsigned w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w[q] = (mb + 2 * q < Mi ? 0x3F80u : 0u) | (mb + 2 * q + 1 < Mi ? 0x3F800000u : 0u);   // bf16 1.0 pairs
      unsigned char* d = img + grp * gstride + (int64_t)N * X3_GROUP_BYTES;
      *(uint4*)(d + x3_piece(c, 0) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      *(uint4*)(d + x3_piece(c, 1) * 16) = make_uint4(0, 0, 0, 0);
      *(uint4*)(d + x3_piece(c, 2) * 16) = make_uint4(0, 0, 0, 0);
    }
  }
}

extern "C" int64_t ogl_x3_row_bytes(int64_t K) {
  if (K < 0) return OGL_EINVAL;
  return ogl_cdiv(K, 32) * X3_GROUP_BYTES;
}

extern "C" int64_t ogl_x3_image_bytes(int64_t rows, int64_t K) {
  if (rows < 0 || K < 0) return OGL_EINVAL;
  return (rows + 1) * ogl_cdiv(K, 32) * X3_GROUP_BYTES;
}

extern "C" int ogl_x3_split(const float* src, int64_t ld, const int64_t* rows, int64_t nrows_src, int64_t R, int K, int append,
                            const float* append_vec, void* image, ogl_stream_t stream) {
  if (R < 0 || K < 0 || ld < K || append < 0 || append > 2 || (append == 2 && R > 0 && !append_vec)) return OGL_EINVAL;
  const int Ki = K + (append ? 1 : 0);   // reduction length of the image
  if (Ki == 0) return OGL_OK;
  if (!image || (R > 0 && K > 0 && !src) || ((uintptr_t)image & 15)) return OGL_EINVAL;
  const int64_t row_bytes = ogl_cdiv(Ki, 32) * X3_GROUP_BYTES;
  const int64_t total = (R + 1) * (row_bytes / X3_GROUP_BYTES) * 4;
  hipLaunchKernelGGL(k_x3_split, dim3((unsigned)min((int64_t)65536, ogl_cdiv(total, 256))), dim3(256), 0, (hipStream_t)stream, src,
                     ld, rows, nrows_src, R, K, append, append_vec, (unsigned char*)image, row_bytes);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

extern "C" int ogl_x3_split_t(const float* src, int64_t ld, const int64_t* rows, int64_t nrows_src, int64_t M, int N, int ones_row,
                              int64_t interleave, void* image, ogl_stream_t stream) {
  if (M < 0 || N < 0 || ld < N || interleave < 0 || (interleave > 0 && 32 * interleave < M)) return OGL_EINVAL;
  if (M == 0) return OGL_OK;
  if (!image || (N > 0 && !src) || ((uintptr_t)image & 15)) return OGL_EINVAL;
  const int64_t Mi = interleave ? 32 * interleave : M;            // reduction length of the image
  const int64_t row_bytes = ogl_cdiv(Mi, 32) * X3_GROUP_BYTES;
  dim3 grid((unsigned)ogl_cdiv(Mi, 64), (unsigned)(N > 0 ? ogl_cdiv(N, 64) : 1));
  hipLaunchKernelGGL(k_x3_split_t, grid, dim3(256), 0, (hipStream_t)stream, src, ld, rows, nrows_src, M, N, ones_row, interleave, Mi,
                     (unsigned char*)image, row_bytes);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// 0: 256 x 128 (4 x 2 waves of 64 x 64)   1: 128 x 128 (2 x 4 waves of 64 x 32)
static int x3_config(int64_t M, int64_t N) {
  const int64_t w0 = ogl_cdiv(M, 256) * 256 * ogl_cdiv(N, 128) * 128;
  const int64_t w1 = ogl_cdiv(M, 128) * 128 * ogl_cdiv(N, 128) * 128;
  return w1 * 10 < w0 * 9 ? 1 : 0;   // the wider wave tile unless it pads > 10 % more MFMA work
}

// Diagnostics: when set, every k_gemm_x3 launch writes per block {s_memtime, s_memrealtime} at entry and at exit (4 x u64
// per block, <= 256 blocks) into `buf` — the in-kernel shader clock is d(memtime) / d(memrealtime) x 100 MHz.
static unsigned long long* g_x3_stamps = nullptr;
static int g_x3_probe = 0;
extern "C" int ogl_x3_debug_stamps(void* buf, int probe) {
  g_x3_stamps = (unsigned long long*)buf;
  g_x3_probe = buf ? probe : 0;    // probe != 0: the wide kernel's stamped build (per-wave cycles of wait / load / multiply)
  return OGL_OK;
}

static int launch_x3(X3Args& g, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0) return OGL_OK;
  g.stamps = g_x3_stamps;
  // (row-major image: rows x row_bytes; group-major image: groups x step_bytes)
  const int64_t a_bytes = std::max((g.a.zero_row + 1) * g.a.row_bytes, (int64_t)g.nsteps * g.a.step_bytes);
  const int64_t b_bytes = std::max((g.b.zero_row + 1) * g.b.row_bytes, (int64_t)g.nsteps * g.b.step_bytes);
  const int cfg = x3_config(g.M, g.N);
  // k_gemm_x3w is EXPERIMENTAL and off by default: parity-green, but 10-18 % slower than the tiles below at the layer-0
  // shapes (DESIGN.md section 8-1 has the per-segment stamps).  OGL_X3_WIDE=1 routes every eligible product through it.
  static const char* wide_env = getenv("OGL_X3_WIDE");
  const bool wide = wide_env && wide_env[0] == '1' && a_bytes < (1ll << 32) && b_bytes < (1ll << 32);
  if (wide) {
    g.NI = (int)ogl_cdiv(g.M, 256);
    g.NJ = (int)ogl_cdiv(g.N, 320);
    const int64_t T = (int64_t)g.NI * g.NJ * g.nsplit;
    dim3 grid((unsigned)(8 * std::min<int64_t>(32, ogl_cdiv(T, 8)))), block(512);
    if (g_x3_probe) hipLaunchKernelGGL((k_gemm_x3w<true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((k_gemm_x3w<false>), grid, block, 0, stream, g);
    OGL_CHECK_LAUNCH();
  } else {
  const int BM = cfg == 0 ? 256 : 128, BN = 128;
  g.NI = (int)ogl_cdiv(g.M, BM);
  g.NJ = (int)ogl_cdiv(g.N, BN);
  const int64_t T = (int64_t)g.NI * g.NJ * g.nsplit;
  dim3 grid((unsigned)(8 * std::min<int64_t>(32, ogl_cdiv(T, 8)))), block(512);   // persistent: at most one block per CU
  // DMA issue: spread between the MFMA groups for the 256 x 128 tile (2-3 % faster, A/B on one device), in one burst at
  // the top of the step for the 128 x 128 tile (its steps are too short to hide a late piece: spread measured 9 % slower)
  if (cfg == 0) hipLaunchKernelGGL((k_gemm_x3<4, 2, 2, 2, true>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((k_gemm_x3<2, 4, 2, 1, false>), grid, block, 0, stream, g);
  OGL_CHECK_LAUNCH();
  }
  if (g.nsplit > 1) {
    hipLaunchKernelGGL(k_x3_splitk_reduce, dim3((unsigned)min((int64_t)2048, ogl_cdiv(g.M * g.N, 256))), dim3(256), 0, stream, g);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
}

extern "C" int ogl_linear_fwd_x3(const void* x_img, int64_t x_img_rows, const int64_t* x_rows, int64_t x_nrows, int64_t M,
                                 int K, const void* w_img, int N, int relu, float* y, int64_t ldy, ogl_stream_t stream) {
  if (M < 0 || K <= 0 || N < 0 || x_img_rows < 0 || x_nrows < 0 || x_nrows > x_img_rows || ldy < N) return OGL_EINVAL;
  if (M == 0 || N == 0) return OGL_OK;
  if (!x_img || !w_img || !y || (!x_rows && M > x_img_rows)) return OGL_EINVAL;
  X3Args g = X3Args();
  const int64_t rb = ogl_cdiv(K, 32) * X3_GROUP_BYTES;
  g.a = X3Operand{(const unsigned char*)x_img, rb, X3_GROUP_BYTES, x_rows, x_rows ? x_nrows : x_img_rows, x_img_rows};
  g.b = X3Operand{(const unsigned char*)w_img, rb, X3_GROUP_BYTES, nullptr, N, N};
  g.M = M; g.N = N; g.nsteps = (int)ogl_cdiv(K, 32);
  g.C = y; g.ldc = ldy; g.relu = relu; g.nsplit = 1;
  return launch_x3(g, (hipStream_t)stream);
}

// split plan of the weight gradient: one round of blocks (one 8-wave block per CU), >= 8 steps per block
static void x3_bww_plan(int64_t M, int N, int K, int* nsplit, int* sps) {
  const int64_t steps = ogl_cdiv(M, 32);
  const int cfg = x3_config(N, K + 1);
  const int64_t tiles = ogl_cdiv(N, cfg == 0 ? 256 : 128) * ogl_cdiv(K + 1, 128);
  int64_t s = 256 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  if (steps / s < 8) s = steps / 8 > 0 ? steps / 8 : 1;
  *sps = (int)ogl_cdiv(steps > 0 ? steps : 1, s);
  *nsplit = (int)ogl_cdiv(steps > 0 ? steps : 1, *sps);
}

extern "C" int64_t ogl_linear_bwd_weight_x3_workspace_bytes(int64_t M, int N, int K) {
  if (M < 0 || N < 0 || K < 0) return OGL_EINVAL;
  int nsplit, sps;
  x3_bww_plan(M, N, K, &nsplit, &sps);
  if (nsplit <= 1) return 16;
  return (int64_t)nsplit * N * ogl_round_up(K + 1, 4) * 4;
}

extern "C" int ogl_linear_bwd_weight_x3(const void* dyT_img, const void* xT_img, int64_t M, int N, int K, float* dw,
                                        int64_t lddw, float* db, void* workspace, int64_t workspace_bytes,
                                        ogl_stream_t stream) {
  if (M <= 0 || N < 0 || K < 0 || lddw < K) return OGL_EINVAL;
  if (N == 0) return OGL_OK;
  if (!dw || !dyT_img || !xT_img) return OGL_EINVAL;
  X3Args g = X3Args();
  // transposed images are GROUP-MAJOR: [group][image row][192 B] (rows of one reduction step are contiguous)
  g.a = X3Operand{(const unsigned char*)dyT_img, X3_GROUP_BYTES, ((int64_t)N + 1) * X3_GROUP_BYTES, nullptr, N, N};
  g.b = X3Operand{(const unsigned char*)xT_img, X3_GROUP_BYTES, ((int64_t)K + 2) * X3_GROUP_BYTES, nullptr, (int64_t)K + 1,
                  (int64_t)K + 1};   // row K = the all-ones row
  g.M = N; g.N = K + 1; g.ones_col = 1; g.nsteps = (int)ogl_cdiv(M, 32);
  g.C = dw; g.ldc = lddw; g.db = db;
  x3_bww_plan(M, N, K, &g.nsplit, &g.steps_per_split);
  if (g.nsplit > 1) {
    g.ws_ld = ogl_round_up(K + 1, 4);
    if (!workspace || workspace_bytes < (int64_t)g.nsplit * N * g.ws_ld * 4) return OGL_EWORKSPACE;
    g.ws = (float*)workspace;
  }
  return launch_x3(g, (hipStream_t)stream);
}
