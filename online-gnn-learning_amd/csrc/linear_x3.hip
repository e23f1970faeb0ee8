// Dense projections on PRE-SPLIT operands ("bf16x3 images"): the same split-bf16 x6 arithmetic as linear.hip
// (x6_arith.h), but the exact 3-term bf16 split of every fp32 operand is done ONCE by a streaming kernel instead of
// by every GEMM block that touches the element.  What that buys on gfx950: the GEMM's staging becomes a pure byte
// copy, so tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no VALU, no ds_write)
// and the matrix pipe only waits for ds_read_b128 fragments.  Replaces the same torch.nn.Linear call sites as
// linear.hip (R/train/graphsage/pytorch/aggregator_dgl.py:85-94,171,181,206) for the large layer-0 products.
//
// Image of an fp32 matrix X[R, K] (reduction index contiguous): R rows + ONE all-zero row (index R), each row
// G = ceil(K/32) groups of 192 bytes; group g holds k = 32g .. 32g+31 as three 64-byte planes (hi, mid, lo terms
// of the split; x3_piece, x6_arith.h), pad k >= K is zero.  One (row, group) is therefore 192 contiguous bytes: a 32-deep
// GEMM step reads 12 consecutive 16-byte pieces per row.
//
// Two kernels compute C[i, j] = epilogue(sum_k A[i, k] B[j, k]) on images (A optionally gathered by an int64 row list:
// rows outside the table read the zero row), persistent, one block per CU, LDS stages of (BM + BN) x 192 B:
//   k_gemm_x3p (the default): PRODUCER / CONSUMER — eight multiplier waves that never issue a load + four mover waves
//               that issue the stage DMA (MUBUF, 32-bit offsets: images < 4 GB); matrix pipe busy 73 % of the cycles;
//   k_gemm_x3  (images >= 4 GB): eight waves that fetch for themselves (64-bit global_load_lds); 56 %.
// Both share the LDS image of a stage = the 12 pieces of row r at pieces 12r .. 12r+11 with
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
  int force_cfg0;               // the k-major weight gradient on the 256 x 128 tile (x3_bwwk_cfg0)
  int defer_reduce;             // host side only: launch_x3 leaves the split-K slabs unreduced (the optimiser launch sums them)
  int xcd_slabs;                // k_gemm_x3p: deal WHOLE split-K slabs to the XCDs (see decode)
  unsigned long long* stamps;   // diagnostics only (ogl_x3_debug_stamps): per block {s_memtime, s_memrealtime} at entry and exit
  // ---- extensions, k_gemm_x3p<..., EXT = true> only (forward products, nsplit == 1) ----
  X3Operand a2;                 // optional SECOND part of the A operand: reduction steps [nsteps1, nsteps) read a2 (its own image,
  int nsteps1;                  //   gather and zero row) — fc_self(x[dst]) + fc_neigh(neigh) as ONE product over a K-concatenated B
  const float* add;             // optional per-row addend: C[i, :] += add[add_rows ? add_rows[i] : i, :] before the activation
  int64_t ld_add; const int64_t* add_rows; int64_t add_nrows;
  const float* mask;            // optional: y[i, :] = (mask[i, :] > 0) ? y[i, :] : 0 after the addend — the ReLU backward of the layer
  int64_t ld_mask;              //   whose output `mask` is, applied to the input gradient this product computes (M rows, N columns)
  const unsigned char* y_keep;  // optional [M]: the fp32 output row i is stored only where y_keep[i] != 0 (its image always is)
  float* db2;                   // optional second copy of the bias gradient (the two biases of a dual projection: one tensor each)
  // ---- k_gemm_x3p<..., BK = true> only: the B operand is a ROW-MAJOR image whose ROWS are the reduction index (b.rows gathers them,
  // b.nrows bounds the ids, b.zero_row / b.row_bytes as for a row-major A) and whose COLUMNS are the output columns: the
  // weight-gradient products read the activations' images as they are (no transposed image of x) ----
  int64_t bk_red;               // reduction positions s >= bk_red are padding (zero row)
  int64_t bk_interleave;        // G > 0: reduction index m stands for position (m % 32) * G + m / 32 (the order of pool_bwd_x3's image)
  int bk_groups;                // 32-column groups per image row (column groups past it read the zero row)
  // ---- k_gemm_x3p<..., BK, AK = true>: the A operand too is a ROW-MAJOR image over the reduction (dy as its producer wrote it): a.img /
  // a.row_bytes / a.zero_row as for a row-major image, reduction position p = its row p (no gather, no dealing) ----
  int ak_groups;                // 32-column groups per row of the A image
  unsigned char* out_img;       // optional: ALSO write the bf16x3 image of the (activated) output, row-major, reduction length
  int64_t out_row_bytes;        //   N (+ 1 when out_append_ones: 1.0 at column N) — the A operand of the next layer's product
  int out_append_ones;
  // ---- k_gemm_x3p<..., BK, AK> only: a TWO-PART B operand by column tile — tiles tj >= NJ1 read the row-major image b2 (its own row
  // gather / zero row / column groups) where tiles tj < NJ1 read b: both weight gradients of a dual-input projection, dy^T . x[rows] and
  // dy^T . x2, as ONE product over [x[rows] | x2] without a concatenated image (output columns: part 1 at 0, part 2 at 128 NJ1) ----
  X3Operand b2; int bk2_groups; int NJ1;
  int stagger;                  // k_gemm_x3p: multiplier waves 4-7 run a step's LAST column block behind the next step's opening barrier
};

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

// Diagnostic build only (-DOGL_X3_PHASE_STAMPS, tools/x3_phase_probe.py; never in the product library): where a step of
// k_gemm_x3p goes.  A wave-uniform cycle stamp (s_memtime, waited for in place: the stamps sit where the wave has no LDS read in
// flight) at the arrival at / release from each of a step's barriers; per-phase SUMS over the block's steps stay in scalar registers
// and waves 0, 4 (the two multipliers of one SIMD) and 8 (a mover) write them once, behind the block's last step, to
// stamps[1024 + 32 blockIdx + 8 role ..].
#ifdef OGL_X3_PHASE_STAMPS
#define X3_PH_T(var) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); var = t_; } while (0)
#define X3_PH(...) __VA_ARGS__
#else
#define X3_PH_T(var)
#define X3_PH(...)
#endif

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
    int l15 = lane & 15, quad = lane >> 4, lane_e = lane;     // opaque copies: see k_gemm_x3p's epilogue
    asm volatile("" : "+v"(l15), "+v"(quad), "+v"(lane_e));
    float* const dst = g.nsplit > 1 ? g.ws + (int64_t)t.split * g.M * g.ws_ld : g.C;
    const int64_t ldd = g.nsplit > 1 ? g.ws_ld : g.ldc;
    const bool vec_ok = (ldd & 3) == 0 && ((uintptr_t)dst & 15) == 0;
    const bool fin = g.nsplit == 1;
    float* const trash = (float*)&g_x3_trash[lane_e];
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
      if (has_next && ks == max(tc.ks_begin, tc.ks_end - 2)) load_rids(tn, rid_next);
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

// ---- PRODUCER / CONSUMER form of the same tiles: 12 waves per block ---------------------------------------------------
// What idles the matrix pipe in k_gemm_x3 is the ISSUE of the stage DMA by the waves that should be multiplying (a wave
// gets one LDS-DMA instruction accepted per 60-300 cycles and issues in order).  Here waves 0-7 (two per SIMD) only
// multiply: fragments from LDS, RB x CB x 6 v_mfma_f32_16x16x32_bf16 per step, epilogue stores — they never issue a DMA
// instruction, so nothing holds them between barriers but the matrix pipe.  Waves 8-11 (one per SIMD) only move data:
// after the barrier that opens step n they issue ALL of stage n + 1 (MUBUF LDS-DMA: 18 pieces per lane for the 256 x 128
// tile) into the buffer step n - 1 just released, wait for them to land, and meet the multipliers at the next barrier.
// One barrier per step.  Three waves per SIMD = 168 registers per wave: the multipliers keep accumulators (64), A
// fragments (48) and a two-deep B ring (24), the movers their piece offsets.  Images must be < 4 GB (32-bit offsets).
// Measured against k_gemm_x3: matrix pipe busy 73 % of the in-kernel cycles instead of 56 %.
// EA ("early A", two-stage rings only): the multipliers hold a step's A fragments in registers from the step's first MFMA group on, so
// the A part of that stage buffer (2/3 of its bytes) is dead for the rest of the step.  A second barrier per step (`mid`), behind the
// first column block's MFMAs, hands it back to the movers, who issue the A part of stage n + 2 there and then — most of a step earlier
// than the B part of its stage.  Without it a step is one DMA latency long whatever the matrix pipe does (the movers issue stage
// n + 1 at the barrier that opens step n and wait for all of it): 2.25-2.33 us per 256 x 128 x 32 step against ~1.6 us of MFMA issue.
template <int WAVES_M, int WAVES_N, int TM, int TN, int NSTAGE, bool EXT = false, bool BK = false, bool AK = false, bool EA = false>
__global__ void __launch_bounds__(768) k_gemm_x3p(X3Args g) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(WAVES_M * WAVES_N == 8, "eight multiplier waves + four mover waves");
  constexpr int RB = TM * 2, CB = TN * 2, WROWS = RB * 16, WCOLS = CB * 16;
  constexpr int BM = WAVES_M * WROWS, BN = WAVES_N * WCOLS;
  constexpr int PIECES = (BM + BN) * 12, STAGE = PIECES * 16, A_PIECES = BM * 12;
  static_assert(PIECES % 256 == 0 && A_PIECES % 256 == 0, "a stage is a whole number of 256-lane instructions, each all-A or all-B");
  constexpr int NLP = PIECES / 256;                        // 18 pieces per mover lane per stage (256 x 128)
  constexpr int NLP_A = A_PIECES / 256;                    // the first 12 are A rows
  constexpr int NSTORE = RB * CB;
  static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
  static_assert(NSTAGE * STAGE <= 160 * 1024, "the ring fits one CU");
  static_assert(!BK || (BN == 128 && NLP - NLP_A == 6), "k-major B: 128 columns = 48 (plane, 8-column chunk) pairs x 32 rows per stage");
  static_assert(!(BK && EXT), "the k-major B operand belongs to the weight-gradient products");
  static_assert(!AK || (BK && BM == 128 && NLP_A == 6), "k-major A: with a k-major B, 128 rows");
  static_assert(!EA || (NSTAGE == 2 && !AK), "early A: a two-stage ring");
  constexpr int B_BASE = A_PIECES * 16;                    // byte offset of the B part inside a stage
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mover = wid >= 8;
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
  struct Tile { int ti, tj, split, ks_begin, ks_end; };
  // xcd_slabs: the tiles of one slab share their operands through ONE XCD's L2, so XCD x takes floor(nsplit / 8) whole slabs
  // (x q .. x q + q - 1) and the slabs left over are dealt tile by tile over what remains of every XCD's run
  const int tiles_per_slab = g.NI * g.NJ, q_slabs = g.nsplit >> 3;
  auto decode = [&](int logical) __attribute__((always_inline)) {
    Tile t;
    int tile;
    if (g.xcd_slabs) {
      const int local = logical - chunk_begin, whole = q_slabs * tiles_per_slab;
      if (local < whole) {
        t.split = xcd * q_slabs + local / tiles_per_slab;
        tile = local % tiles_per_slab;
      } else {
        const int e = (chunk_begin - xcd * whole) + (local - whole);
        t.split = 8 * q_slabs + e / tiles_per_slab;
        tile = e % tiles_per_slab;
      }
    } else {
      t.split = logical / tiles_per_slab;
      tile = logical - t.split * tiles_per_slab;
    }
    t.ti = tile / g.NJ; t.tj = tile - t.ti * g.NJ;
    t.ks_begin = 0; t.ks_end = g.nsteps;
    if (g.nsplit > 1) {
      t.ks_begin = t.split * g.steps_per_split;
      t.ks_end = min(g.nsteps, t.ks_begin + g.steps_per_split);
    }
    return t;
  };
  int total = 0;
  for (int l = first; l < last_logical; l += nslots) { const Tile t = decode(l); total += t.ks_end - t.ks_begin; }
  auto swz = [](int r) __attribute__((always_inline)) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; };
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };

  if (mover) {
    // ---- movers: the fetch cursor walks the block's stages in order, across tiles -------------------------------------
    const int ml = (wid - 8) * 64 + lane;                  // lane of the 256-lane mover group
    auto u_is_a = [&](int u) __attribute__((always_inline)) { return u < NLP_A; };
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)g.a.img, 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)g.b.img, 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a2 =
        __builtin_amdgcn_make_buffer_rsrc((void*)((EXT && g.a2.img) ? g.a2.img : g.a.img), 0, 0xFFFFFFFF, 0x00020000);
    const bool two = EXT && g.a2.img != nullptr;
    constexpr bool B2 = BK && AK;                          // (the two-part B operand lives in the k-major x k-major instantiation)
    const bool two_b = B2 && g.b2.img != nullptr;
    const __amdgpu_buffer_rsrc_t rsrc_b2 =
        __builtin_amdgcn_make_buffer_rsrc((void*)((B2 && g.b2.img) ? g.b2.img : g.b.img), 0, 0xFFFFFFFF, 0x00020000);
    bool bc_p2 = false;                                    // wave-uniform: the B cursor's tile reads the second B part
    bool part2 = false;                                    // wave-uniform: the fetch cursor is inside the second A part
    unsigned step_a = (unsigned)g.a.step_bytes;
    const unsigned step_b = (unsigned)g.b.step_bytes;
    unsigned src[NLP];
    int cur_ti = 0;
    // BK: a step's B tile is 32 reduction rows x 768 contiguous bytes (48 pieces: 4 column groups).  One DMA instruction of mover
    // wave w moves 4 rows x 16 consecutive pieces (4 runs of 256 B): instruction ub covers rows 8w + 4 (ub / 3) + 0..3, pieces
    // 16 (ub % 3) + 0..15; lane (rsub, i) = (lane >> 4, lane & 15) takes piece 16 t + (i ^ f) of row rsub, f = 2 rsub | 8 (w & 1):
    // the XOR spreads the pieces one transposed read touches (8 rows x 2 adjacent pieces per 32-lane half) over the 16 sixteen-byte
    // bank slots.  So a lane moves pieces of TWO rows per step; their ids are requested one step ahead, BEFORE that step's DMA
    // pieces, so the wait that covers them leaves the pieces in flight.
    const int bk_rsub = (lane >> 4) & 3, bk_w = wid - 8;
    const int bk_ipc = (lane & 15) ^ ((bk_rsub << 1) | ((bk_w & 1) << 3));
    int64_t bk_id[2] = {0, 0};                             // raw ids (or positions) of the rows the NEXT fetch moves
    bool bk_ok[2] = {false, false};
    unsigned bk_zmask = 0, ak_zmask = 0;                   // pieces whose column group is past the image row
    auto bk_request = [&](int ks, bool p2 = false) __attribute__((always_inline)) {
      const int64_t* const rows = (B2 && p2) ? g.b2.rows : g.b.rows;     // (the rows of the part the requested step's tile reads)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kl = 8 * bk_w + 4 * h + bk_rsub;
        const int64_t pos = g.bk_interleave ? (int64_t)kl * g.bk_interleave + ks : (int64_t)ks * 32 + kl;
        bk_ok[h] = pos < g.bk_red && (!g.bk_interleave || ks < g.bk_interleave);
        bk_id[h] = rows ? rows[bk_ok[h] ? pos : 0] : pos;
      }
    };
    // second part of A: the rows of tile row `ti` in a2 (its own gather / zero row), from a2's first reduction step
    auto make_src_a2 = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NLP_A; ++u) {
        if (!u_is_a(u)) continue;
        const int i = u * 256 + ml;
        const int r = i / 12, jp = i - r * 12;
        const int j = (jp & ~3) | ((jp & 3) ^ swz(r));
        const int64_t gi = (int64_t)ti * BM + r;
        int64_t id = gi;
        if (g.a2.rows) id = g.a2.rows[gi < g.M ? gi : g.M - 1];
        const bool ok = gi < g.M && id >= 0 && id < g.a2.nrows;
        src[u] = (unsigned)((ok ? id : g.a2.zero_row) * g.a2.row_bytes + j * 16);
      }
    };
    auto make_src = [&](const Tile& t, auto doA_, auto doB_) __attribute__((always_inline)) {
      constexpr bool doA = decltype(doA_)::value, doB = decltype(doB_)::value;   // (compile-time: a runtime flag here sends src[] to scratch)
      int64_t rid[NLP_A];
      if (doA) {
#pragma unroll
        for (int u = 0; u < NLP_A; ++u) {
          const int64_t gi = (int64_t)t.ti * BM + (u * 256 + ml) / 12;
          rid[u] = gi;
          if (g.a.rows) rid[u] = g.a.rows[gi < g.M ? gi : g.M - 1];
        }
      }
      if (BK && doB) bk_zmask = 0;
      if (AK && doA) ak_zmask = 0;
#pragma unroll
      for (int u = 0; u < NLP; ++u) {
        if (EA && (u < NLP_A ? !doA : !doB)) continue;         // (EA: this call sets up one operand's part only)
        const int i = u * 256 + ml;
        const int r = i / 12, jp = i - r * 12;
        const int j = (jp & ~3) | ((jp & 3) ^ swz(r));
        int64_t off;
        if (AK && u < NLP_A) {
          const int pc = 16 * (u % 3) + bk_ipc;                // as the B pieces below, over the A image's columns
          const int grp = t.ti * (BM / 32) + pc / 12;
          if (grp >= g.ak_groups) ak_zmask |= 1u << u;
          src[u] = (unsigned)((grp < g.ak_groups ? grp : 0) * X3_GROUP_BYTES + (pc % 12) * 16);
          continue;
        } else if (u_is_a(u)) {
          const int64_t id = rid[u < NLP_A ? u : 0];
          const bool ok = (int64_t)t.ti * BM + r < g.M && id >= 0 && id < g.a.nrows;
          off = (ok ? id : g.a.zero_row) * g.a.row_bytes + (int64_t)t.ks_begin * g.a.step_bytes;
        } else if (BK) {
          const int pc = 16 * ((u - NLP_A) % 3) + bk_ipc;     // piece 0 .. 47 of the row's 768-byte run
          const bool p2 = two_b && t.tj >= g.NJ1;             // (tile-uniform: the second B part's column tiles)
          const int grp = (p2 ? t.tj - g.NJ1 : t.tj) * (BN / 32) + pc / 12;
          const int lim = p2 ? g.bk2_groups : g.bk_groups;
          if (grp >= lim) bk_zmask |= 1u << (u - NLP_A);
          src[u] = (unsigned)((grp < lim ? grp : 0) * X3_GROUP_BYTES + (pc % 12) * 16);
          if (B2) bc_p2 = p2;
          continue;
        } else {
          const int64_t gj = (int64_t)t.tj * BN + (r - BM);
          off = (gj < g.N ? gj : g.b.zero_row) * g.b.row_bytes + (int64_t)t.ks_begin * g.b.step_bytes;
        }
        src[u] = (unsigned)(off + j * 16);
      }
      if (EXT && doA) { part2 = false; step_a = (unsigned)g.a.step_bytes; cur_ti = t.ti; }
    };
    // fetch cursors: (tile, reduction step) of the next A part / the next B part to be issued (EA: A runs one stage ahead of B;
    // otherwise they move together and only `ca` is advanced)
    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    int a_logical = first, a_ks, a_end, b_logical = first, b_ks, b_end;
    { const Tile t = decode(first); make_src(t, yes_t(), yes_t()); a_ks = b_ks = t.ks_begin; a_end = b_end = t.ks_end; }
    if (BK) bk_request(b_ks, bc_p2);
    auto fetch = [&](int stage, auto doA_, auto doB_) __attribute__((always_inline)) {
      constexpr bool doA = decltype(doA_)::value, doB = decltype(doB_)::value;
      const int fks = doA ? a_ks : b_ks;
      unsigned bk_row[2] = {0, 0}, bk_zero = 0;
      const bool b_p2 = B2 && bc_p2;                          // (this step's tile: make_src set it when the cursor reached the tile)
      if (BK && doB) {
        // this step's rows (requested a step ago), then the request for the step after it — ahead of this step's pieces
        const int64_t b_rb = b_p2 ? g.b2.row_bytes : g.b.row_bytes, b_nr = b_p2 ? g.b2.nrows : g.b.nrows;
        bk_zero = (unsigned)((b_p2 ? g.b2.zero_row : g.b.zero_row) * b_rb);
#pragma unroll
        for (int h = 0; h < 2; ++h)
          bk_row[h] = (bk_ok[h] && bk_id[h] >= 0 && bk_id[h] < b_nr) ? (unsigned)(bk_id[h] * b_rb) : bk_zero;
        if (b_ks + 1 < b_end) bk_request(b_ks + 1, b_p2);
        else if (b_logical + nslots < last_logical) {
          const Tile tnx = decode(b_logical + nslots);
          bk_request(tnx.ks_begin, two_b && tnx.tj >= g.NJ1);
        }
      }
      unsigned ak_row[2] = {0, 0}, ak_zero = 0;
      if (AK && doA) {                                     // A rows of this step: reduction position = image row, nothing to fetch
        ak_zero = (unsigned)(g.a.zero_row * g.a.row_bytes);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int64_t pos = (int64_t)fks * 32 + 8 * bk_w + 4 * h + bk_rsub;
          ak_row[h] = pos < g.bk_red ? (unsigned)(pos * g.a.row_bytes) : ak_zero;
        }
      }
      if (EXT && doA && two && !part2 && fks >= g.nsteps1) {      // entering the second A part of this tile
        make_src_a2(cur_ti);
        part2 = true; step_a = (unsigned)g.a2.step_bytes;
      }
      const __amdgpu_buffer_rsrc_t r1 = rsrc_a, r2 = rsrc_a2;   // copies first: a conditional over two captured
      const __amdgpu_buffer_rsrc_t rs_a = (EXT && part2) ? r2 : r1;   // references indexes the closure dynamically and pins it in scratch
      const __amdgpu_buffer_rsrc_t r3 = rsrc_b, r4 = rsrc_b2;
      const __amdgpu_buffer_rsrc_t rs_b = b_p2 ? r4 : r3;
      static_for<0, NLP>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        unsigned so = src[u];
        if (BK && u >= NLP_A) so += ((bk_zmask >> (u - NLP_A)) & 1) ? bk_zero : bk_row[(u - NLP_A) / 3];
        if (AK && u < NLP_A) so += ((ak_zmask >> u) & 1) ? ak_zero : ak_row[u / 3];
        const bool ia = u_is_a(u);
        const bool mine = !EA || (u < NLP_A ? doA : doB);    // (EA: this call issues one operand's part)
        if (mine) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ia ? rs_a : rs_b,
                                                   (lptr_t)(smem + (stage % NSTAGE) * STAGE + (u * 256 + (wid - 8) * 64) * 16), 16, so, 0, 0, 0);
          if (!(BK && u >= NLP_A) && !(AK && u < NLP_A)) src[u] += ia ? step_a : step_b;
        }
      });
      // on to the next tile when a cursor has issued its tile's last step (EA: each operand has its own cursor)
      if constexpr (doA) {
        if (++a_ks == a_end && a_logical + nslots < last_logical) {
          a_logical += nslots;
          const Tile t = decode(a_logical);
          make_src(t, yes_t(), std::integral_constant<bool, doB>());
          a_ks = t.ks_begin; a_end = t.ks_end;
        }
        if constexpr (doB) { b_logical = a_logical; b_ks = a_ks; b_end = a_end; }
      } else {
        if (++b_ks == b_end && b_logical + nslots < last_logical) {
          b_logical += nslots;
          const Tile t = decode(b_logical);
          make_src(t, no_t(), yes_t());
          b_ks = t.ks_begin; b_end = t.ks_end;
        }
      }
    };
    // the movers run NSTAGE - 1 stages ahead of the multipliers: after the barrier that opens step n they issue stage
    // n + NSTAGE - 1 into the buffer step n - 1 released, then wait for stage n + 1 only (with three stages the newest
    // NLP pieces stay in flight across the barrier: a stage has two steps to land)
    if constexpr (EA) {
      // A(n) / B(n): the A / B part of stage n.  Issue order: A(0) B(0) A(1) | step n: B(n + 1) after the barrier that opens it (the B
      // region of the other buffer was last read in step n - 1), A(n + 2) after its SECOND barrier (the multipliers have this
      // step's A fragments in registers).  Before the next step opens, A(n + 1) and B(n + 1) must have landed; A(n + 2) — the
      // youngest NLP_A instructions — may stay in flight.
      // (n = -1 is the prologue: B(0), A(0), A(1).  One call site per operand: each inlined copy of the fetch code costs registers.)
      // A(n) / B(n): the A / B part of stage n.  Step n: B(n + 1) after the barrier that opens it (the B region of the other buffer was
      // last read in step n - 1), A(n + 2) after `mid`.  Before the next step opens, A(n + 1) and B(n + 1) must have landed; A(n + 2) —
      // the youngest NLP_A instructions — may stay in flight.
      // (Also tried: a THIRD barrier before the last column block's MFMAs that returns the B part as well, so that all of stage n + 2
      // is issued during step n — the step got 6 % SLOWER (0.970 -> 1.026 ms per train step) where `mid` alone gains 2.7 %: every
      // barrier puts the eight multiplier waves back in lockstep, and two waves of a SIMD that issue their MFMA groups together wait
      // for each other's matrix pipe.  And NO barrier at all — six LDS counters, per stage buffer `full` / `A taken` / `B read`, added to
      // by ds_add and polled by ds_read_b32 + s_sleep, every wave waiting only for what it needs: parity-green and 22-45 % slower
      // (dW_pool0 0.227 -> 0.330 ms with s_sleep 1, block durations 260 -> 317 us with s_sleep 4): gfx950 has no blocking wait but
      // s_barrier, and four mover waves polling LDS take the cycles the fragment reads need.  And `mid` moved to the HALF of the step
      // (column block 3's fragments requested ahead of it, so that it returns the whole buffer and all of stage n + 2 is issued there,
      // same two barriers per step): 0.955 -> 1.106 ms per train step.)
      int ia = 0, ib = 0;
      X3_PH(unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;)
      for (int n = -1; n < total; ++n) {
        X3_PH_T(m0);
        if (n >= 0) barrier();                             // opens step n
        X3_PH_T(m1);
        if (ib < total) { fetch(ib, no_t(), yes_t()); ++ib; }
        X3_PH_T(m2);
        if (n >= 0) barrier();                             // mid: the A part of step n's buffer is free
        X3_PH_T(m3);
        bool young = false;
        for (int r = (n < 0 ? 0 : 1); r < 2; ++r) {
          young = false;
          if (ia < total) { fetch(ia, yes_t(), no_t()); ++ia; young = true; }
        }
        X3_PH_T(m4);
        if (young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLP_A) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X3_PH_T(m5);
        X3_PH(if (n >= 0) { ph[0] += m1 - m0; ph[1] += m2 - m1; ph[2] += m3 - m2; ph[3] += m4 - m3; ph[4] += m5 - m4; ph[5] += 1; })
      }
      X3_PH(if (g.stamps && wid == 8 && lane == 0) for (int i = 0; i < 6; ++i) g.stamps[1024 + 32 * blockIdx.x + 16 + i] = ph[i];)
    } else {
    int issued = 0;
    for (; issued < NSTAGE - 1 && issued < total; ++issued) fetch(issued, yes_t(), yes_t());
    if (NSTAGE == 3 && issued == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int n = 0; n < total; ++n) {
      barrier();                                           // opens step n: the buffer of step n - 1 is free
      if (issued < total) { fetch(issued, yes_t(), yes_t()); ++issued; }
      // stage n + 1 must have landed before the next barrier; what was issued after it may stay in flight
      if (NSTAGE == 3 && issued >= n + 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }
  } else {
    // ---- multipliers: 4 x 2 waves of 64 x 64 ------------------------------------------------------------------------------
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int l15 = lane & 15, quad = lane >> 4;
    int offp[3];
    {
      const int q = swz(l15);
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) {
        const int j = x3_piece(quad, sp);
        offp[sp] = ((j & ~3) | ((j & 3) ^ q)) * 16;
      }
    }
    const int rowa = (wm * WROWS + l15) * 192, rowb = (BM + wn * WCOLS + l15) * 192;
    // BK: the B stage holds 32 reduction rows x 48 pieces; piece pc of row k sits at piece index
    // (3 ((k >> 2) & 1) + (pc >> 4)) * 256 + (k >> 3) * 64 + (k & 3) * 16 + ((pc & 15) ^ (2 (k & 3) | 8 ((k >> 3) & 1))) of the B part (what
    // the movers' lane-linear instructions produce).  A fragment (16 columns, 8 k per lane) is two ds_read_b64_tr_b16: lane
    // 4q + p of a 16-lane group supplies row 8 quad + 4 h + q, columns 4p .. 4p + 3 of the block = half of piece
    // 12 group + 4 plane + chunk.
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4;
    const int pq = l15 >> 2, pp = l15 & 3;
    const int bk_mask = (pp >> 1) | (pq << 1) | ((quad & 1) << 3);
    const int bk_lane = (quad * 64 + pq * 16) * 16 + 8 * (pp & 1);
    auto k_frag = [&](const unsigned char* st, int n0, int sp) __attribute__((always_inline)) {   // st: the operand's part of the stage
      const int pce = (n0 >> 5) * 12 + sp * 4 + ((n0 >> 3) & 3);     // first piece (even) of the block whose first column is n0
      const int at = bk_lane + ((pce >> 4) * 256 + ((pce & 15) ^ bk_mask)) * 16;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(st + at));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(st + at + 3 * 256 * 16));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto bk_frag = [&](const unsigned char* st, int y, int sp) __attribute__((always_inline)) {
      return k_frag(st + B_BASE, wn * WCOLS + y * 16, sp);
    };
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[RB][CB];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.f;
    };
    // rbv / cbv: how many of the wave tile's 16-row / 16-column blocks hold any real output (the rest is tile padding:
    // M or N not a multiple of the tile).  Padding blocks are not multiplied: the kernel is paced by power, not by the
    // busiest SIMD, so every MFMA not issued comes back as clock (N = 602: 2 of 40 column blocks, 5-10 % of the MFMAs).
    // (EA: `mid` = the step's second barrier, met by every wave of the block — ONE call site, behind the first column block's MFMAs,
    // which have consumed every A fragment of the step: all of them are in registers and the A part of this stage buffer goes back
    // to the movers.  A second call site on the early-return path cost 40-80 spilled registers.)
    X3_PH(unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;)
    // STAGGER (g.stagger; measured with the phase stamps of tools/x3_phase_probe.py): the two multiplier waves of a SIMD (w, w + 4) open a
    // step together — both request their fragments behind the barrier and the matrix pipe idles for an LDS round trip (~370 of a 256 x
    // 128 x 32 step's ~4 500 cycles) before either can issue.  Waves 4-7 therefore keep the LAST column block of every step but a
    // tile's last one for AFTER the next step's opening barrier: its operands (all A fragments, ring slot 1 of B) stay in registers
    // across the barrier, its MFMAs fill the pipe while the partner's fragments arrive, and the wave requests its own A fragments
    // row block by row block as those MFMAs release the registers.  Every accumulator still sees its steps in order: same bits.
    constexpr bool STAG_OK = (CB % 2) == 0;                   // (the deferred block reads ring slot 1; slot 0 takes the new step's first block)
    const bool late = STAG_OK && (g.stagger & 1) && wid >= 4;
    // (A/B, g.stagger & 2: a static issue priority for the second-dispatched half — MI355X_MICROARCH.md, "Two waves per SIMD" item 4)
    if ((g.stagger & 2) && wid >= 4) __builtin_amdgcn_s_setprio(1);
    bf16x8 a[RB][3], b[2][3];
#pragma unroll
    for (int t = 0; t < RB; ++t)
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) a[t][sp] = bf16x8{};
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) { b[0][sp] = bf16x8{}; b[1][sp] = bf16x8{}; }
    auto compute = [&](int buf, int rbv, int cbv, bool pend, bool defer) __attribute__((always_inline)) {
      const unsigned char* st = smem + buf * STAGE;
      constexpr int YL = CB - 1;                              // the column block a late wave defers (ring slot YL & 1 = 1)
      const bool work = rbv > 0 && cbv > 0;
      if (!EA && !work) return;
      {                                                       // (no branch around the loads: a control-flow join in front of the first
#pragma unroll                                                //  MFMA makes the compiler wait for every load in flight — lgkmcnt(0))
        for (int sp = 0; sp < 3; ++sp) {
          if constexpr (BK) b[0][sp] = bk_frag(st, 0, sp);
          else b[0][sp] = *(const bf16x8*)(st + rowb + offp[sp]);
        }
        // (EVERY fragment of the wave tile is requested, padding blocks too — their rows exist in the stage, as zero rows or stale
        // bytes nobody multiplies: with the loads under `t < rbv` / `y + 1 < cbv` branches the compiler cannot count what is in
        // flight, so the first MFMA waited for 16 of the step's 18 fragment loads (`s_waitcnt lgkmcnt(2)`) although it needs the
        // first four — with eight waves in lockstep behind the barrier that is ~150 KB through LDS, a quarter of the step, before
        // any matrix instruction issues)
#pragma unroll
        for (int t = 0; t < RB; ++t) {
          if constexpr (STAG_OK) {
            // the previous step's deferred column block, row block t, on the fragments still in registers — then their reload
            if (pend && work && t < rbv && YL < cbv) {
              acc[t][YL] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][2], a[t][0], acc[t][YL], 0, 0, 0);
              acc[t][YL] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][1], a[t][1], acc[t][YL], 0, 0, 0);
              acc[t][YL] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][0], a[t][2], acc[t][YL], 0, 0, 0);
              acc[t][YL] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][1], a[t][0], acc[t][YL], 0, 0, 0);
              acc[t][YL] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][0], a[t][1], acc[t][YL], 0, 0, 0);
              acc[t][YL] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][0], a[t][0], acc[t][YL], 0, 0, 0);
            }
          }
#pragma unroll
          for (int sp = 0; sp < 3; ++sp) {
            if constexpr (AK) a[t][sp] = k_frag(st, wm * WROWS + t * 16, sp);
            else a[t][sp] = *(const bf16x8*)(st + rowa + t * 16 * 192 + offp[sp]);
          }
        }
      }
      static_for<0, CB>([&](auto yc) __attribute__((always_inline)) {
        constexpr int y = decltype(yc)::value;
        if (work && y < cbv) {
          if constexpr (y + 1 < CB) {
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
              if constexpr (BK) b[(y + 1) & 1][sp] = bk_frag(st, y + 1, sp);
              else b[(y + 1) & 1][sp] = *(const bf16x8*)(st + rowb + (y + 1) * 16 * 192 + offp[sp]);
            }
          }
          if (!(STAG_OK && y == YL && defer)) {
#pragma unroll
            for (int x = 0; x < RB; ++x)
              if (x < rbv) {
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][2], a[x][0], acc[x][y], 0, 0, 0);
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][1], a[x][1], acc[x][y], 0, 0, 0);
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][0], a[x][2], acc[x][y], 0, 0, 0);
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][1], a[x][0], acc[x][y], 0, 0, 0);
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][0], a[x][1], acc[x][y], 0, 0, 0);
                acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[y & 1][0], a[x][0], acc[x][y], 0, 0, 0);
              }
          }
        }
        if constexpr (EA && y == 0) {
          X3_PH_T(p2);
          barrier();                                          // mid
          X3_PH_T(p3);
        }
      });
    };
    auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
      // lane coordinates re-read here through an opaque copy: address arithmetic hoisted out of the tile loop would live in
      // (and spill from) registers the main loop needs
      int l15 = lane & 15, quad = lane >> 4;
      asm volatile("" : "+v"(l15), "+v"(quad));
      float* const dst = g.nsplit > 1 ? g.ws + (int64_t)t.split * g.M * g.ws_ld : g.C;
      const int64_t ldd = g.nsplit > 1 ? g.ws_ld : g.ldc;
      const bool vec_ok = (ldd & 3) == 0 && ((uintptr_t)dst & 15) == 0;
      const bool fin = g.nsplit == 1;
      const bool add_vec = EXT && g.add && (g.ld_add & 3) == 0 && ((uintptr_t)g.add & 15) == 0;
      if constexpr (EXT) {
        // The per-row addend (the self term S0[dst] of an inference layer) FIRST, every 16-byte load of the wave tile issued
        // before anything is stored: written into the store loop below, each load sat behind the previous group's stores (the
        // output and the table may alias for all the compiler knows) and a tile paid RB x CB memory latencies in a row.  Loads
        // are unconditional — an invalid group reads the table's first row and is masked afterwards — so nothing between them
        // needs a wait (the same rule as the aggregator's row loads).
        if (add_vec) {
          const float* ap[RB];
          bool aok[RB];
#pragma unroll
          for (int x = 0; x < RB; ++x) {
            const int64_t row = (int64_t)t.ti * BM + wm * WROWS + x * 16 + l15;
            const bool rok = row < g.M;
            const int64_t ar = g.add_rows ? g.add_rows[rok ? row : 0] : row;
            aok[x] = rok && ar >= 0 && ar < g.add_nrows;
            ap[x] = g.add + (aok[x] ? ar : 0) * g.ld_add;
          }
          float4 tv[RB][CB];
#pragma unroll
          for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int y = 0; y < CB; ++y) {
              const int64_t col = (int64_t)t.tj * BN + wn * WCOLS + y * 16 + 4 * quad;
              tv[x][y] = *(const float4*)(ap[x] + (col + 4 <= g.N ? col : 0));
            }
#pragma unroll
          for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int y = 0; y < CB; ++y) {
              const int64_t col = (int64_t)t.tj * BN + wn * WCOLS + y * 16 + 4 * quad;
              if (aok[x] && col + 4 <= g.N) {
                acc[x][y][0] += tv[x][y].x; acc[x][y][1] += tv[x][y].y; acc[x][y][2] += tv[x][y].z; acc[x][y][3] += tv[x][y].w;
              }
            }
        }
        // ... then the ReLU mask (the forward output of the layer whose input gradient this is), the same way: all loads, then
        // the selects — dX = (dY . W + head) (.) [y > 0] leaves the kernel already masked, with its image
        if (g.mask && (g.ld_mask & 3) == 0 && ((uintptr_t)g.mask & 15) == 0 && (g.N & 3) == 0) {
          float4 mv[RB][CB];
#pragma unroll
          for (int x = 0; x < RB; ++x) {
            const int64_t row = (int64_t)t.ti * BM + wm * WROWS + x * 16 + l15;
            const float* mp = g.mask + (row < g.M ? row : 0) * g.ld_mask;
#pragma unroll
            for (int y = 0; y < CB; ++y) {
              const int64_t col = (int64_t)t.tj * BN + wn * WCOLS + y * 16 + 4 * quad;
              mv[x][y] = *(const float4*)(mp + (col + 4 <= g.N ? col : 0));
            }
          }
#pragma unroll
          for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int y = 0; y < CB; ++y) {
              acc[x][y][0] = mv[x][y].x > 0.f ? acc[x][y][0] : 0.f; acc[x][y][1] = mv[x][y].y > 0.f ? acc[x][y][1] : 0.f;
              acc[x][y][2] = mv[x][y].z > 0.f ? acc[x][y][2] : 0.f; acc[x][y][3] = mv[x][y].w > 0.f ? acc[x][y][3] : 0.f;
            }
        }
      }
#pragma unroll
      for (int x = 0; x < RB; ++x)
#pragma unroll
        for (int y = 0; y < CB; ++y) {
          const int64_t row = (int64_t)t.ti * BM + wm * WROWS + x * 16 + l15, col = (int64_t)t.tj * BN + wn * WCOLS + y * 16 + 4 * quad;
          float v[4] = {acc[x][y][0], acc[x][y][1], acc[x][y][2], acc[x][y][3]};
          const bool rok = row < g.M;
          const bool has_oc = fin && g.ones_col && col + 3 >= g.N - 1 && col < g.N;
          if (EXT && g.add && rok && col < g.N && !(add_vec && col + 4 <= g.N)) {   // (partial column groups, unaligned tables)
            const int64_t ar = g.add_rows ? g.add_rows[row] : row;
            if (ar >= 0 && ar < g.add_nrows) {
              const float* ap = g.add + ar * g.ld_add + col;
#pragma unroll
              for (int c = 0; c < 4; ++c) if (col + c < g.N) v[c] += ap[c];
            }
          }
          if (fin && g.relu) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
          }
          if (EXT && g.out_img && rok && col * 2 < g.out_row_bytes / 3) {
            // the output's bf16x3 image beside the fp32 store: this thread's 4 columns are half a 16-byte piece of each plane;
            // columns past N are the image's padding (zero; 1.0 at column N when out_append_ones), row M is the zero row
            float e[4], z[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const bool one = g.out_append_ones && col + c == g.N;
              e[c] = col + c < g.N ? v[c] : (one ? 1.f : 0.f);
              z[c] = one ? 1.f : 0.f;
            }
            const int ch = (int)(col >> 2);
            const int64_t off = (int64_t)(ch >> 3) * X3_GROUP_BYTES + (ch & 1) * 8;
            unsigned h0, m0, l0, h1, m1, l1;
            split3(e[0], e[1], h0, m0, l0); split3(e[2], e[3], h1, m1, l1);
            unsigned char* rp = g.out_img + row * g.out_row_bytes + off;
            *(uint2*)(rp + x3_piece((ch & 7) >> 1, 0) * 16) = make_uint2(h0, h1);
            *(uint2*)(rp + x3_piece((ch & 7) >> 1, 1) * 16) = make_uint2(m0, m1);
            *(uint2*)(rp + x3_piece((ch & 7) >> 1, 2) * 16) = make_uint2(l0, l1);
            if (row == g.M - 1) {
              split3(z[0], z[1], h0, m0, l0); split3(z[2], z[3], h1, m1, l1);
              rp += g.out_row_bytes;
              *(uint2*)(rp + x3_piece((ch & 7) >> 1, 0) * 16) = make_uint2(h0, h1);
              *(uint2*)(rp + x3_piece((ch & 7) >> 1, 1) * 16) = make_uint2(m0, m1);
              *(uint2*)(rp + x3_piece((ch & 7) >> 1, 2) * 16) = make_uint2(l0, l1);
            }
          }
          if (EXT && g.y_keep && rok && !g.y_keep[row]) continue;   // (a row only the next layer's image product reads)
          const bool vec = rok && vec_ok && !has_oc && col < g.N && col + 4 <= ldd;
          if (vec) *(float4*)(dst + row * ldd + col) = make_float4(v[0], v[1], v[2], v[3]);
          else if (rok && col < g.N) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (col + c >= g.N) continue;
              if (fin && g.ones_col && col + c == g.N - 1) { if (g.db) g.db[row] = v[c]; if (g.db2) g.db2[row] = v[c]; }
              else dst[row * ldd + col + c] = v[c];
            }
          }
        }
    };
    zero_acc();
    int n = 0;
    for (int logical = first; logical < last_logical; logical += nslots) {
      const Tile tc = decode(logical);
      const int64_t rleft = g.M - ((int64_t)tc.ti * BM + wm * WROWS), cleft = g.N - ((int64_t)tc.tj * BN + wn * WCOLS);
      const int rbv = rleft >= RB * 16 ? RB : (int)((rleft + 15) >> 4), cbv = cleft >= CB * 16 ? CB : (int)((cleft + 15) >> 4);
      bool pend = false;
      for (int ks = tc.ks_begin; ks < tc.ks_end; ++ks, ++n) {
        X3_PH_T(p0);
        X3_PH(if (n > 0) ph[3] += p0 - p4;)                // (the step's tail: behind `mid` up to the arrival at the next opening barrier)
        // (a deferred block's fragments were requested a column block ago; they must have LEFT the stage before the barrier hands its
        // B part to the movers)
        if (late) __builtin_amdgcn_s_waitcnt(0xC07F);
        barrier();                                         // stage n has landed (the movers waited for it)
        X3_PH_T(p1);
        const bool defer = late && ks + 1 < tc.ks_end;     // (a tile's last step runs whole: the epilogue follows)
        compute(n % NSTAGE, rbv, cbv, pend, defer);
        pend = defer;
        X3_PH_T(p4);
        X3_PH(ph[0] += p1 - p0; ph[1] += p2 - p1; ph[2] += p3 - p2; ph[4] += p4 - p3; ph[5] += 1;)
      }
      // (nothing is deferred across a tile's end: the fragment registers are dead through the epilogue)
#pragma unroll
      for (int t = 0; t < RB; ++t)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) a[t][sp] = bf16x8{};
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) { b[0][sp] = bf16x8{}; b[1][sp] = bf16x8{}; }
      epilogue(tc);
      zero_acc();
    }
    X3_PH(if (g.stamps && (wid == 0 || wid == 4) && lane == 0) for (int i = 0; i < 6; ++i) g.stamps[1024 + 32 * blockIdx.x + 2 * wid + i] = ph[i];)
  }
  (void)NSTORE;
  if (g.stamps && tid == 0) {
    g.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    g.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
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
    if (oc) { if (g.db) g.db[row] = v; if (g.db2) g.db2[row] = v; }
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
                                                  int64_t row_bytes, int64_t img_row_bytes) {
  // row_bytes = this matrix's own groups x 192; img_row_bytes = the distance between image rows (larger when the matrix is one
  // part of a K-concatenated image: img then points at the part's first group)
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
    unsigned char* d = img + r * img_row_bytes + (int64_t)(ch >> 2) * X3_GROUP_BYTES;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) *(uint4*)(d + x3_piece(ch & 3, sp) * 16) = o[sp];
  }
}

// image of the TRANSPOSE: image row n, reduction index m, stored GROUP-MAJOR (the rows of one 32-deep reduction step are
// contiguous: one tile of this kernel writes two 12 KB runs, and a GEMM stage reads one run):
// image[m/32][n][plane][m%32] = split(src[row(m), n]), 64 x 64 tiles through LDS.  ones_row: image row N is 1.0 for m < M (bias gradient operand).  Pad m >= M is zero.
template <bool ROWS, bool WIDE>
__global__ void __launch_bounds__(256) k_x3_split_t(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ rows,
                                                    int64_t nrows_src, int64_t M, int N, int ones_row, int64_t G_il, int64_t Mi,
                                                    unsigned char* __restrict__ img, int64_t row_bytes) {
  __shared__ float tile[64][65];
  const int nt = blockIdx.y;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int n0 = nt * 64;
  const int tid = threadIdx.x, ty = tid >> 4, tx = (tid & 15) * 4;
  const int G = (int)(row_bytes / X3_GROUP_BYTES);
  const int64_t zero_row = (int64_t)N + (ones_row ? 1 : 0);
  const int64_t gstride = (zero_row + 1) * X3_GROUP_BYTES;       // GROUP-MAJOR image: [group][image row][192 B]
  // The 4 row ids first, then the 4 row reads, nothing between the loads that waits: a missing row (padding, id out
  // of range) reads row 0 and is zeroed afterwards; the last float4 of a row whose length is not a multiple of 4 is
  // read 16 bytes back from the row's end and shifted.  (With a branch per row each of the 8 loads waited for the one
  // before; measured gain of issuing them together: ~4 % of this kernel — it moves 380 MB in ~72 us for the
  // 62 750 x 602 image, i.e. it was already near what HBM gives a gather + write mix.)
  const bool have = !ROWS || nrows_src > 0;                       // any readable row at all (block-uniform)
  const int col0 = n0 + tx;
  constexpr bool wide = WIDE;                                     // N >= 4
  const int cl = wide ? min(col0, N - 4) : 0, sh = col0 - cl;     // sh = 0: a whole float4; 1..3: a row tail; >= 4: past the row
  bool okk[4];
  int64_t rk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t ip = m0 + ty + 16 * k;                          // position in the image's reduction index
    const int64_t i = G_il ? (ip & 31) * G_il + (ip >> 5) : ip;   // the source row it stands for (round-robin dealing)
    okk[k] = have && ip < Mi && i < M;
    rk[k] = ROWS ? rows[okk[k] ? i : 0] : i;
  }
  float4 w[4];
  const float* pk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    okk[k] = okk[k] && (!ROWS || (rk[k] >= 0 && rk[k] < nrows_src));
    pk[k] = src + (okk[k] ? rk[k] : 0) * ld;
    w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (have) {
    if (wide) {
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = ld16(pk[k] + cl);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (col0 < N) w[k].x = pk[k][col0];
        if (col0 + 1 < N) w[k].y = pk[k][col0 + 1];
        if (col0 + 2 < N) w[k].z = pk[k][col0 + 2];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 v = w[k];
    if (wide && sh) {                                             // elements col0 .. col0 + 3 of which only those < N exist
      const float4 t = v;
      v.x = sh == 1 ? t.y : sh == 2 ? t.z : sh == 3 ? t.w : 0.f;
      v.y = sh == 1 ? t.z : sh == 2 ? t.w : 0.f;
      v.z = sh == 1 ? t.w : 0.f;
      v.w = 0.f;
    }
    if (!okk[k]) v = make_float4(0.f, 0.f, 0.f, 0.f);
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
  if (nt == 0 && tid >= 64 && tid < 64 + 24) {            // the zero row (last row of every group slab)
    const int t = tid - 64, gl = t / 12;
    const int64_t grp = m0 / 32 + gl;
    if (grp < G) *(uint4*)(img + grp * gstride + zero_row * X3_GROUP_BYTES + (t - gl * 12) * 16) = make_uint4(0, 0, 0, 0);
  }
  if (ones_row && nt == 0 && tid < 8) {   // 8 chunks of 8 m: row N of the image
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
                     ld, rows, nrows_src, R, K, append, append_vec, (unsigned char*)image, row_bytes, row_bytes);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// One PART of a K-concatenated image (the B operand of a two-part product, ogl_linear_fwd_x3_ext): the image rows are
// image_row_bytes apart (= 192 x the groups of ALL parts) and this part starts at group `group_offset` of every row.
extern "C" int ogl_x3_split_into(const float* src, int64_t ld, int64_t R, int K, int append, const float* append_vec, void* image,
                                 int64_t image_row_bytes, int64_t group_offset, ogl_stream_t stream) {
  if (R < 0 || K < 0 || ld < K || append < 0 || append > 2 || (append == 2 && R > 0 && !append_vec) || group_offset < 0) return OGL_EINVAL;
  const int Ki = K + (append ? 1 : 0);
  if (Ki == 0) return OGL_OK;
  const int64_t row_bytes = ogl_cdiv(Ki, 32) * X3_GROUP_BYTES;
  if (!image || (R > 0 && K > 0 && !src) || ((uintptr_t)image & 15) || image_row_bytes < group_offset * X3_GROUP_BYTES + row_bytes ||
      image_row_bytes % X3_GROUP_BYTES)
    return OGL_EINVAL;
  const int64_t total = (R + 1) * (row_bytes / X3_GROUP_BYTES) * 4;
  hipLaunchKernelGGL(k_x3_split, dim3((unsigned)min((int64_t)65536, ogl_cdiv(total, 256))), dim3(256), 0, (hipStream_t)stream, src,
                     ld, nullptr, R, R, K, append, append_vec, (unsigned char*)image + group_offset * X3_GROUP_BYTES, row_bytes,
                     image_row_bytes);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// Several small images in ONE launch: the weight images of a train step (every one a ~600 x 600 matrix: as separate launches
// each costs a launch-bound 6-7 us, and the transposed ones a transpose launch before that).  blockIdx.y = the part.
// A transposed part reads the matrix column-wise (image row r = column r of src): its threads walk r fastest, so the loads
// stay contiguous.  The appended slot holds vec1[r] + vec2[r] (either may be null: 0) — the summed bias of a dual projection.
#define OGL_X3_SPLIT_MAX_PARTS 8
struct X3SplitBatch {
  const float* src[OGL_X3_SPLIT_MAX_PARTS];
  int64_t ld[OGL_X3_SPLIT_MAX_PARTS];
  int64_t R[OGL_X3_SPLIT_MAX_PARTS];
  int K[OGL_X3_SPLIT_MAX_PARTS];
  int flags[OGL_X3_SPLIT_MAX_PARTS];               // bit 0: transposed, bit 1: appended slot
  const float* vec1[OGL_X3_SPLIT_MAX_PARTS];
  const float* vec2[OGL_X3_SPLIT_MAX_PARTS];
  unsigned char* img[OGL_X3_SPLIT_MAX_PARTS];      // at the part's first group
  int64_t img_row_bytes[OGL_X3_SPLIT_MAX_PARTS];
  // optional passenger (ogl_x3_split_multi with step_dev): the optimiser's per-step scalars — ++*adam_step and {lr / (1 - beta1^t),
  // 1 / sqrt(1 - beta2^t)} in double arithmetic, what k_adam_prepare (loss_optim.hip) does as a launch of its own at the END of the
  // step, computed by one thread of this launch at its START
  int64_t* adam_step; float* adam_scal; double adam_lr, adam_b1, adam_b2;
};

__global__ void __launch_bounds__(256) k_x3_split_multi(X3SplitBatch b) {
  if (b.adam_step && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const int64_t t = *b.adam_step + 1;
    *b.adam_step = t;
    b.adam_scal[0] = (float)(b.adam_lr / (1.0 - pow(b.adam_b1, (double)t)));
    b.adam_scal[1] = (float)(1.0 / sqrt(1.0 - pow(b.adam_b2, (double)t)));
  }
  const int part = blockIdx.y;
  const float* __restrict__ src = b.src[part];
  const int64_t ld = b.ld[part], R = b.R[part], img_row_bytes = b.img_row_bytes[part];
  const int K = b.K[part], flags = b.flags[part];
  const float* __restrict__ v1 = b.vec1[part];
  const float* __restrict__ v2 = b.vec2[part];
  unsigned char* __restrict__ img = b.img[part];
  if (flags & 4) {                                    // a vector-sum part: fp32 out[r] = vec1[r] + vec2[r] (the summed bias of a
    float* out = (float*)img;                         // dual projection whose product runs on fp32 operands)
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < R; t += (int64_t)gridDim.x * blockDim.x)
      out[t] = (v1 ? v1[t] : 0.f) + (v2 ? v2[t] : 0.f);
    return;
  }
  const bool tr = flags & 1, slot = flags & 2;
  const int cpr = ((K + (slot ? 1 : 0) + 31) / 32) * 4;       // 8-element chunks per image row
  const int64_t total = (R + 1) * cpr;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r; int ch;
    if (tr) { ch = (int)(t / (R + 1)); r = t - (int64_t)ch * (R + 1); }
    else { r = t / cpr; ch = (int)(t - r * cpr); }
    const int k = ch * 8;
    float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < R && k < K) {
      if (tr) {
#pragma unroll
        for (int q = 0; q < 8; ++q) if (k + q < K) e[q] = src[(int64_t)(k + q) * ld + r];
      } else {
        const float* p = src + r * ld + k;
        if (k + 8 <= K) {
          const float4 lo = ld16(p), hi = ld16(p + 4);
          e[0] = lo.x; e[1] = lo.y; e[2] = lo.z; e[3] = lo.w; e[4] = hi.x; e[5] = hi.y; e[6] = hi.z; e[7] = hi.w;
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) if (k + q < K) e[q] = p[q];
        }
      }
    }
    if (slot && K >= k && K < k + 8) {
      float av = 0.f;
      if (r < R) { if (v1) av = v1[r]; if (v2) av += v2[r]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) if (k + q == K) e[q] = av;
    }
    uint4 o[3];
    split3(e[0], e[1], o[0].x, o[1].x, o[2].x);
    split3(e[2], e[3], o[0].y, o[1].y, o[2].y);
    split3(e[4], e[5], o[0].z, o[1].z, o[2].z);
    split3(e[6], e[7], o[0].w, o[1].w, o[2].w);
    unsigned char* d = img + r * img_row_bytes + (int64_t)(ch >> 2) * X3_GROUP_BYTES;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) *(uint4*)(d + x3_piece(ch & 3, sp) * 16) = o[sp];
  }
}

static int x3_split_multi(const ogl_x3_split_part* parts, int n_parts, int64_t* adam_step, float* adam_scal, double lr, double beta1,
                          double beta2, ogl_stream_t stream) {
  if (n_parts < 0 || n_parts > OGL_X3_SPLIT_MAX_PARTS || (n_parts > 0 && !parts)) return OGL_EINVAL;
  if (n_parts == 0) return adam_step ? OGL_EINVAL : OGL_OK;       // (the passenger needs a launch to ride in)
  X3SplitBatch b;
  b.adam_step = adam_step; b.adam_scal = adam_scal; b.adam_lr = lr; b.adam_b1 = beta1; b.adam_b2 = beta2;
  int64_t most = 0;
  for (int i = 0; i < n_parts; ++i) {
    const ogl_x3_split_part& q = parts[i];
    if (q.transpose == 2) {                                // vector sum: image = float[R] <- vec1 + vec2
      if (q.R < 0 || !q.image || (!q.vec1 && !q.vec2)) return OGL_EINVAL;
      b.src[i] = nullptr; b.ld[i] = 0; b.R[i] = q.R; b.K[i] = 0; b.flags[i] = 4;
      b.vec1[i] = q.vec1; b.vec2[i] = q.vec2; b.img[i] = (unsigned char*)q.image; b.img_row_bytes[i] = 0;
      most = max(most, q.R);
      continue;
    }
    const int Ki = q.K + (q.append ? 1 : 0);
    if (q.R < 0 || q.K < 0 || Ki == 0 || q.group_offset < 0 || !q.image || ((uintptr_t)q.image & 15)) return OGL_EINVAL;
    if (q.ld < (q.transpose ? q.R : (int64_t)q.K) || (q.R > 0 && q.K > 0 && !q.src)) return OGL_EINVAL;
    if (!q.append && (q.vec1 || q.vec2)) return OGL_EINVAL;
    const int64_t row_bytes = ogl_cdiv(Ki, 32) * X3_GROUP_BYTES;
    if (q.image_row_bytes < q.group_offset * X3_GROUP_BYTES + row_bytes || q.image_row_bytes % X3_GROUP_BYTES) return OGL_EINVAL;
    b.src[i] = q.src; b.ld[i] = q.ld; b.R[i] = q.R; b.K[i] = q.K;
    b.flags[i] = (q.transpose ? 1 : 0) | (q.append ? 2 : 0);
    b.vec1[i] = q.vec1; b.vec2[i] = q.vec2;
    b.img[i] = (unsigned char*)q.image + q.group_offset * X3_GROUP_BYTES;
    b.img_row_bytes[i] = q.image_row_bytes;
    most = max(most, (q.R + 1) * (row_bytes / X3_GROUP_BYTES) * 4);
  }
  for (int i = n_parts; i < OGL_X3_SPLIT_MAX_PARTS; ++i) {
    b.src[i] = nullptr; b.ld[i] = 0; b.R[i] = -1; b.K[i] = 0; b.flags[i] = 0; b.vec1[i] = b.vec2[i] = nullptr; b.img[i] = nullptr;
    b.img_row_bytes[i] = 0;
  }
  hipLaunchKernelGGL(k_x3_split_multi, dim3((unsigned)min((int64_t)4096, ogl_cdiv(most, 256)), (unsigned)n_parts), dim3(256), 0,
                     (hipStream_t)stream, b);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// step_dev / scalars_dev (both or neither): the optimiser's per-step scalars ride along (k_adam_prepare moved from the END of the step —
// where its 5 us and the gap in front of it are on the critical path — into the launch that STARTS the step): ++*step_dev;
// scalars_dev[0] = lr / (1 - beta1^t), scalars_dev[1] = 1 / sqrt(1 - beta2^t).  The optimiser launch of the same step then runs with
// prepare = 0 (ogl_adam_step_multi_slabs).
extern "C" int ogl_x3_split_multi(const ogl_x3_split_part* parts, int n_parts, int64_t* step_dev, float* scalars_dev, double lr,
                                  double beta1, double beta2, ogl_stream_t stream) {
  if ((step_dev == nullptr) != (scalars_dev == nullptr)) return OGL_EINVAL;
  return x3_split_multi(parts, n_parts, step_dev, scalars_dev, lr, beta1, beta2, stream);
}

// dy (.) [y > 0] (ogl_relu_bwd) that ALSO writes the bf16x3 image of its result: the masked gradient is the A operand of the
// input-gradient product that follows (dX = dY . W on the image kernel), so its image costs no pass of its own.
__global__ void __launch_bounds__(256) k_relu_bwd_img(const float* __restrict__ dy, int64_t ldy, const float* __restrict__ y,
                                                      int64_t ldyy, int64_t M, int N, float* __restrict__ out, int64_t ldo,
                                                      unsigned char* __restrict__ img, int64_t row_bytes) {
  const int cpr = (int)(row_bytes / X3_GROUP_BYTES) * 4;   // 8-element chunks per image row
  const int64_t total = (M + 1) * cpr;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / cpr;
    const int ch = (int)(t - r * cpr);
    const int k = ch * 8;
    float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < M && k < N) {
      const float* pd = dy + r * ldy + k;
      const float* py = y + r * ldyy + k;
      float* po = out + r * ldo + k;
      if (k + 8 <= N) {
        const float4 d0 = ld16(pd), d1 = ld16(pd + 4), y0 = ld16(py), y1 = ld16(py + 4);
        e[0] = y0.x > 0.f ? d0.x : 0.f; e[1] = y0.y > 0.f ? d0.y : 0.f; e[2] = y0.z > 0.f ? d0.z : 0.f; e[3] = y0.w > 0.f ? d0.w : 0.f;
        e[4] = y1.x > 0.f ? d1.x : 0.f; e[5] = y1.y > 0.f ? d1.y : 0.f; e[6] = y1.z > 0.f ? d1.z : 0.f; e[7] = y1.w > 0.f ? d1.w : 0.f;
        if ((ldo & 3) == 0 && ((uintptr_t)out & 15) == 0) {
          *(float4*)po = make_float4(e[0], e[1], e[2], e[3]);
          *(float4*)(po + 4) = make_float4(e[4], e[5], e[6], e[7]);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) po[q] = e[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (k + q < N) { e[q] = py[q] > 0.f ? pd[q] : 0.f; po[q] = e[q]; }
      }
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

extern "C" int ogl_relu_bwd_img(const float* dy, int64_t ldy, const float* y, int64_t ldyy, int64_t M, int N, float* out,
                                int64_t ldo, void* image, ogl_stream_t stream) {
  if (M < 0 || N <= 0 || ldy < N || ldyy < N || ldo < N) return OGL_EINVAL;
  if (!image || ((uintptr_t)image & 15) || (M > 0 && (!dy || !y || !out))) return OGL_EINVAL;
  const int64_t row_bytes = ogl_cdiv(N, 32) * X3_GROUP_BYTES;
  const int64_t total = (M + 1) * (row_bytes / X3_GROUP_BYTES) * 4;
  hipLaunchKernelGGL(k_relu_bwd_img, dim3((unsigned)min((int64_t)65536, ogl_cdiv(total, 256))), dim3(256), 0, (hipStream_t)stream,
                     dy, ldy, y, ldyy, M, N, out, ldo, (unsigned char*)image, row_bytes);
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
#define OGL_SPLIT_T(R, W)                                                                                                       \
  hipLaunchKernelGGL((k_x3_split_t<R, W>), grid, dim3(256), 0, (hipStream_t)stream, src, ld, rows, nrows_src, M, N, ones_row,  \
                     interleave, Mi, (unsigned char*)image, row_bytes)
  if (rows) { if (N >= 4) OGL_SPLIT_T(true, true); else OGL_SPLIT_T(true, false); }
  else { if (N >= 4) OGL_SPLIT_T(false, true); else OGL_SPLIT_T(false, false); }
#undef OGL_SPLIT_T
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// 0: 256 x 128 (4 x 2 waves of 64 x 64)   1: 128 x 128 (2 x 4 waves of 64 x 32)   2 (launch_x3 only, one-round products): 192 x 128 (2 x 4 waves of 96 x 32)
static int x3_config(int64_t M, int64_t N) {
  const int64_t w0 = ogl_cdiv(M, 256) * 256 * ogl_cdiv(N, 128) * 128;
  const int64_t w1 = ogl_cdiv(M, 128) * 128 * ogl_cdiv(N, 128) * 128;
  return w1 * 10 < w0 * 9 ? 1 : 0;   // the wider wave tile unless it pads > 10 % more MFMA work
}

// Diagnostics: when set, every k_gemm_x3 launch writes per block {s_memtime, s_memrealtime} at entry and at exit (4 x u64
// per block, <= 256 blocks) into `buf` — the in-kernel shader clock is d(memtime) / d(memrealtime) x 100 MHz.
static unsigned long long* g_x3_stamps = nullptr;
extern "C" int ogl_x3_debug_stamps(void* buf, int reserved) {
  (void)reserved;
  g_x3_stamps = (unsigned long long*)buf;
  return OGL_OK;
}

static int g_x3_tile = -1;
int oglx_knob_x3_tile(int cfg, int* prev) {                      // (ogl_debug_set, csrc/graph.hip)
  if (cfg < -1 || cfg > 2) return OGL_EINVAL;
  *prev = g_x3_tile;
  g_x3_tile = cfg;
  return OGL_OK;
}

// Which instantiation the LAST launch_x3 call ran (template arguments as written at the launch site; trailing defaults omitted):
// bench.py compares it with the kernel name of the committed PMC pass before quoting that pass's traffic beside a launch it timed.
static int g_x3_stagger = -1;            // -1: OGL_X3_STAGGER; 0 / 1: pinned (tests, A/B runs)
int oglx_knob_x3_stagger(int on, int* prev) {
  if (on < -1 || on > 3) return OGL_EINVAL;
  *prev = g_x3_stagger;
  g_x3_stagger = on;
  return OGL_OK;
}
static const char* g_x3_last_kernel = "";
extern "C" const char* ogl_x3_last_kernel(void) { return g_x3_last_kernel; }
#define X3P_LAUNCH(...)                                                                           \
  do {                                                                                            \
    g_x3_last_kernel = "k_gemm_x3p<" #__VA_ARGS__ ">";                                            \
    hipLaunchKernelGGL((k_gemm_x3p<__VA_ARGS__>), grid, block, 0, stream, g);                     \
  } while (0)

static int launch_x3(X3Args& g, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0) return OGL_OK;
  g.stamps = g_x3_stamps;
  // (row-major image: rows x row_bytes; group-major image: groups x step_bytes)
  const int64_t a_bytes = std::max((g.a.zero_row + 1) * g.a.row_bytes, (int64_t)g.nsteps * g.a.step_bytes);
  if (g.ak_groups > 0 && g.bk_groups <= 0) return OGL_EINVAL;
  const int64_t b_bytes = std::max((g.b.zero_row + 1) * g.b.row_bytes, (int64_t)g.nsteps * g.b.step_bytes);
  if (g.a2.img && (g.a2.zero_row + 1) * g.a2.row_bytes >= (1ll << 32)) return OGL_EINVAL;
  if (g.b2.img && ((g.b2.zero_row + 1) * g.b2.row_bytes >= (1ll << 32) || g.ak_groups <= 0 || g.NJ1 <= 0)) return OGL_EINVAL;
  int cfg = x3_config(g.M, g.N);                          // 0: 256 x 128, 1: 128 x 128, 2: 192 x 128
  const bool bk = g.bk_groups > 0;                        // row-major B over the reduction: 128 x 128 (three stages) or 256 x 128 (two)
  if (bk) cfg = g.force_cfg0 ? 0 : 1;
  // producer / consumer kernels (k_gemm_x3p) whenever both images fit 32-bit offsets; OGL_X3_PC=0 forces the self-fetching kernel
  // (the form images of 4 GB and more take) for the plain products: tests/test_gpu_x3.py::test_self_fetching_kernels_parity
  static const char* pc_env = getenv("OGL_X3_PC");
  const bool ext = g.a2.img || g.add || g.out_img || g.mask || g.y_keep;
  const bool pc = (bk || ext || !(pc_env && pc_env[0] == '0')) && a_bytes < (1ll << 32) && b_bytes < (1ll << 32);
  if ((bk || ext) && !pc) return OGL_EINVAL;              // (k-major operands and the extensions live in the producer / consumer kernel)
  if (bk && ext) return OGL_EINVAL;
  if (pc) {
    // a product that is one round of tiles either way takes the smallest tile that still is one round (one block per
    // CU): its critical path is one tile.  [7 199, 602] -> 600: 145 tiles of 256 x 128, 285 of 128 x 128, 190 of 192 x 128
    // (measured 45 us on 256 x 128, 37 us on 192 x 128).
    // (160-row and 160-column tiles were built in round 3 — the fastest tiles of their shapes alone, 1.5-3 % slower inside the
    // replayed train step — and left the library in round 6: DESIGN.md section 8.)
    if (!bk && g.nsplit == 1 && ogl_cdiv(g.M, 256) * ogl_cdiv(g.N, 128) <= 256) {
      // (a step's duration follows its DMA pieces — 12 per tile row and column: 3 072 / 3 840 / 4 608 for 128 / 192 / 256 rows x 128
      // columns — so the shortest one-round tile wins)
      if (ogl_cdiv(g.M, 128) * ogl_cdiv(g.N, 128) <= 256) cfg = 1;
      else if (ogl_cdiv(g.M, 192) * ogl_cdiv(g.N, 128) <= 256) cfg = 2;
    } else if (!bk && g.nsplit == 1 && ogl_cdiv(g.M, 256) * ogl_cdiv(g.N, 128) <= 1024) {
      // a few rounds of tiles: the launch lasts (rounds of 256 tiles) x (tile height) / (the tile's efficiency in steady state:
      // 1 : 0.93 : 0.86 for 256 : 192 : 128 rows) — e.g. [15 500, 600] is 2 rounds of 256 rows, 2 of 192 or 3 of 128
      const int64_t nj = ogl_cdiv(g.N, 128);
      const double c0 = (double)ogl_cdiv(ogl_cdiv(g.M, 256) * nj, 256) * 256.0;
      const double c2 = (double)ogl_cdiv(ogl_cdiv(g.M, 192) * nj, 256) * 192.0 / 0.93;
      const double c1 = (double)ogl_cdiv(ogl_cdiv(g.M, 128) * nj, 256) * 128.0 / 0.86;
      cfg = (c0 <= c1 && c0 <= c2) ? 0 : (c2 <= c1 ? 2 : 1);
    }
    if (g_x3_tile >= 0 && !bk && g.nsplit == 1) cfg = g_x3_tile;
    const int BMp = cfg == 0 ? 256 : cfg == 2 ? 192 : 128;
    g.NI = (int)ogl_cdiv(g.M, BMp);
    g.NJ = (int)ogl_cdiv(g.N, 128);
    const int64_t T = (int64_t)g.NI * g.NJ * g.nsplit;
    dim3 grid((unsigned)(8 * std::min<int64_t>(32, ogl_cdiv(T, 8)))), block(768);
    // staggered multiplier waves (see k_gemm_x3p) unless switched off
    static const char* stag_env = getenv("OGL_X3_STAGGER");
    // (3 = staggered + a static issue priority for waves 4-7: 0.9300-0.9366 / 0.9203-0.9259 / 0.9202-0.9219 ms per step at 0 / 1 / 3, one
    // box, three alternations — profiles/r06_ab_stagger.txt)
    g.stagger = g_x3_stagger >= 0 ? g_x3_stagger : (stag_env ? atoi(stag_env) : 3);
    // ring depth: two stages of the 256 x 128 tile fill the LDS (144 KB); the 128 x 128 tile takes three (144 KB): its movers
    // run two stages ahead (+4 % on the layer-0 weight gradient, whose operands both stream from HBM)
    // the two-stage tiles (256 x 128, 192 x 128) run in their early-A form (template parameter EA; the one-barrier form of the same
    // tiles — the same bits, 2.7 % slower per step, round 4 — left the library in round 6)
    if (bk && g.ak_groups > 0) X3P_LAUNCH(2, 4, 2, 1, 3, false, true, true);
    else if (bk && cfg == 0) X3P_LAUNCH(4, 2, 2, 2, 2, false, true, false, true);
    else if (bk) X3P_LAUNCH(2, 4, 2, 1, 3, false, true);
    else if (ext) {
      if (g.nsplit != 1 || g.ones_col) return OGL_EINVAL;
      if (cfg == 0) X3P_LAUNCH(4, 2, 2, 2, 2, true, false, false, true);
      else if (cfg == 2) X3P_LAUNCH(2, 4, 3, 1, 2, true, false, false, true);
      else X3P_LAUNCH(2, 4, 2, 1, 3, true);
    } else if (cfg == 0) X3P_LAUNCH(4, 2, 2, 2, 2, false, false, false, true);
    else if (cfg == 2) X3P_LAUNCH(2, 4, 3, 1, 2, false, false, false, true);
    else X3P_LAUNCH(2, 4, 2, 1, 3);
    OGL_CHECK_LAUNCH();
  } else {
    // (the 256 x 128 form of this kernel needs 64-bit piece addresses on top of 128 accumulators: it does not fit 256 registers
    // without spilling, so images of 4 GB and more take the 128 x 128 tile)
    const int BM = 128, BN = 128;
    g.NI = (int)ogl_cdiv(g.M, BM);
    g.NJ = (int)ogl_cdiv(g.N, BN);
    const int64_t T = (int64_t)g.NI * g.NJ * g.nsplit;
    dim3 grid((unsigned)(8 * std::min<int64_t>(32, ogl_cdiv(T, 8)))), block(512);   // persistent: at most one block per CU
    // DMA issue: spread between the MFMA groups for the 256 x 128 tile (2-3 % faster, A/B on one device), in one burst at
    // the top of the step for the 128 x 128 tile (its steps are too short to hide a late piece: spread measured 9 % slower)
    g_x3_last_kernel = "k_gemm_x3<2, 4, 2, 1, false>";
    hipLaunchKernelGGL((k_gemm_x3<2, 4, 2, 1, false>), grid, block, 0, stream, g);
    OGL_CHECK_LAUNCH();
  }
  if (g.nsplit > 1 && !g.defer_reduce) {
    hipLaunchKernelGGL(k_x3_splitk_reduce, dim3((unsigned)min((int64_t)2048, ogl_cdiv(g.M * g.N, 256))), dim3(256), 0, stream, g);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
}

// the 32-column groups [g0, g0 + ng) of a row-major image that an image-writing product's tiles did not reach (see
// ogl_linear_fwd_x3_ext): zeros, 1.0 at column N when the image carries the ones slot — rows 0 .. M, the zero row included
__global__ void __launch_bounds__(256) k_x3_image_tail(unsigned char* __restrict__ img, int64_t row_bytes, int64_t M, int N, int append_ones,
                                                       int g0, int ng) {
  const int64_t total = (M + 1) * (int64_t)ng * 4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / (ng * 4);
    const int u = (int)(t - r * (ng * 4)), gi = g0 + (u >> 2), c = u & 3;
    const int k = gi * 32 + c * 8;
    uint4 hi = make_uint4(0u, 0u, 0u, 0u);
    if (append_ones && N >= k && N < k + 8) {
      const int q = N - k;                                  // bf16 1.0 = 0x3F80 in element q of the chunk
      unsigned w[4] = {0u, 0u, 0u, 0u};
      w[q >> 1] = (q & 1) ? 0x3F800000u : 0x3F80u;
      hi = make_uint4(w[0], w[1], w[2], w[3]);
    }
    unsigned char* d = img + r * row_bytes + (int64_t)gi * X3_GROUP_BYTES;
    *(uint4*)(d + x3_piece(c, 0) * 16) = hi;
    *(uint4*)(d + x3_piece(c, 1) * 16) = make_uint4(0u, 0u, 0u, 0u);
    *(uint4*)(d + x3_piece(c, 2) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
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

// ogl_linear_fwd_x3 with the extensions of k_gemm_x3p<..., EXT>: a second A part (K-concatenated product), a per-row addend, the
// output's own bf16x3 image.  K1 / K2 = reduction lengths the A images were built with (their appended element included);
// the w image is K-concatenated: ceil(K1 / 32) + ceil(K2 / 32) groups per row.
extern "C" int ogl_linear_fwd_x3_ext(const void* x_img, int64_t x_img_rows, const int64_t* x_rows, int64_t x_nrows, int K1,
                                     const void* x2_img, int64_t x2_img_rows, const int64_t* x2_rows, int64_t x2_nrows, int K2,
                                     int64_t M, const void* w_img, int N, const float* add, int64_t ld_add, const int64_t* add_rows,
                                     int64_t add_nrows, int relu, float* y, int64_t ldy, void* out_img, int out_append_ones,
                                     const float* mask, int64_t ld_mask, const unsigned char* y_keep, ogl_stream_t stream) {
  if (M < 0 || K1 <= 0 || K2 < 0 || N < 0 || x_img_rows < 0 || x_nrows < 0 || x_nrows > x_img_rows || ldy < N) return OGL_EINVAL;
  if (M == 0 || N == 0) return OGL_OK;
  if (!x_img || !w_img || !y || (!x_rows && M > x_img_rows)) return OGL_EINVAL;
  if (K2 > 0 && (!x2_img || x2_img_rows < 0 || x2_nrows < 0 || x2_nrows > x2_img_rows || (!x2_rows && M > x2_img_rows))) return OGL_EINVAL;
  if (add && (ld_add < N || add_nrows < 0)) return OGL_EINVAL;
  if (out_img && ((uintptr_t)out_img & 15)) return OGL_EINVAL;
  if (mask && (ld_mask < N || (ld_mask & 3) || ((uintptr_t)mask & 15) || (N & 3))) return OGL_EINVAL;   // whole 16-byte groups only
  X3Args g = X3Args();
  const int G1 = (int)ogl_cdiv(K1, 32), G2 = (int)ogl_cdiv(K2, 32);
  g.a = X3Operand{(const unsigned char*)x_img, (int64_t)G1 * X3_GROUP_BYTES, X3_GROUP_BYTES, x_rows, x_rows ? x_nrows : x_img_rows,
                  x_img_rows};
  if (K2 > 0)
    g.a2 = X3Operand{(const unsigned char*)x2_img, (int64_t)G2 * X3_GROUP_BYTES, X3_GROUP_BYTES, x2_rows,
                     x2_rows ? x2_nrows : x2_img_rows, x2_img_rows};
  g.nsteps1 = G1;
  g.b = X3Operand{(const unsigned char*)w_img, (int64_t)(G1 + G2) * X3_GROUP_BYTES, X3_GROUP_BYTES, nullptr, N, N};
  g.M = M; g.N = N; g.nsteps = G1 + G2;
  g.C = y; g.ldc = ldy; g.relu = relu; g.nsplit = 1;
  g.add = add; g.ld_add = ld_add; g.add_rows = add_rows; g.add_nrows = add_nrows;
  g.out_img = (unsigned char*)out_img; g.out_append_ones = out_append_ones;
  g.out_row_bytes = ogl_cdiv(N + (out_append_ones ? 1 : 0), 32) * X3_GROUP_BYTES;
  g.mask = mask; g.ld_mask = ld_mask;
  if (y_keep && !out_img) return OGL_EINVAL;                // (rows without an fp32 copy must at least have their image)
  g.y_keep = y_keep;
  const int rc = launch_x3(g, (hipStream_t)stream);
  if (rc != OGL_OK || !out_img) return rc;
  // The epilogue writes the image columns its tiles cover: ceil(N / 128) x 128.  When N is a multiple of 128 and a ones slot is
  // appended, the image's last 32-column group (the slot at column N + its padding) lies past the last tile and stayed UNWRITTEN:
  // the next layer's fc_pool then multiplied whatever the allocation held (found in round 5: hidden width 256 -> logits of 1e37 in
  // the split-bf16 train forward; 600 and 32, the widths of the shipped settings and of every test until then, are covered).
  const int64_t covered = (int64_t)g.NJ * 128, img_cols = g.out_row_bytes / 6;
  if (img_cols > covered) {
    const int g0 = (int)(covered / 32), ng = (int)(img_cols / 32) - g0;
    const int64_t total = (M + 1) * (int64_t)ng * 4;
    hipLaunchKernelGGL(k_x3_image_tail, dim3((unsigned)std::min<int64_t>(ogl_cdiv(total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned char*)out_img, g.out_row_bytes, M, N, out_append_ones, g0, ng);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
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

// The weight gradient with the activations' ROW-MAJOR image as it is (k_gemm_x3p<..., BK>): dw = dy^T . x[rows] without a
// transposed image of x.  The split-K plan is the one of the 128 x 128 tile.
// The k-major weight gradient of a WIDE layer (>= 512 output rows, dy^T a transposed image: layer 0's dW_pool) runs on the 256 x 128
// two-stage tile: 72 KB of stage DMA per 256 x 128 x 32 step instead of 2 x 48 KB for the same work on 128 x 128 tiles — these
// products are paced by the LDS-DMA issue rate, not by the matrix pipe (a tile whose padding blocks skip their MFMAs is no faster:
// an uneven split-K that gave the 90-valid-row last tile longer reduction ranges was 15-40 % SLOWER, round 3) — measured
// 0.2385 -> 0.2330 ms for the Reddit dW_pool0 in spite of 602 rows filling only 2.35 of 3 row tiles.
static bool x3_bwwk_cfg0(int N, bool dy_rows) { return !dy_rows && N >= 512; }

// split plan of a k-major weight gradient: one round of blocks (one per CU), >= 8 steps per block.
// (An UNEVEN plan — longer reduction ranges for a last row tile that holds few valid rows and skips their DMA — was built in rounds
// 3-4, bit-checked, 8 % shorter per block and 15 us LONGER inside the train step: it left the library in round 6)
static void x3_bwwk_plan(int64_t steps, int N, int Kc, int* nsplit, int* sps, bool cfg0 = false) {
  const int64_t tiles = ogl_cdiv(N, cfg0 ? 256 : 128) * ogl_cdiv(Kc, 128);
  int64_t s = 256 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  if (steps / s < 8) s = steps / 8 > 0 ? steps / 8 : 1;
  *sps = (int)ogl_cdiv(steps > 0 ? steps : 1, s);
  *nsplit = (int)ogl_cdiv(steps > 0 ? steps : 1, *sps);
}

extern "C" int64_t ogl_linear_bwd_weight_x3k_workspace_bytes(int64_t M, int64_t interleave, int N, int K, int has_ones) {
  if (M < 0 || N < 0 || K < 0 || interleave < -1) return OGL_EINVAL;
  int nsplit, sps;
  x3_bwwk_plan(interleave > 0 ? interleave : ogl_cdiv(M, 32), N, K + (has_ones ? 1 : 0), &nsplit, &sps, x3_bwwk_cfg0(N, interleave == -1));
  if (nsplit <= 1) return 16;
  return (int64_t)nsplit * N * ogl_round_up(K + 1, 4) * 4;
}

static int bwd_weight_x3k(const void* dyT_img, int64_t interleave, const void* x_img, int64_t x_img_rows,
                          const int64_t* x_rows, int64_t x_nrows, int64_t M, int N, int K, int has_ones, float* dw,
                          int64_t lddw, float* db, float* db2, void* workspace, int64_t workspace_bytes,
                          ogl_stream_t stream, int defer, int* nsplit_out, int64_t* ws_ld_out) {
  // interleave == -1: `dyT_img` is the ROW-MAJOR image of dy itself ([M + 1 rows, N], ogl_relu_bwd_img / ogl_x3_split): read k-major too
  const bool dy_rows = interleave == -1;
  if (dy_rows) interleave = 0;
  if (M <= 0 || N < 0 || K <= 0 || lddw < K || interleave < 0 || (interleave > 0 && 32 * interleave < M)) return OGL_EINVAL;
  if (x_img_rows < 0 || x_nrows < 0 || x_nrows > x_img_rows || (!x_rows && M > x_img_rows)) return OGL_EINVAL;
  if (N == 0) return OGL_OK;
  if (!dw || !dyT_img || !x_img || ((db || db2) && !has_ones)) return OGL_EINVAL;
  X3Args g = X3Args();
  const int Kc = K + (has_ones ? 1 : 0);                   // columns of the product: dw and, from the ones slot, db
  const int64_t xrb = ogl_cdiv(Kc, 32) * X3_GROUP_BYTES;
  g.a = X3Operand{(const unsigned char*)dyT_img, X3_GROUP_BYTES, ((int64_t)N + 1) * X3_GROUP_BYTES, nullptr, N, N};
  g.b = X3Operand{(const unsigned char*)x_img, xrb, 0, x_rows, x_rows ? x_nrows : x_img_rows, x_img_rows};
  g.bk_red = M; g.bk_interleave = interleave; g.bk_groups = (int)ogl_cdiv(Kc, 32);
  if (dy_rows) {
    g.ak_groups = (int)ogl_cdiv(N, 32);
    g.a = X3Operand{(const unsigned char*)dyT_img, (int64_t)g.ak_groups * X3_GROUP_BYTES, 0, nullptr, M, M};
  }
  g.M = N; g.N = Kc; g.ones_col = has_ones ? 1 : 0;
  g.nsteps = (int)(interleave ? interleave : ogl_cdiv(M, 32));
  g.C = dw; g.ldc = lddw; g.db = db; g.db2 = db2;
  x3_bwwk_plan(g.nsteps, N, Kc, &g.nsplit, &g.steps_per_split, x3_bwwk_cfg0(N, dy_rows));
  g.force_cfg0 = x3_bwwk_cfg0(N, dy_rows) ? 1 : 0;
  g.xcd_slabs = g.nsplit >= 8 ? 1 : 0;                      // whole slabs per XCD (round 3: 1.014-1.017 -> 1.002-1.011 ms per step)
  if (g.nsplit > 1) {
    g.ws_ld = ogl_round_up(K + 1, 4);
    if (!workspace || workspace_bytes < (int64_t)g.nsplit * N * g.ws_ld * 4) return OGL_EWORKSPACE;
    g.ws = (float*)workspace;
  }
  g.defer_reduce = defer;
  if (nsplit_out) *nsplit_out = g.nsplit;
  if (ws_ld_out) *ws_ld_out = g.nsplit > 1 ? g.ws_ld : 0;
  return launch_x3(g, (hipStream_t)stream);
}

extern "C" int ogl_linear_bwd_weight_x3k(const void* dyT_img, int64_t interleave, const void* x_img, int64_t x_img_rows,
                                         const int64_t* x_rows, int64_t x_nrows, int64_t M, int N, int K, int has_ones, float* dw,
                                         int64_t lddw, float* db, float* db2, void* workspace, int64_t workspace_bytes,
                                         ogl_stream_t stream) {
  return bwd_weight_x3k(dyT_img, interleave, x_img, x_img_rows, x_rows, x_nrows, M, N, K, has_ones, dw, lddw, db, db2, workspace,
                        workspace_bytes, stream, 0, nullptr, nullptr);
}

// The same product with its split-K reduction LEFT TO THE CONSUMER: when *nsplit_out > 1 the call has written nsplit slabs
// workspace[s][N rows][*ws_ld_out floats] (s-th partial sum of [dw | db] over its range of the reduction, column K = the bias gradient)
// and NOTHING into dw / db / db2; the gradient is sum_s slab_s in slab order — what ogl_adam_step_multi*_slabs and ogl_x3_slab_reduce
// compute (the same order as the reduction launch of ogl_linear_bwd_weight_x3k: identical bits).  *nsplit_out == 1: dw / db / db2
// are final, as after ogl_linear_bwd_weight_x3k.  Four ~8 us reduction launches per Reddit train step exist only to sum slabs the
// optimiser is about to read.
extern "C" int ogl_linear_bwd_weight_x3k_slabs(const void* dyT_img, int64_t interleave, const void* x_img, int64_t x_img_rows,
                                               const int64_t* x_rows, int64_t x_nrows, int64_t M, int N, int K, int has_ones,
                                               float* dw, int64_t lddw, float* db, float* db2, void* workspace,
                                               int64_t workspace_bytes, int* nsplit_out, int64_t* ws_ld_out, ogl_stream_t stream) {
  if (!nsplit_out || !ws_ld_out) return OGL_EINVAL;
  return bwd_weight_x3k(dyT_img, interleave, x_img, x_img_rows, x_rows, x_nrows, M, N, K, has_ones, dw, lddw, db, db2, workspace,
                        workspace_bytes, stream, 1, nsplit_out, ws_ld_out);
}

// BOTH weight gradients of a dual-input projection y = x[x_rows] . w1^T + x2 . w2^T (+ biases) in ONE product (round 5):
// [dw1 | db | pad | dw2] = dy^T . [x[x_rows] | 1 | x2] with the two activations read where they lie — x as the resident table's
// row-major image (K1 + ones slot when has_ones), x2 as the image its producer wrote (K2) — through a two-part B operand by column tile
// (X3Args.b2): fc_self / fc_neigh of the live layer's combine (R/inference_optimized.py:136-139,276), fc_neigh(cat(h_self, h_neigh)) of the
// in-repo layer (R/train/graphsage/pytorch/aggregator_dgl.py:206).  dy_img: the ROW-MAJOR image of dy [M + 1 rows, N] (ogl_relu_bwd_img /
// ogl_x3_split).  The result is left as split-K slabs in `workspace`: slab s = workspace[s][N rows][*ws_ld_out floats] with dw1 in
// columns [0, K1), the bias gradient in column K1 (has_ones), dw2 in columns [*col2_out, *col2_out + K2); consumers sum the
// *nsplit_out slabs in slab order (ogl_adam_step_multi_slabs, ogl_x3_slab_reduce).  One launch of 50 tiles x 5 slabs x 44 steps where
// the two separate products were 2 x (25 tiles x 10 slabs x 22 steps): the same MFMAs, half the pipeline fills and epilogues.
extern "C" int64_t ogl_linear_bwd_weight_x3k_dual_workspace_bytes(int64_t M, int N, int K1, int has_ones, int K2) {
  if (M <= 0 || N <= 0 || K1 <= 0 || K2 <= 0) return OGL_EINVAL;
  const int64_t cols = ogl_cdiv(K1 + (has_ones ? 1 : 0), 128) * 128 + ogl_round_up(K2, 4);
  int nsplit, sps;
  x3_bwwk_plan(ogl_cdiv(M, 32), N, (int)cols, &nsplit, &sps);
  if (nsplit < 2) return 0;                                 // (no slab form for this shape: the caller runs two products)
  return (int64_t)nsplit * N * ogl_round_up(cols, 4) * 4 + 16;
}

extern "C" int ogl_linear_bwd_weight_x3k_dual_slabs(const void* dy_img, int64_t M, int N, const void* x_img, int64_t x_img_rows,
                                                    const int64_t* x_rows, int64_t x_nrows, int K1, int has_ones, const void* x2_img,
                                                    int64_t x2_img_rows, int K2, void* workspace, int64_t workspace_bytes, int* nsplit_out,
                                                    int64_t* ws_ld_out, int* col2_out, ogl_stream_t stream) {
  if (M <= 0 || N <= 0 || K1 <= 0 || K2 <= 0 || !nsplit_out || !ws_ld_out || !col2_out) return OGL_EINVAL;
  if (!dy_img || !x_img || !x2_img || x_img_rows < 0 || x_nrows < 0 || x_nrows > x_img_rows || (!x_rows && M > x_img_rows) || M > x2_img_rows)
    return OGL_EINVAL;
  X3Args g = X3Args();
  const int Kc1 = K1 + (has_ones ? 1 : 0);
  g.NJ1 = (int)ogl_cdiv(Kc1, 128);
  const int col2 = g.NJ1 * 128;
  g.ak_groups = (int)ogl_cdiv(N, 32);
  g.a = X3Operand{(const unsigned char*)dy_img, (int64_t)g.ak_groups * X3_GROUP_BYTES, 0, nullptr, M, M};
  g.b = X3Operand{(const unsigned char*)x_img, ogl_cdiv(Kc1, 32) * X3_GROUP_BYTES, 0, x_rows, x_rows ? x_nrows : x_img_rows, x_img_rows};
  g.b2 = X3Operand{(const unsigned char*)x2_img, ogl_cdiv(K2, 32) * X3_GROUP_BYTES, 0, nullptr, x2_img_rows, x2_img_rows};
  g.bk_red = M; g.bk_interleave = 0; g.bk_groups = (int)ogl_cdiv(Kc1, 32); g.bk2_groups = (int)ogl_cdiv(K2, 32);
  g.M = N; g.N = col2 + K2; g.ones_col = 0;
  g.nsteps = (int)ogl_cdiv(M, 32);
  x3_bwwk_plan(g.nsteps, N, (int)g.N, &g.nsplit, &g.steps_per_split);
  if (g.nsplit < 2) return OGL_EINVAL;                      // (the slab form only: a one-split product has one output pointer)
  g.ws_ld = ogl_round_up(g.N, 4);
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < (int64_t)g.nsplit * N * g.ws_ld * 4) return OGL_EWORKSPACE;
  g.ws = (float*)workspace;
  g.C = g.ws; g.ldc = g.ws_ld;                              // (never written: nsplit > 1)
  g.xcd_slabs = g.nsplit >= 8 ? 1 : 0;
  g.defer_reduce = 1;
  *nsplit_out = g.nsplit; *ws_ld_out = g.ws_ld; *col2_out = col2;
  return launch_x3(g, (hipStream_t)stream);
}

// out[r, c] = sum_s ws[s * slab_stride + r * ws_ld + col0 + c] (slab order): the reduction of deferred slabs for a consumer that is
// not the optimiser launch (a gradient somebody reads before the step).
__global__ void __launch_bounds__(256) k_x3_slab_reduce(const float* __restrict__ ws, int64_t slab_stride, int64_t ws_ld, int nsplit,
                                                        int64_t rows, int ncols, int col0, float* __restrict__ out, int64_t ldo) {
  const int64_t total = rows * ncols;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / ncols, col = t - row * ncols;
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += ws[(int64_t)s * slab_stride + row * ws_ld + col0 + col];
    out[row * ldo + col] = v;
  }
}

extern "C" int ogl_x3_slab_reduce(const float* ws, int64_t slab_stride, int64_t ws_ld, int nsplit, int64_t rows, int ncols, int col0,
                                  float* out, int64_t ldo, ogl_stream_t stream) {
  if (nsplit <= 0 || rows < 0 || ncols <= 0 || col0 < 0 || ws_ld < col0 + ncols || ldo < ncols || slab_stride < rows * ws_ld) return OGL_EINVAL;
  if (rows == 0) return OGL_OK;
  if (!ws || !out) return OGL_EINVAL;
  hipLaunchKernelGGL(k_x3_slab_reduce, dim3((unsigned)min((int64_t)2048, ogl_cdiv(rows * ncols, 256))), dim3(256), 0, (hipStream_t)stream,
                     ws, slab_stride, ws_ld, nsplit, rows, ncols, col0, out, ldo);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
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
