// Dense projections of SAGEConv (fc_pool / fc_self / fc_neigh = torch.nn.Linear in the reference,
// R/train/graphsage/pytorch/aggregator_dgl.py:85-94,171,181,206; DGL 'pool' parameterisation
// R/inference_optimized.py:136-139,260,276) as ONE LDS-tiled fp32-MFMA GEMM kernel.
//
// MFMA-bound (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD, 157 TFLOP/s peak).
//   C[i,j] = epilogue( sum_parts sum_r  A(i,r) * B(r,j) )
// Every operand X(p,r) (p = output index, r = reduction index) comes in one of two layouts
//   RC ("reduction-contiguous"):  X(p,r) = ptr[row(p)*ld + r]
//   NC ("non-reduction-contiguous"): X(p,r) = ptr[row(r)*ld + p]
// which covers forward (x RC, w RC), input-gradient (dy RC, w NC) and weight-gradient (dy NC, x NC)
// without materialising a transpose.  `row()` is an optional int64 gather (feature rows are read
// straight from the resident table), the ReLU mask of a
// backward pass is applied by ogl_relu_bwd before the GEMM (nothing in the loaders consumes a loaded value).
// Tiles are staged k-major in LDS (Xs[r][p], row stride P+4) so each MFMA operand is one
// conflict-free ds_read_b32; global loads are 16 B/lane when the operand is 16-B aligned.
// Block -> tile mapping is XCD-aware and bijective (see k_gemm): consecutive logical tiles, which share
// an operand panel, run on one XCD so the panel is fetched into that XCD's L2 once.
#include "ogl_common.h"

#include "x6_arith.h"

#define X6_ROW_U4 7   // LDS row of a P x 16 tile: 3 splits x 2 k-halves x 16 B + 16 B pad = 112 B (odd multiple of 16 B)

#define GEMM_BK 16
#define GEMM_THREADS 256

struct Operand {
  const float* ptr;
  int64_t ld;
  const int64_t* rows;  // optional gather on the major (row) index
  int64_t nrows;        // bound for gathered rows (rows outside -> zeros)
};

struct GemmPart {
  Operand a, b;
  int64_t R;  // reduction length of this part
};

struct GemmArgs {
  GemmPart part[2];
  int nparts;
  int64_t M;        // i range (output rows)
  int64_t N;        // j range (output cols), includes the synthetic ones column if ones_col
  int ones_col;     // B(r, N-1) == 1  -> column N-1 of the result = row sums of A (bias gradient)
  float* C; int64_t ldc;
  const float* bias;  // [N] or null, added per output column
  const float* bias2; // [N] or null (with bias): the second projection's own bias — (bias + bias2) is formed first, then added
  // optional per-ROW addend (forward only, nsplit == 1): C[i, :] += add[row(i), :], row(i) = add_rows ? add_rows[i] : i; rows
  // outside [0, add_nrows) add nothing.  The self term of an inference layer read from a per-vertex table (S0[dst]).
  const float* add; int64_t ld_add; const int64_t* add_rows; int64_t add_nrows;
  int relu;
  float* db;        // destination of column N-1 when ones_col
  int nsplit;       // >1: partial sums go to ws[split][M][ws_ld], epilogue runs in k_splitk_reduce
  int tiles_per_split;
  float* ws; int64_t ws_ld;
  int NI, NJ;
  int force_cfg;    // host-side only: 1 + tile configuration chosen by the caller's plan, 0 = automatic
};

// Stages one P x BK operand tile: global -> registers (load) -> LDS k-major (store).
// All per-thread addressing that does not depend on the k-tile is hoisted into init(); interior
// tiles take a branch-free path (invalid rows read row 0 and are zeroed by a select), only the
// last partial k-tile / partial column group takes guarded dword loads.
template <int P, bool RC, bool ONES_ROW = false>
struct TileLoader {
  static constexpr int NV = P / 64;  // float4 per thread
  const float* ptr[NV];   // RC: row base + this thread's k offset.  NC: unused
  bool ok[NV];
  bool one[NV];           // RC: this row is the synthetic all-ones row (bias gradient of the transposed weight-gradient form)
  int pcol;               // NC: first of this thread's 4 columns
  int pmode;              // NC: 0 none, 1 one 16-B load (fixed up later), 2 guarded dword loads (tight rows)
  int keep;               // NC: how many of the 4 columns exist in memory
  int onesq;              // NC: component holding the synthetic ones column, or -1
  int nrow[NV];           // NC + gather: row id of the NEXT k-tile (fetched one tile ahead)

  __device__ __forceinline__ void init(const Operand& op, int64_t p0, int64_t Plim, int64_t ones_p, int tid,
                                       int64_t r_first, int64_t R) {
    if (RC) {
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const int64_t p = p0 + (tid >> 2) + 64 * h;
        bool v = p < Plim;
        int64_t row = p;
        if (op.rows) {
          row = v ? op.rows[p] : 0;
          v = v && row >= 0 && row < op.nrows;
        }
        one[h] = ONES_ROW && p == ones_p;
        if (one[h]) v = false;
        if (!v) row = 0;
        ok[h] = v;
        ptr[h] = op.ptr + row * op.ld + (tid & 3) * 4;
      }
    } else {
      pcol = (int)(p0 + (tid % (P / 4)) * 4);
      const int64_t psrc = ones_p >= 0 ? ones_p : Plim;
      const int64_t left = psrc - pcol;
      keep = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
      onesq = (ones_p >= pcol && ones_p < pcol + 4) ? (int)(ones_p - pcol) : -1;
      // a partial group is still one 16-B load when the row stride leaves room for it (padded matrices): the pad
      // is read and discarded at fix-up time, so no lane diverges inside the k-loop
      const bool vec_ok = keep == 4 || (keep > 0 && op.ld >= pcol + 4) || (keep == 0 && onesq >= 0);
      pmode = pcol >= Plim ? 0 : (vec_ok ? 1 : 2);
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const int64_t r = r_first + (tid / (P / 4)) * NV + h;
        nrow[h] = (op.rows && r < R) ? (int)op.rows[r] : 0;   // row ids fit 31 bits (checked at graph create)
      }
    }
  }

  // RAW loads.  On the hot paths (RC interior tile, NC 16-B group) NOTHING here consumes a loaded value — no select,
  // no compare — so the compiler's s_waitcnt lands in fix(), after the MFMAs of the tile being multiplied, and the
  // loads of a wave overlap its own matrix work.  Lanes that must not contribute read a safe address (row 0) and are
  // zeroed in fix().  `bad` collects gathered rows that were out of range (bit h).
  __device__ __forceinline__ void load(const Operand& op, int64_t Plim, int64_t r0, int64_t R, int64_t ones_p, int tid,
                                       float4 (&reg)[NV], unsigned& bad) {
    bad = 0;
    if (RC) {
      const bool interior = r0 + GEMM_BK <= R;  // block-uniform
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        if (!ok[h]) bad |= 1u << h;             // validity travels with the staging set (the loader may have moved on
                                                // to the second part of a dual GEMM by the time this tile is fixed up)
        if (ONES_ROW && one[h]) bad |= 1u << (8 + h);
        if (interior) {
          reg[h] = ld16(ptr[h] + r0);
        } else {                                // last, partial k-tile of the part: guarded dwords (once per block)
          const int64_t r = r0 + (tid & 3) * 4;
          float e[4] = {0.f, 0.f, 0.f, 0.f};
          if (ok[h]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (r + q < R) e[q] = ptr[h][r0 + q];
          }
          reg[h] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const int64_t r = r0 + (tid / (P / 4)) * NV + h;
        const bool rv = r < R;
        int64_t row = rv ? r : 0;
        if (op.rows) {
          row = nrow[h];                                       // fetched while the previous tile was computed
          nrow[h] = (r + GEMM_BK < R) ? (int)op.rows[r + GEMM_BK] : 0;
          if (!rv || row < 0 || row >= op.nrows) { if (rv) bad |= 1u << h; row = 0; }
        }
        if (pmode == 1 && keep > 0) {
          reg[h] = ld16(op.ptr + (int64_t)((uint64_t)(uint32_t)row * (uint32_t)op.ld) + pcol);
        } else if (pmode == 2 && rv && !((bad >> h) & 1)) {    // tight rows (ld < pcol + 4): guarded dwords
          const int64_t psrc = ones_p >= 0 ? ones_p : Plim;
          float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (pcol + q < psrc) e[q] = op.ptr[row * op.ld + pcol + q];
          reg[h] = make_float4(e[0], e[1], e[2], e[3]);
        } else {
          reg[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }

  // Fix-ups of a loaded tile, applied right before it is written to LDS: zero the lanes that must not contribute,
  // drop pad columns, plant the synthetic ones column.  Pure selects.
  __device__ __forceinline__ void fix(float4 (&reg)[NV], int64_t r0, int64_t R, int tid, unsigned bad) const {
    if (RC) {
      const int64_t rr = r0 + (tid & 3) * 4;        // pure selects: no branch may split the MFMA / staging region
      const float4 ones = make_float4(rr < R ? 1.f : 0.f, rr + 1 < R ? 1.f : 0.f, rr + 2 < R ? 1.f : 0.f, rr + 3 < R ? 1.f : 0.f);
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const bool z = (bad >> h) & 1, o = ONES_ROW && ((bad >> (8 + h)) & 1);
        float4 v = reg[h];
        v.x = o ? ones.x : (z ? 0.f : v.x); v.y = o ? ones.y : (z ? 0.f : v.y);
        v.z = o ? ones.z : (z ? 0.f : v.z); v.w = o ? ones.w : (z ? 0.f : v.w);
        reg[h] = v;
      }
    } else {
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        const int64_t r = r0 + (tid / (P / 4)) * NV + h;
        float4 v = reg[h];
        const bool dead = !(r < R) || pmode == 0 || ((bad >> h) & 1);
        const bool oneok = r < R && pmode != 0;
        v.x = (dead || keep < 1) ? 0.f : v.x; v.y = (dead || keep < 2) ? 0.f : v.y;
        v.z = (dead || keep < 3) ? 0.f : v.z; v.w = (dead || keep < 4) ? 0.f : v.w;
        v.x = (oneok && onesq == 0) ? 1.f : v.x; v.y = (oneok && onesq == 1) ? 1.f : v.y;
        v.z = (oneok && onesq == 2) ? 1.f : v.z; v.w = (oneok && onesq == 3) ? 1.f : v.w;
        reg[h] = v;
      }
    }
  }

  // split-bf16 image: Xs[p][s*2 + kh] = the 8 bf16 of split s, k = 8*kh .. 8*kh+7, of row p (MFMA operand order)
  __device__ __forceinline__ void store_x6(uint4 (*Xs)[X6_ROW_U4], int tid, const float4 (&reg)[NV]) const {
#pragma unroll
    for (int h = 0; h < NV; ++h) {
      if (RC) {
        const int pl = (tid >> 2) + 64 * h, q = tid & 3;            // k = 4q .. 4q+3
        unsigned h0, m0, l0, h1, m1, l1;
        split3(reg[h].x, reg[h].y, h0, m0, l0);
        split3(reg[h].z, reg[h].w, h1, m1, l1);
        uint2* row = (uint2*)&Xs[pl][0];                             // 8-B granules: granule = chunk*2 + (q&1)
        row[(0 * 2 + (q >> 1)) * 2 + (q & 1)] = make_uint2(h0, h1);
        row[(1 * 2 + (q >> 1)) * 2 + (q & 1)] = make_uint2(m0, m1);
        row[(2 * 2 + (q >> 1)) * 2 + (q & 1)] = make_uint2(l0, l1);
      }
    }
  }

  // split-bf16 image of an NC-sourced tile: kept k-major exactly as it arrives, Xk[s][k][p] (row stride P + 32
  // halfwords = conflict-free for the transposed reads); the MFMA fragments are gathered by ds_read_b64_tr_b16.
  __device__ __forceinline__ void store_x6k(unsigned short (*Xk)[GEMM_BK][P + 32], int tid, const float4 (&reg)[NV]) const {
#pragma unroll
    for (int h = 0; h < NV; ++h) {
      const int rl = (tid / (P / 4)) * NV + h;
      const int pl = (tid % (P / 4)) * 4;
      unsigned h0, m0, l0, h1, m1, l1;
      split3(reg[h].x, reg[h].y, h0, m0, l0);
      split3(reg[h].z, reg[h].w, h1, m1, l1);
      *(uint2*)&Xk[0][rl][pl] = make_uint2(h0, h1);
      *(uint2*)&Xk[1][rl][pl] = make_uint2(m0, m1);
      *(uint2*)&Xk[2][rl][pl] = make_uint2(l0, l1);
    }
  }

  __device__ __forceinline__ void store(float (*Xs)[P + 4], int tid, const float4 (&reg)[NV]) const {
#pragma unroll
    for (int h = 0; h < NV; ++h) {
      if (RC) {
        const int pl = (tid >> 2) + 64 * h;
        const int rl = (tid & 3) * 4;
        Xs[rl + 0][pl] = reg[h].x; Xs[rl + 1][pl] = reg[h].y;
        Xs[rl + 2][pl] = reg[h].z; Xs[rl + 3][pl] = reg[h].w;
      } else {
        const int rl = (tid / (P / 4)) * NV + h;
        const int pl = (tid % (P / 4)) * 4;
        *(float4*)&Xs[rl][pl] = reg[h];
      }
    }
  }
};

template <bool A_RC, bool B_RC, int WAVES_M, int WAVES_N, int TM, int TN, bool X6, bool B_ONES = false>
__global__ void __launch_bounds__(GEMM_THREADS) k_gemm(GemmArgs g) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  // one LDS arena, two images: fp32 k-major tiles (exact fp32 MFMA) or split-bf16 operand-order tiles (x6)
  constexpr int A_BYTES = X6 ? (A_RC ? BM * X6_ROW_U4 * 16 : 3 * GEMM_BK * (BM + 32) * 2) : GEMM_BK * (BM + 4) * 4;
  constexpr int B_BYTES = X6 ? (B_RC ? BN * X6_ROW_U4 * 16 : 3 * GEMM_BK * (BN + 32) * 2) : GEMM_BK * (BN + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
  float (*As)[GEMM_BK][BM + 4] = (float (*)[GEMM_BK][BM + 4])smem;
  float (*Bs)[GEMM_BK][BN + 4] = (float (*)[GEMM_BK][BN + 4])(smem + 2 * A_BYTES);
  uint4 (*Ax)[BM][X6_ROW_U4] = (uint4 (*)[BM][X6_ROW_U4])smem;                                   // RC-sourced x6 image
  uint4 (*Bx)[BN][X6_ROW_U4] = (uint4 (*)[BN][X6_ROW_U4])(smem + 2 * A_BYTES);
  unsigned short (*Ak)[3][GEMM_BK][BM + 32] = (unsigned short (*)[3][GEMM_BK][BM + 32])smem;     // NC-sourced x6 image
  unsigned short (*Bk)[3][GEMM_BK][BN + 32] = (unsigned short (*)[3][GEMM_BK][BN + 32])(smem + 2 * A_BYTES);

  // XCD-aware, bijective tile mapping.  Hardware deals block L to XCD L % 8; the logical tile space
  // (split, row panel, column tile), column tile fastest, is cut into 8 contiguous chunks and chunk c
  // is served by the blocks with L % 8 == c.  Consecutive logical tiles share the A row panel (and, for
  // split-K, the same slice of both operands), so each XCD's L2 fetches a panel once.  No padding: every
  // XCD gets work whatever NI is.  Placement only affects speed, never results.
  const int T = g.NI * g.NJ * g.nsplit;
  const int L = blockIdx.x;
  const int xcd = L & 7, kth = L >> 3;
  const int base = T >> 3, rem = T & 7;
  const int logical = xcd * base + min(xcd, rem) + kth;
  const int split = logical / (g.NI * g.NJ);
  const int tile = logical - split * (g.NI * g.NJ);
  const int ti = tile / g.NJ, tj = tile - ti * g.NJ;
  const int64_t i0 = (int64_t)ti * BM, j0 = (int64_t)tj * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int l31 = lane & 31, half = lane >> 5;

  int nk[2];
  nk[0] = (int)((g.part[0].R + GEMM_BK - 1) / GEMM_BK);
  nk[1] = g.nparts > 1 ? (int)((g.part[1].R + GEMM_BK - 1) / GEMM_BK) : 0;
  const int nk_total = nk[0] + nk[1];
  const int64_t R_part0 = g.part[0].R, R_part1 = g.nparts > 1 ? g.part[1].R : 0;
  int kt_begin = 0, kt_end = nk_total;
  if (g.nsplit > 1) {
    kt_begin = split * g.tiles_per_split;
    kt_end = min(nk_total, kt_begin + g.tiles_per_split);
  }

  f32x16 acc[TM][TN];
  if (g.add && g.nsplit == 1) {
    // the per-row addend is the accumulators' INITIAL value (D layout, see the epilogue): its gathered loads go out with the
    // first operand tiles instead of after the last MFMA — as an epilogue step this was slower than the product itself at
    // K = 128 (arxiv rung: 41 vs 25 us)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = i0 + wm * TM * 32 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        int64_t ar = -1;
        if (row < g.M) ar = g.add_rows ? g.add_rows[row] : row;
        const bool ok = ar >= 0 && ar < g.add_nrows;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int64_t col = j0 + wn * TN * 32 + b * 32 + l31;
          acc[a][b][e] = (ok && col < g.N) ? g.add[ar * g.ld_add + col] : 0.f;
        }
      }
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  }

  TileLoader<BM, A_RC> la;
  TileLoader<BN, B_RC, B_ONES> lb;   // the all-ones row exists only in the transposed weight-gradient product
  float4 ra0[BM / 64], rb0[BN / 64], ra1[BM / 64], rb1[BN / 64];   // two register staging sets (2-tile-deep prefetch)
  unsigned ma0 = 0, mb0 = 0, ma1 = 0, mb1 = 0;                      // per-set "gathered row out of range" bits
  const int64_t ones_p = g.ones_col ? g.N - 1 : -1;
  int cur_part = -1;

  // tiles at or beyond kt_end (the pipeline below always runs an even number of tiles) load as zeros
  auto issue = [&](int kt, float4 (&ra)[BM / 64], float4 (&rb)[BN / 64], unsigned& ma, unsigned& mb) {
    const bool live = kt < kt_end;
    const int pi = (live && kt >= nk[0]) ? 1 : 0;
    const GemmPart& pt = g.part[pi];
    const int64_t r0 = live ? (int64_t)(pi == 0 ? kt : kt - nk[0]) * GEMM_BK : 0;
    const int64_t Reff = live ? pt.R : 0;
    if (pi != cur_part) {   // block-uniform: (re)derive the per-thread row pointers of this part
      la.init(pt.a, i0, g.M, -1, tid, r0, Reff);
      lb.init(pt.b, j0, g.N, ones_p, tid, r0, Reff);
      cur_part = pi;
    }
    la.load(pt.a, g.M, r0, Reff, -1, tid, ra, ma);
    lb.load(pt.b, g.N, r0, Reff, ones_p, tid, rb, mb);
  };

  // the tile in a staging set is fixed up (this is where its loads are first waited for) and written to LDS
  auto stage = [&](int buf, int kt, float4 (&ra)[BM / 64], float4 (&rb)[BN / 64], unsigned ma, unsigned mb) {
    {
      // scalar selects only (R of both parts is held in SGPRs): a dynamic g.part[pi] access would put a branch and a
      // scalar load between the MFMAs and the staging code and split the region the scheduler interleaves
      const bool live = kt < kt_end;
      const bool second = live && kt >= nk[0];
      const int64_t r0 = live ? (int64_t)(second ? kt - nk[0] : kt) * GEMM_BK : 0;
      const int64_t Reff = live ? (second ? R_part1 : R_part0) : 0;
      la.fix(ra, r0, Reff, tid, ma);
      lb.fix(rb, r0, Reff, tid, mb);
    }
    if (X6) {
      if (A_RC) la.store_x6(Ax[buf], tid, ra); else la.store_x6k(Ak[buf], tid, ra);
      if (B_RC) lb.store_x6(Bx[buf], tid, rb); else lb.store_x6k(Bk[buf], tid, rb);
    } else {
      la.store(As[buf], tid, ra);
      lb.store(Bs[buf], tid, rb);
    }
  };

  auto compute = [&](int buf) {
    if (X6) {
      // one 16-deep MFMA step per tile: operand fragments are single ds_read_b128 (conflict-free: 112-B rows)
      bf16x8 a[TM][3], b[TN][3];
      // transposed-read lane roles (per 16-lane group): lane 4q+p supplies row q, columns 4p..4p+3 of a 4 x 16 block and
      // receives column (lane & 15); groups 0/1 are rows 0-15 / 16-31 of the fragment, lanes 32+ the upper k half
      const int tq = (lane & 15) >> 2, tp = lane & 3, tc = 16 * ((lane >> 4) & 1);
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
          if (A_RC) {
            a[t][sp] = __builtin_bit_cast(bf16x8, Ax[buf][wm * TM * 32 + t * 32 + l31][sp * 2 + half]);
          } else {
            const int c = wm * TM * 32 + t * 32 + tc + 4 * tp;
            v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)&Ak[buf][sp][8 * half + tq][c]);
            v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)&Ak[buf][sp][8 * half + 4 + tq][c]);
            a[t][sp] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        }
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
          if (B_RC) {
            b[t][sp] = __builtin_bit_cast(bf16x8, Bx[buf][wn * TN * 32 + t * 32 + l31][sp * 2 + half]);
          } else {
            const int c = wn * TN * 32 + t * 32 + tc + 4 * tp;
            v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)&Bk[buf][sp][8 * half + tq][c]);
            v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)&Bk[buf][sp][8 * half + 4 + tq][c]);
            b[t][sp] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        }
#pragma unroll
      for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) {
          // smallest terms first (i + j = 4, then 3, then 2)
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][2], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][1], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][2], b[y][0], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][1], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][0], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][0], acc[x][y], 0, 0, 0);
        }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < GEMM_BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = As[buf][kk + half][wm * TM * 32 + t * 32 + l31];
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = Bs[buf][kk + half][wn * TN * 32 + t * 32 + l31];
#pragma unroll
      for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
    }
  };

  // Pipeline: while tile t is multiplied out of LDS buffer t&1, tile t+1 sits in one register set (written to the
  // other LDS buffer after the MFMAs) and tile t+2 is in flight into the other set: global latency is covered by
  // two tiles of MFMA work.  The loop is unrolled by two so both sets are statically indexed (no scratch).
  issue(kt_begin, ra0, rb0, ma0, mb0);
  stage(0, kt_begin, ra0, rb0, ma0, mb0);
  issue(kt_begin + 1, ra1, rb1, ma1, mb1);
  __syncthreads();

  // one pipeline step: multiply tile t out of LDS buffer `buf` while tile t+1 (already in registers) is converted
  // and written to the other buffer.  In x6 mode the split/pack VALU work and the LDS writes are interleaved with the
  // 24 MFMAs (one MFMA : ~4 VALU : DS write every other MFMA) so the matrix pipe is not left idle behind them.
  auto step = [&](int buf, int kt_next, float4 (&ra)[BM / 64], float4 (&rb)[BN / 64], unsigned ma, unsigned mb) {
    compute(buf);
    stage(buf ^ 1, kt_next, ra, rb, ma, mb);
    if (X6) {
      __builtin_amdgcn_sched_group_barrier(0x100, (A_RC ? 1 : 2) * TM * 3 + (B_RC ? 1 : 2) * TN * 3, 0);   // fragment ds_reads first
#pragma unroll
      for (int i = 0; i < TM * TN * 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);               // 4 VALU (split / pack / address)
        if (i & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // 1 DS write
      }
    }
  };

  // the loop body is branch-free: an odd tile count is padded with one all-zero tile (see issue())
  for (int kt = kt_begin; kt < kt_end; kt += 2) {
    issue(kt + 2, ra0, rb0, ma0, mb0);
    step(0, kt + 1, ra1, rb1, ma1, mb1);   // tile kt out of buffer 0, tile kt+1 -> buffer 1
    __syncthreads();
    issue(kt + 3, ra1, rb1, ma1, mb1);
    step(1, kt + 2, ra0, rb0, ma0, mb0);   // tile kt+1 out of buffer 1, tile kt+2 -> buffer 0
    __syncthreads();
  }

  // epilogue: D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int x = 0; x < TM; ++x) {
#pragma unroll
    for (int y = 0; y < TN; ++y) {
      const int64_t col = j0 + wn * TN * 32 + y * 32 + l31;
      if (col >= g.N) continue;
      const float bv = (g.nsplit == 1 && g.bias && !(g.ones_col && col == g.N - 1)) ? (g.bias2 ? g.bias[col] + g.bias2[col] : g.bias[col]) : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = i0 + wm * TM * 32 + x * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row >= g.M) continue;
        float v = acc[x][y][e];
        if (g.nsplit > 1) {
          g.ws[((int64_t)split * g.M + row) * g.ws_ld + col] = v;
        } else {
          v += bv;
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.ones_col && col == g.N - 1) { if (g.db) g.db[row] = v; }
          else g.C[row * g.ldc + col] = v;
        }
      }
    }
  }
}


// ---- skinny forward GEMM: few rows x few columns x long K (the [B, 2H] -> C output projection) -------
// One 32-row x 32-column output tile per block; the SK_WAVES waves split K (wave w takes the 8-deep chunks
// c = w mod SK_WAVES), operands go global -> registers directly (16 B per lane: row l&31, k = k0 + 4*(l>>5) ..+3;
// MFMA step j uses component j of both operands, so half 0 supplies k0+j and half 1 supplies k0+4+j),
// partial accumulators are summed through LDS.  Launch-latency bound shapes only: few blocks, long K — the kernel is
// a serial chain of global round trips followed by exact-fp32 MFMAs on a handful of CUs, so what it costs is the
// NUMBER of rounds and the MFMAs per block.  A round is up to SK_U chunks per wave OF EACH INPUT with every load
// issued before the first MFMA and nothing between the loads that waits (a chunk past the end reads chunk 0 and is
// zeroed): [512, 600 + 600] -> 41 is ONE round on 16 x 2 blocks (4 waves x 4 chunks on 16 blocks of 64 columns took
// ten rounds and twice the MFMAs per block: 26 us).
#define SK_WAVES 8
#define SK_U 10
template <int NPARTS>
__global__ void __launch_bounds__(64 * SK_WAVES) k_gemm_skinny(GemmArgs g) {
  __shared__ float red[SK_WAVES][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  const int64_t j0 = (int64_t)blockIdx.y * 32;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int64_t i = i0 + l31, j = j0 + l31;
  const bool bok = j < g.N;
  const float* ap[NPARTS];
  const float* bp[NPARTS];
  bool aok[NPARTS];
  int64_t nfull[NPARTS], most = 0;
#pragma unroll
  for (int p = 0; p < NPARTS; ++p) {
    const GemmPart& pt = g.part[p];
    aok[p] = i < g.M;
    int64_t arow = i;
    if (pt.a.rows) { arow = aok[p] ? pt.a.rows[i] : 0; aok[p] = aok[p] && arow >= 0 && arow < pt.a.nrows; }
    ap[p] = pt.a.ptr + (aok[p] ? arow : 0) * pt.a.ld;
    bp[p] = pt.b.ptr + (bok ? j : 0) * pt.b.ld;
    nfull[p] = pt.R / 8;                                   // whole 8-deep chunks
    most = nfull[p] > most ? nfull[p] : most;
  }
  for (int64_t c = wid; c < most; c += SK_WAVES * SK_U) {
    float4 a4[NPARTS][SK_U], b4[NPARTS][SK_U];
#pragma unroll
    for (int p = 0; p < NPARTS; ++p)
#pragma unroll
      for (int u = 0; u < SK_U; ++u) {
        const int64_t cu = c + SK_WAVES * u;
        const int64_t k = (cu < nfull[p] ? cu : 0) * 8 + 4 * half;
        a4[p][u] = ld16(ap[p] + k);
        b4[p][u] = ld16(bp[p] + k);
      }
#pragma unroll
    for (int p = 0; p < NPARTS; ++p)
#pragma unroll
      for (int u = 0; u < SK_U; ++u) {
        const bool live = c + SK_WAVES * u < nfull[p];     // wave-uniform
        const float4 a = (aok[p] && live) ? a4[p][u] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = bok ? b4[p][u] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
  }
#pragma unroll
  for (int p = 0; p < NPARTS; ++p) {
    const int64_t R = g.part[p].R;
    if ((R & 7) && wid == (int)(nfull[p] % SK_WAVES)) {   // the ragged last chunk: element-wise, one wave
      const int64_t k = nfull[p] * 8 + 4 * half;
      float ea[4] = {0.f, 0.f, 0.f, 0.f}, eb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (aok[p] && k + q < R) ea[q] = ap[p][k + q];
        if (bok && k + q < R) eb[q] = bp[p][k + q];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[0], eb[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[1], eb[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[2], eb[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[3], eb[3], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wid][e][lane] = acc[e];
  __syncthreads();
  // the block's threads sum the SK_WAVES partials (fixed order): 16*64 values
  for (int v = tid; v < 16 * 64; v += 64 * SK_WAVES) {
    const int e = v / 64, ln = v % 64;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) sum += red[w][e][ln];
    const int64_t row = i0 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
    const int64_t col = j0 + (ln & 31);
    if (row < g.M && col < g.N) {
      if (g.bias) sum += g.bias2 ? g.bias[col] + g.bias2[col] : g.bias[col];
      if (g.relu) sum = fmaxf(sum, 0.f);
      g.C[row * g.ldc + col] = sum;
    }
  }
}

__global__ void __launch_bounds__(256) k_splitk_reduce(GemmArgs g) {
  const int64_t total = g.M * g.N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / g.N, col = t - row * g.N;
    float v = 0.f;
    for (int s = 0; s < g.nsplit; ++s) v += g.ws[((int64_t)s * g.M + row) * g.ws_ld + col];  // fixed order
    const bool oc = g.ones_col && col == g.N - 1;
    if (g.bias && !oc) v += g.bias2 ? g.bias[col] + g.bias2[col] : g.bias[col];
    if (g.relu) v = fmaxf(v, 0.f);
    if (oc) { if (g.db) g.db[row] = v; }
    else g.C[row * g.ldc + col] = v;
  }
}

static int g_gemm_mode = OGL_GEMM_F32;

extern "C" int ogl_set_gemm_mode(int mode) {
  if (mode == OGL_GEMM_QUERY) return g_gemm_mode;
  if (mode != OGL_GEMM_F32 && mode != OGL_GEMM_BF16X6 && mode != OGL_GEMM_AUTO) return OGL_EINVAL;
  g_gemm_mode = mode;
  return OGL_OK;
}

// 0: 128x128 (2x2 waves of 64x64)   1: 256x64 (narrow N)   2: 64x64 (few tiles: fill the chip / cut the tail)
static inline int gemm_config(int64_t M, int64_t N, int nsplit, int* BM, int* BN) {
  if (N <= 64) { *BM = 256; *BN = 64; return 1; }
  const int64_t big_tiles = ogl_cdiv(M, 128) * ogl_cdiv(N, 128);
  if (nsplit == 1 && big_tiles < 3 * 256) { *BM = 64; *BN = 64; return 2; }
  *BM = 128; *BN = 128; return 0;
}

// epilogue of an empty product (every part has a zero-length reduction): C = act(bias), db = 0
__global__ void __launch_bounds__(256) k_gemm_empty(GemmArgs g) {
  const int64_t total = g.M * g.N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / g.N, col = t - row * g.N;
    const bool oc = g.ones_col && col == g.N - 1;
    float v = (g.bias && !oc) ? (g.bias2 ? g.bias[col] + g.bias2[col] : g.bias[col]) : 0.f;
    if (g.relu) v = fmaxf(v, 0.f);
    if (oc) { if (g.db) g.db[row] = 0.f; }
    else g.C[row * g.ldc + col] = v;
  }
}

template <bool A_RC, bool B_RC>
static int launch_gemm(GemmArgs& g, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0) return OGL_OK;
  // The loaders read row 0 of an operand for masked-off lanes (branch-free), so an operand must have memory behind
  // it: parts with an empty reduction (torch passes nullptr for empty tensors) are dropped here.
  {
    int n = 0;
    for (int pi = 0; pi < g.nparts; ++pi)
      if (g.part[pi].R > 0 && g.part[pi].a.ptr && g.part[pi].b.ptr) g.part[n++] = g.part[pi];
    g.nparts = n;
    if (n == 0) {
      hipLaunchKernelGGL(k_gemm_empty, dim3((unsigned)min((int64_t)1024, ogl_cdiv(g.M * g.N, 256))), dim3(256), 0, stream, g);
      OGL_CHECK_LAUNCH();
      return OGL_OK;
    }
    if (n == 1) g.part[1] = GemmPart();
  }
  int BM, BN;
  int cfg = gemm_config(g.M, g.N, g.nsplit, &BM, &BN);
  if (g.force_cfg) {
    cfg = g.force_cfg - 1;
    BM = cfg == 1 ? 256 : (cfg == 2 ? 64 : 128);
    BN = cfg == 0 ? 128 : 64;
  }
  g.NI = (int)ogl_cdiv(g.M, BM);
  g.NJ = (int)ogl_cdiv(g.N, BN);
  dim3 grid((unsigned)((int64_t)g.NI * g.NJ * g.nsplit)), block(GEMM_THREADS);
  // AUTO: split-bf16 where it is measured faster (operands with at least one reduction-contiguous side: forward and
  // input-gradient GEMMs); the weight-gradient GEMM converts both operands on the fly and stays on the fp32 MFMA.
  const bool x6 = g_gemm_mode == OGL_GEMM_BF16X6 || (g_gemm_mode == OGL_GEMM_AUTO && (A_RC || B_RC));
  if (cfg == 1) {
    if (x6) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 4, 1, 2, 2, true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 4, 1, 2, 2, false>), grid, block, 0, stream, g);
  } else if (cfg == 2) {
    if (x6) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 2, 2, 1, 1, true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 2, 2, 1, 1, false>), grid, block, 0, stream, g);
  } else if (B_RC && g.ones_col) {
    if (x6) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 2, 2, 2, 2, true, B_RC>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 2, 2, 2, 2, false, B_RC>), grid, block, 0, stream, g);
  } else {
    if (x6) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 2, 2, 2, 2, true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((k_gemm<A_RC, B_RC, 2, 2, 2, 2, false>), grid, block, 0, stream, g);
  }
  OGL_CHECK_LAUNCH();
  if (g.nsplit > 1) {
    int64_t total = g.M * g.N;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)min((int64_t)2048, ogl_cdiv(total, 256))), dim3(256), 0,
                       stream, g);
    OGL_CHECK_LAUNCH();
  }
  return OGL_OK;
}

static inline void zero_args(GemmArgs& g) { g = GemmArgs(); g.nsplit = 1; }

// dy (.) [y > 0]: the backward of a fused-ReLU projection, applied once before its two backward GEMMs
__global__ void __launch_bounds__(256) k_relu_bwd(const float* __restrict__ dy, int64_t ldy, const float* __restrict__ y,
                                                  int64_t ldyy, int64_t M, int N, float* __restrict__ out, int64_t ldo) {
  const int64_t total = M * N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / N, c = t - r * N;
    out[r * ldo + c] = y[r * ldyy + c] > 0.f ? dy[r * ldy + c] : 0.f;
  }
}

extern "C" int ogl_relu_bwd(const float* dy, int64_t ldy, const float* y, int64_t ldyy, int64_t M, int N, float* out,
                            int64_t ldo, ogl_stream_t stream) {
  if (M < 0 || N < 0 || ldy < N || ldyy < N || ldo < N) return OGL_EINVAL;
  if (M == 0 || N == 0) return OGL_OK;
  if (!dy || !y || !out) return OGL_EINVAL;
  hipLaunchKernelGGL(k_relu_bwd, dim3((unsigned)min((int64_t)4096, ogl_cdiv(M * N, 256))), dim3(256), 0, (hipStream_t)stream, dy,
                     ldy, y, ldyy, M, N, out, ldo);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

static int linear_fwd_impl(const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows, int64_t M, int K, const float* w, int64_t ldw,
                           int N, const float* bias, const float* bias2, const float* x2, int64_t ldx2, const int64_t* x2_rows,
                           int64_t x2_nrows, int K2, const float* w2, int64_t ldw2, int relu, float* y, int64_t ldy, ogl_stream_t stream) {
  if (M < 0 || K < 0 || N < 0 || K2 < 0 || ldx < K || ldw < K || ldy < N || (bias2 && !bias)) return OGL_EINVAL;
  if (M == 0 || N == 0) return OGL_OK;
  if (!y || (K > 0 && (!x || !w)) || (K2 > 0 && (!x2 || !w2 || ldx2 < K2 || ldw2 < K2))) return OGL_EINVAL;
  GemmArgs g; zero_args(g);
  g.part[0].a = Operand{x, ldx, x_rows, x_nrows};
  g.part[0].b = Operand{w, ldw, nullptr, 0};
  g.part[0].R = K;
  g.nparts = 1;
  if (K2 > 0) {
    g.part[1].a = Operand{x2, ldx2, x2_rows, x2_nrows};
    g.part[1].b = Operand{w2, ldw2, nullptr, 0};
    g.part[1].R = K2;
    g.nparts = 2;
  }
  g.M = M; g.N = N; g.C = y; g.ldc = ldy; g.bias = bias; g.bias2 = bias2; g.relu = relu;
  // few output tiles and a long reduction: in-block split-K straight from global memory
  const int64_t Ktot = (int64_t)K + K2;
  // (every input at least one whole 8-deep chunk: a chunk past an input's end re-reads its chunk 0)
  // (M: up to a fused chunk of inference batches — a row's result must not depend on how many rows travel with it, so the kernel
  // choice for this shape class does not either: the sharded passes are bit-identical to the one-rank pass whatever the chunking)
  if (N <= 64 && M <= (1 << 17) && Ktot >= 256 && K >= 8 && (K2 == 0 || (g.nparts == 2 && K2 >= 8))) {
    dim3 grid((unsigned)ogl_cdiv(M, 32), (unsigned)ogl_cdiv(N, 32));
    if (g.nparts == 1) hipLaunchKernelGGL(k_gemm_skinny<1>, grid, dim3(64 * SK_WAVES), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(k_gemm_skinny<2>, grid, dim3(64 * SK_WAVES), 0, (hipStream_t)stream, g);
    OGL_CHECK_LAUNCH();
    return OGL_OK;
  }
  return launch_gemm<true, true>(g, (hipStream_t)stream);
}

extern "C" int ogl_linear_fwd(const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows, int64_t M, int K,
                              const float* w, int64_t ldw, int N, const float* bias,
                              const float* x2, int64_t ldx2, const int64_t* x2_rows, int64_t x2_nrows, int K2,
                              const float* w2, int64_t ldw2, int relu, float* y, int64_t ldy,
                              ogl_stream_t stream) {
  return linear_fwd_impl(x, ldx, x_rows, x_nrows, M, K, w, ldw, N, bias, nullptr, x2, ldx2, x2_rows, x2_nrows, K2, w2, ldw2, relu, y, ldy, stream);
}

// ogl_linear_fwd for a layer whose two projections each carry a bias (fc_self(h) + fc_neigh(neigh), both nn.Linear(bias=True)):
// y = act(x . w^T + x2 . w2^T + (bias + bias2)) — the sum of the two biases is formed first, as the separate launch it replaces did.
extern "C" int ogl_linear_fwd_dual_bias(const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows, int64_t M, int K,
                                        const float* w, int64_t ldw, int N, const float* bias, const float* bias2,
                                        const float* x2, int64_t ldx2, const int64_t* x2_rows, int64_t x2_nrows, int K2,
                                        const float* w2, int64_t ldw2, int relu, float* y, int64_t ldy, ogl_stream_t stream) {
  if (!bias || !bias2) return OGL_EINVAL;
  return linear_fwd_impl(x, ldx, x_rows, x_nrows, M, K, w, ldw, N, bias, bias2, x2, ldx2, x2_rows, x2_nrows, K2, w2, ldw2, relu, y, ldy, stream);
}

// y[M, N] = act(x[rows?] . w^T + bias + add[add_rows?[i], :]): ogl_linear_fwd with a per-row addend read from a table.
extern "C" int ogl_linear_fwd_addrows(const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows, int64_t M, int K,
                                      const float* w, int64_t ldw, int N, const float* bias, const float* add, int64_t ld_add,
                                      const int64_t* add_rows, int64_t add_nrows, int relu, float* y, int64_t ldy,
                                      ogl_stream_t stream) {
  if (M < 0 || K <= 0 || N < 0 || ldx < K || ldw < K || ldy < N || ld_add < N || add_nrows < 0) return OGL_EINVAL;
  if (M == 0 || N == 0) return OGL_OK;
  if (!y || !x || !w || !add) return OGL_EINVAL;
  GemmArgs g; zero_args(g);
  g.part[0].a = Operand{x, ldx, x_rows, x_nrows};
  g.part[0].b = Operand{w, ldw, nullptr, 0};
  g.part[0].R = K;
  g.nparts = 1;
  g.M = M; g.N = N; g.C = y; g.ldc = ldy; g.bias = bias; g.relu = relu;
  g.add = add; g.ld_add = ld_add; g.add_rows = add_rows; g.add_nrows = add_nrows;
  return launch_gemm<true, true>(g, (hipStream_t)stream);
}

extern "C" int ogl_linear_bwd_input(const float* dy, int64_t ldy, int64_t M,
                                    int N, const float* w, int64_t ldw, int K, float* dx, int64_t lddx,
                                    ogl_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || ldy < N || ldw < K || lddx < K) return OGL_EINVAL;
  if (M == 0 || K == 0) return OGL_OK;
  if (!dx || (N > 0 && (!dy || !w))) return OGL_EINVAL;
  GemmArgs g; zero_args(g);
  // dx[m,k] = sum_n dy[m,n] w[n,k] : A = dy (RC over n), B = w (NC: B(r=n, j=k) = w[n*ldw + k])
  g.part[0].a = Operand{dy, ldy, nullptr, 0};
  g.part[0].b = Operand{w, ldw, nullptr, 0};
  g.part[0].R = N;
  g.nparts = 1;
  g.M = M; g.N = K; g.C = dx; g.ldc = lddx;
  return launch_gemm<true, false>(g, (hipStream_t)stream);
}

// Split plan of the weight-gradient GEMM: [N, K+1] output, reduction over M rows.  Aim at one full round of
// resident blocks (3 per CU x 256 CUs) with at least 8 k-tiles each.  (64x64 tiles with fewer splits were
// measured slower for the n1-row reductions: 88 vs 75 us per call.)
static void bwd_weight_plan(int64_t M, int N, int K, int* nsplit, int* tps, int* cfg) {
  const int No = K + 1;  // + ones column
  const int64_t nk = ogl_cdiv(M, GEMM_BK);
  *cfg = No <= 64 ? 1 : (N <= 64 ? 2 : 0);   // few output rows (the [C, 2H] output layer): 64 x 64 tiles, 128-row tiles would be 2/3 padding
  if (nk == 0) { *nsplit = 1; *tps = 0; return; }
  int BM = No <= 64 ? 256 : (N <= 64 ? 64 : 128), BN = (No <= 64 || N <= 64) ? 64 : 128;
  int64_t tiles = ogl_cdiv(N, BM) * ogl_cdiv(No, BN);
  int64_t s = 768 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  if (s > nk) s = nk;
  // k-steps per block: >= 8 amortise a block's prologue / epilogue; the few-row output layer ([C, H] from 512 rows:
  // 10 tiles) is latency-bound instead — 16 splits of 2 steps measured 16.9 us against 20.9 us for 4 splits of 8
  const int min_steps = N <= 64 ? 2 : 8;
  if (nk / s < min_steps) s = nk / min_steps > 0 ? nk / min_steps : 1;
  *tps = (int)ogl_cdiv(nk, s);
  *nsplit = (int)ogl_cdiv(nk, *tps);
  if (*nsplit < 1) *nsplit = 1;
}

// ---- weight gradient of a FEW-COLUMN projection (the [H, C] output layer: N = C <= 64 output rows of dw) -------------
// [dw | db][n, k] = sum_m dy[m, n] * [x | 1][row(m), k] with N <= 64 and a short reduction (M = batch size).  The tiled
// kernel needs split-K plus a reduce launch for 10 output tiles (8.7 + 7 us at M = 512, K = 600, N = 41); here one block
// owns 8 columns k of [dw | db] for ALL n: thread (mg, kk) runs over the rows m = mg, mg + 64, ... with one x value and
// the whole dy row (float4s, N accumulators in registers) per row, plain fp32 FMAs; the 64 row groups are summed
// through LDS in a fixed order.  One launch (~10 us), no workspace, deterministic.
#define BWS_KT 8
#define BWS_MG 64
template <int NV>                                                // float4s per dy row held in registers: N <= 4 * NV
__global__ void __launch_bounds__(BWS_KT * BWS_MG) k_bwd_weight_skinny(const float* __restrict__ dy, int64_t ldy,
                                                                       const float* __restrict__ x, int64_t ldx,
                                                                       const int64_t* __restrict__ x_rows, int64_t x_nrows,
                                                                       int64_t M, int N, int K, float* __restrict__ dw,
                                                                       int64_t lddw, float* __restrict__ db) {
  __shared__ float red[BWS_MG][16][BWS_KT + 1];
  const int tid = threadIdx.x, kk = tid & (BWS_KT - 1), mg = tid / BWS_KT;
  const int64_t k = (int64_t)blockIdx.x * BWS_KT + kk;            // column of [dw | db]; k == K is the ones column
  const bool kx = k < K, kone = (k == K);
  const int nvr = (N + 3) / 4;                                    // float4s a dy row really has
  float4 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // rows m = mg, mg + 64, ...: two rows per trip, every load of the trip issued before the first FMA (a row past M
  // re-reads row mg and is weighted by 0)
  for (int64_t m0 = mg; m0 < M; m0 += 2 * BWS_MG) {
    float xv[2];
    float4 d4[2][NV];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t m = m0 + u * BWS_MG;
      const bool live = m < M;
      const int64_t mm = live ? m : mg;
      int64_t r = mm;
      bool ok = live;
      if (x_rows) { r = x_rows[mm]; ok = ok && r >= 0 && r < x_nrows; }
      const float xr = x[(ok ? r : 0) * ldx + (kx ? k : 0)];
      xv[u] = kone ? (live ? 1.f : 0.f) : ((kx && ok) ? xr : 0.f);
      const float4* dr = (const float4*)(dy + mm * ldy);
#pragma unroll
      for (int i = 0; i < NV; ++i) d4[u][i] = dr[i < nvr ? i : 0];   // float4s past the row's end are never read
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        acc[i].x += d4[u][i].x * xv[u]; acc[i].y += d4[u][i].y * xv[u];
        acc[i].z += d4[u][i].z * xv[u]; acc[i].w += d4[u][i].w * xv[u];
      }
  }
  // the row groups' partials, 16 output rows n at a time, summed in a fixed order
#pragma unroll
  for (int c4 = 0; c4 < NV; c4 += 4) {
    __syncthreads();                                             // the previous chunk's partials are consumed
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c4 + i < NV) {
        red[mg][4 * i + 0][kk] = acc[c4 + i].x; red[mg][4 * i + 1][kk] = acc[c4 + i].y;
        red[mg][4 * i + 2][kk] = acc[c4 + i].z; red[mg][4 * i + 3][kk] = acc[c4 + i].w;
      }
    __syncthreads();
    if (tid < 16 * BWS_KT) {                                      // thread = (n local, kk)
      const int i = tid / BWS_KT, c = tid & (BWS_KT - 1);
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < BWS_MG; ++g) sum += red[g][i][c];
      const int n = 4 * c4 + i;
      const int64_t kc = (int64_t)blockIdx.x * BWS_KT + c;
      if (n < N && n < 4 * NV) {
        if (kc < K) dw[n * lddw + kc] = sum;
        else if (kc == K && db) db[n] = sum;
      }
    }
  }
}

extern "C" int64_t ogl_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K) {
  if (M < 0 || N < 0 || K < 0) return OGL_EINVAL;
  int nsplit, tps, cfg;
  bwd_weight_plan(M, N, K, &nsplit, &tps, &cfg);
  if (nsplit <= 1) return 16;
  return (int64_t)nsplit * N * ogl_round_up(K + 1, 4) * 4;
}

extern "C" int ogl_linear_bwd_weight(const float* dy, int64_t ldy,
                                     const float* x, int64_t ldx, const int64_t* x_rows, int64_t x_nrows,
                                     int64_t M, int N, int K, float* dw, int64_t lddw, float* db,
                                     void* workspace, int64_t workspace_bytes, ogl_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || ldy < N || ldx < K || lddw < K) return OGL_EINVAL;
  if (N == 0) return OGL_OK;
  if (!dw || (M > 0 && (!dy || (K > 0 && !x)))) return OGL_EINVAL;
  // the output layer: one launch, no split-K (dy rows are read as float4s: 16-B aligned rows padded to a multiple of 4)
  if (N <= 64 && M > 0 && M <= 4096 && K >= 64 && x && (!x_rows || x_nrows > 0) && ldy % 4 == 0 && ldy >= (N + 3) / 4 * 4 && ((uintptr_t)dy & 15) == 0) {
    dim3 grid((unsigned)ogl_cdiv((int64_t)K + 1, BWS_KT)), block(BWS_KT * BWS_MG);
    const int nv = (N + 3) / 4;
#define OGL_BWS(NV_)                                                                                                   \
    hipLaunchKernelGGL(k_bwd_weight_skinny<NV_>, grid, block, 0, (hipStream_t)stream, dy, ldy, x, ldx, x_rows, x_nrows, M, N, K, \
                       dw, lddw, db)
    if (nv <= 4) OGL_BWS(4); else if (nv <= 8) OGL_BWS(8); else if (nv <= 12) OGL_BWS(12); else OGL_BWS(16);
#undef OGL_BWS
    OGL_CHECK_LAUNCH();
    return OGL_OK;
  }
  GemmArgs g; zero_args(g);
  // [dw | db][n, k] = sum_m dy[m,n] * [x | 1][m,k]: A = dy (NC: A(i=n, r=m) = dy[m*ldy + n]),
  // B = x (NC: B(r=m, j=k) = x[row(m)*ldx + k]) with a synthetic ones column at j = K.
  g.part[0].a = Operand{dy, ldy, nullptr, 0};
  g.part[0].b = Operand{x, ldx, x_rows, x_nrows};
  g.part[0].R = M;
  g.nparts = 1;
  g.M = N; g.N = K + 1; g.ones_col = 1; g.C = dw; g.ldc = lddw; g.db = db;
  bwd_weight_plan(M, N, K, &g.nsplit, &g.tiles_per_split, &g.force_cfg);
  g.force_cfg += 1;   // 0 = let launch_gemm choose
  if (M == 0 || K == 0) { g.nsplit = 1; g.tiles_per_split = 0; }
  if (g.nsplit > 1) {
    g.ws_ld = ogl_round_up(K + 1, 4);
    if (!workspace || workspace_bytes < (int64_t)g.nsplit * N * g.ws_ld * 4) return OGL_EWORKSPACE;
    g.ws = (float*)workspace;
  }
  return launch_gemm<false, false>(g, (hipStream_t)stream);
}


// ---- weight gradient from TRANSPOSED operands ------------------------------------------------------------------------
// dw[n,k] = sum_m dyT[n,m] * xT[k,m] is a reduction-contiguous x reduction-contiguous product: it runs on the same
// fast path as the forward GEMM (and on the split-bf16 arithmetic in BF16X6 / AUTO mode) instead of the k-major
// (NC x NC) path.  db = row sums of dyT through a synthetic all-ones row of the second operand.
extern "C" int64_t ogl_linear_bwd_weight_t_workspace_bytes(int64_t M, int N, int K) {
  return ogl_linear_bwd_weight_workspace_bytes(M, N, K);
}

extern "C" int ogl_linear_bwd_weight_t(const float* dyT, int64_t lddyT, const float* xT, int64_t ldxT, int64_t M, int N,
                                       int K, float* dw, int64_t lddw, float* db, void* workspace,
                                       int64_t workspace_bytes, ogl_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || lddyT < M || ldxT < M || lddw < K) return OGL_EINVAL;
  if (N == 0) return OGL_OK;
  if (!dw || (M > 0 && (!dyT || (K > 0 && !xT)))) return OGL_EINVAL;
  GemmArgs g; zero_args(g);
  g.part[0].a = Operand{dyT, lddyT, nullptr, 0};
  g.part[0].b = Operand{xT, ldxT, nullptr, 0};
  g.part[0].R = M;
  g.nparts = 1;
  g.M = N; g.N = K + 1; g.ones_col = 1; g.C = dw; g.ldc = lddw; g.db = db;
  bwd_weight_plan(M, N, K, &g.nsplit, &g.tiles_per_split, &g.force_cfg);
  g.force_cfg = 1;      // 128 x 128 tiles: the only configuration built with the ones-row loader
  if (M == 0 || K == 0) { g.nsplit = 1; g.tiles_per_split = 0; }
  if (g.nsplit > 1) {
    g.ws_ld = ogl_round_up(K + 1, 4);
    if (!workspace || workspace_bytes < (int64_t)g.nsplit * N * g.ws_ld * 4) return OGL_EWORKSPACE;
    g.ws = (float*)workspace;
  }
  return launch_gemm<true, true>(g, (hipStream_t)stream);
}

// dst[j, i] = src[row(i), j]: 64 x 64 tiles through LDS (both sides 16-B vectorised and coalesced), optional row gather
__global__ void __launch_bounds__(256) k_transpose(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ rows,
                                                   int64_t nrows, int64_t M, int N, float* __restrict__ dst, int64_t ldt) {
  __shared__ float tile[64][65];
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int tid = threadIdx.x, ty = tid >> 4, tx = (tid & 15) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = i0 + ty + 16 * k;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < M) {
      int64_t row = rows ? rows[i] : i;
      if (!rows || (row >= 0 && row < nrows)) {
        const float* p = src + row * ld + j0 + tx;
        if (j0 + tx + 3 < N) v = ld16(p);
        else {
          if (j0 + tx < N) v.x = p[0];
          if (j0 + tx + 1 < N) v.y = p[1];
          if (j0 + tx + 2 < N) v.z = p[2];
        }
      }
    }
    tile[ty + 16 * k][tx] = v.x; tile[ty + 16 * k][tx + 1] = v.y; tile[ty + 16 * k][tx + 2] = v.z; tile[ty + 16 * k][tx + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = j0 + ty + 16 * k;
    if (j >= N) continue;
    const int64_t i = i0 + tx;
    float* o = dst + (int64_t)j * ldt + i;
    const float a = tile[tx][ty + 16 * k], b = tile[tx + 1][ty + 16 * k], c = tile[tx + 2][ty + 16 * k], d = tile[tx + 3][ty + 16 * k];
    if (i + 3 < M && (ldt & 3) == 0 && (((uintptr_t)dst & 15) == 0)) *(float4*)o = make_float4(a, b, c, d);
    else {
      if (i < M) o[0] = a;
      if (i + 1 < M) o[1] = b;
      if (i + 2 < M) o[2] = c;
      if (i + 3 < M) o[3] = d;
    }
  }
}

extern "C" int ogl_transpose(const float* src, int64_t ld, const int64_t* rows, int64_t nrows, int64_t M, int N, float* dst,
                             int64_t ldt, ogl_stream_t stream) {
  if (M < 0 || N < 0 || ld < N || ldt < M) return OGL_EINVAL;
  if (M == 0 || N == 0) return OGL_OK;
  if (!src || !dst) return OGL_EINVAL;
  dim3 grid((unsigned)ogl_cdiv(M, 64), (unsigned)ogl_cdiv(N, 64));
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, (hipStream_t)stream, src, ld, rows, nrows, M, N, dst, ldt);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
