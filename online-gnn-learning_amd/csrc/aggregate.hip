// Row gather + fixed-fanout neighbour reduction (the aggregator) and its backward.
// Replaces: graph.ndata['feat'][input_nodes] (R/train/graphsage/pytorch/model.py:54,88,182,232) and the
// message-passing reduce of SAGEConv — DGL copy_src->max on the live 'pool' layer, mailbox
// .mean/.sum(axis=1) in R/train/graphsage/pytorch/aggregator_dgl.py:158,165,175,185.
//
// HBM-bound.  One 64-lane wavefront per destination; lanes stride the feature row in 16-B pieces
// (a 602-float row padded to ld=608 is 152 float4 = 2.375 wave-instructions), four neighbour rows
// in flight per wave, row bases wave-uniform (v_readlane -> scalar address), max/sum kept in
// registers, one coalesced store per destination.  Algorithmic bytes per launch:
// E*(4*D + idx) + n_dst*4*D  (SURVEY.md §8d).
#include <type_traits>
#include "ogl_common.h"
#include "x6_arith.h"

#define WAVES_PER_BLOCK 4

__device__ __forceinline__ int64_t bcast_idx(int32_t v, int j) { return (int64_t)__builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ int64_t bcast_idx(int64_t v, int j) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v & 0xFFFFFFFF), j);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), j);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ---- vectorised forward: NCH float4 chunks per lane (d <= 256*NCH floats) -------------------
// One wave per (destination, column slice): `parts` waves share a destination, each reducing a contiguous slice of
// ceil(d4 / parts) float4 columns.  A batch of 7 000 destinations is only 1.4 rounds of one-wave-per-destination on the
// chip (256 CUs x 20 waves): two slices per destination make it 2 rounds of smaller waves — no half-empty tail round.
// IMG: ALSO write the bf16x3 image of the output (row-major, n_dst + 1 rows, reduction length d; x6_arith.h) — the A operand of the
// projection that consumes the reduced rows, without a split pass of its own: a lane's float4 is half a 16-byte piece of each
// plane (3 x 8-byte stores beside the 16-byte fp32 store).
template <int OP, typename IdxT, bool ARG, int NCH, bool IMG = false, int U = (NCH == 1 ? 8 : 4)>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK)
k_reduce_fwd_v4(const float* __restrict__ src, int64_t lds, int64_t n_src, const IdxT* __restrict__ idx,
                int64_t n_dst, int S, int d, float* __restrict__ out, int64_t ldo,
                int32_t* __restrict__ argmax, int parts, unsigned char* __restrict__ img = nullptr, int64_t img_row_bytes = 0,
                const int64_t* __restrict__ rows = nullptr, int64_t n_rows = 0) {
  // rows (optional): idx holds positions into `rows`, the reduced row is src[rows[idx]] — a block's local indices over the
  // resident table through the block's source ids (one more dependent load per WAVE, not per neighbour row)
  const int lane = threadIdx.x & 63;
  const int64_t wg = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int64_t w = wg / parts;
  const bool live = w < n_dst;
  if (!live) return;
  const int part = (int)(wg - w * parts);
  const int dall4 = (d + 3) >> 2, cper = (dall4 + parts - 1) / parts;
  const int c0 = part * cper;
  const int d4 = min(dall4, c0 + cper);            // this wave's slice: float4 columns [c0, d4)
  float4 acc[NCH];
  int arg[NCH][4];
  bool any = false;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    arg[c][0] = arg[c][1] = arg[c][2] = arg[c][3] = -1;
  }
  for (int s0 = 0; s0 < S && live; s0 += 64) {
    const int sc = min(64, S - s0);
    IdxT mine = lane < sc ? idx[w * S + s0 + lane] : (IdxT)-1;
    if (rows) {
      const int64_t m = (int64_t)mine;
      mine = (m >= 0 && m < n_rows) ? (IdxT)rows[m] : (IdxT)-1;
    }
    for (int j0 = 0; j0 < sc; j0 += U) {
      // The U x NCH row loads are issued back to back with nothing between them that needs a wait: all row ids are
      // broadcast first (scalars), a missing row (past S, id -1, id >= n_src) reads row 0 and is skipped below, a lane
      // past the slice reads the slice's last float4.  (A branch around each row's loads makes the compiler drain
      // vmcnt at every join, i.e. one row in flight per wave — measured 0.086 ms against 0.060 ms for layer 0.)
      float4 v[U][NCH];
      int64_t r[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u;
        const int64_t q = bcast_idx(mine, j < sc ? j : sc - 1);
        ok[u] = (j < sc) && q >= 0 && q < n_src;
        r[u] = ok[u] ? q : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4* rp = (const float4*)(src + r[u] * lds);
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[u][c] = rp[min(c0 + c * 64 + lane, d4 - 1)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;  // wave-uniform
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (OP == OGL_REDUCE_MAX) {
            if (!any) {
              acc[c] = v[u][c];
              if (ARG) arg[c][0] = arg[c][1] = arg[c][2] = arg[c][3] = (int)r[u];
             
            } else {
              if (v[u][c].x > acc[c].x) { acc[c].x = v[u][c].x; if (ARG) arg[c][0] = (int)r[u]; }
              if (v[u][c].y > acc[c].y) { acc[c].y = v[u][c].y; if (ARG) arg[c][1] = (int)r[u]; }
              if (v[u][c].z > acc[c].z) { acc[c].z = v[u][c].z; if (ARG) arg[c][2] = (int)r[u]; }
              if (v[u][c].w > acc[c].w) { acc[c].w = v[u][c].w; if (ARG) arg[c][3] = (int)r[u]; }
            }
          } else {
            acc[c].x += v[u][c].x; acc[c].y += v[u][c].y; acc[c].z += v[u][c].z; acc[c].w += v[u][c].w;
          }
        }
        any = true;
      }
    }
  }
  const float fS = (float)S;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int ch = c0 + c * 64 + lane;
    if (ch < d4) {
      float4 o = acc[c];
      if (OP == OGL_REDUCE_MEAN && any) { o.x /= fS; o.y /= fS; o.z /= fS; o.w /= fS; }
      if (!IMG || out) ((float4*)(out + w * ldo))[ch] = o;     // (IMG: `out` may be null — a consumer that reads the image only)
      if (IMG) {
        const int b4 = ch * 4;
        const float e0 = b4 < d ? o.x : 0.f, e1 = b4 + 1 < d ? o.y : 0.f, e2 = b4 + 2 < d ? o.z : 0.f, e3 = b4 + 3 < d ? o.w : 0.f;
        unsigned h0, m0, l0, h1, m1, l1;
        split3(e0, e1, h0, m0, l0); split3(e2, e3, h1, m1, l1);
        unsigned char* rp = img + w * img_row_bytes + (int64_t)(ch >> 3) * 192 + (ch & 1) * 8;
        const int pc = (ch & 7) >> 1;
        *(uint2*)(rp + x3_piece(pc, 0) * 16) = make_uint2(h0, h1);
        *(uint2*)(rp + x3_piece(pc, 1) * 16) = make_uint2(m0, m1);
        *(uint2*)(rp + x3_piece(pc, 2) * 16) = make_uint2(l0, l1);
        if (w == n_dst - 1) {                               // the image's all-zero row sits behind the last destination
          rp += img_row_bytes;
          *(uint2*)(rp + x3_piece(pc, 0) * 16) = make_uint2(0u, 0u);
          *(uint2*)(rp + x3_piece(pc, 1) * 16) = make_uint2(0u, 0u);
          *(uint2*)(rp + x3_piece(pc, 2) * 16) = make_uint2(0u, 0u);
        }
      }
      if (ARG) {
        int base = ch * 4;
        int32_t* ap = argmax + w * (int64_t)d + base;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (base + e < d) ap[e] = arg[c][e];
      }
    }
  }
  if (IMG && part == parts - 1) {
    const int kpad4 = (int)(img_row_bytes / 192) * 8;       // float4 columns of the padded image row
    for (int ch = dall4 + lane; ch < kpad4; ch += 64) {
      for (int rr = 0; rr < (w == n_dst - 1 ? 2 : 1); ++rr) {
        unsigned char* rp = img + (w + rr) * img_row_bytes + (int64_t)(ch >> 3) * 192 + (ch & 1) * 8;
        const int pc = (ch & 7) >> 1;
        *(uint2*)(rp + x3_piece(pc, 0) * 16) = make_uint2(0u, 0u);
        *(uint2*)(rp + x3_piece(pc, 1) * 16) = make_uint2(0u, 0u);
        *(uint2*)(rp + x3_piece(pc, 2) * 16) = make_uint2(0u, 0u);
      }
    }
  }
}

// ---- narrow rows (d <= 128 floats, max without argmax: the inference passes over a cached projection table of <= 512-byte rows) ----
// A 512-byte row fills HALF a wave's 16-byte-per-lane load: with one destination per wave, lanes 32-63 re-read the row's last float4.
// Here the two halves of the wave walk the destination's EVEN and ODD sampling slots: lane l reads float4 column l & 31 of the row of
// slot 2 u + (l >> 5), 2 U rows in flight per wave-instruction pair instead of U, and the halves meet in one shuffle at the end.  max
// is exact and order-free (no argmax here: the first-winner rule of the training path needs the slot order), so the result has the bits
// of k_reduce_fwd_v4.  The arxiv-like priority forward (D = 128, 87 MB table in the Infinity Cache): see DESIGN.md section 4.
template <typename IdxT, bool IMG, int U = 8>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK)
k_reduce_fwd_max_half(const float* __restrict__ src, int64_t lds, int64_t n_src, const IdxT* __restrict__ idx, int64_t n_dst, int S, int d,
                      float* __restrict__ out, int64_t ldo, unsigned char* __restrict__ img, int64_t img_row_bytes,
                      const int64_t* __restrict__ rows, int64_t n_rows) {
  const int lane = threadIdx.x & 63, hl = lane & 31, hh = lane >> 5;
  const int64_t w = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (w >= n_dst) return;
  const int d4 = (d + 3) >> 2;                       // <= 32
  const int col = min(hl, d4 - 1);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  bool any = false;                                  // (per half)
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int sc = min(64, S - s0);
    IdxT mine = lane < sc ? idx[w * S + s0 + lane] : (IdxT)-1;
    if (rows) {
      const int64_t m = (int64_t)mine;
      mine = (m >= 0 && m < n_rows) ? (IdxT)rows[m] : (IdxT)-1;
    }
    for (int j0 = 0; j0 < sc; j0 += 2 * U) {
      // (as in k_reduce_fwd_v4: every row id first, then every load, nothing between them that needs a wait; a missing row reads
      // row 0 and is skipped)
      float4 v[U];
      bool ok[U];
      const float4* rp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int je = j0 + 2 * u, jo = je + 1;
        const int64_t qe = bcast_idx(mine, je < sc ? je : sc - 1), qo = bcast_idx(mine, jo < sc ? jo : sc - 1);
        const int64_t q = hh ? qo : qe;
        ok[u] = ((hh ? jo : je) < sc) && q >= 0 && q < n_src;
        rp[u] = (const float4*)(src + (ok[u] ? q : 0) * lds);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = rp[u][col];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;                        // (uniform over a half)
        if (!any) acc = v[u];
        else {
          acc.x = v[u].x > acc.x ? v[u].x : acc.x; acc.y = v[u].y > acc.y ? v[u].y : acc.y;
          acc.z = v[u].z > acc.z ? v[u].z : acc.z; acc.w = v[u].w > acc.w ? v[u].w : acc.w;
        }
        any = true;
      }
    }
  }
  // the halves meet: the lower half (even slots, slot 0 among them) keeps its value on ties, as the sequential walk does
  float4 o;
  o.x = __shfl_xor(acc.x, 32); o.y = __shfl_xor(acc.y, 32); o.z = __shfl_xor(acc.z, 32); o.w = __shfl_xor(acc.w, 32);
  const bool any_o = __shfl_xor((int)any, 32) != 0;
  if (any && any_o) {
    acc.x = o.x > acc.x ? o.x : acc.x; acc.y = o.y > acc.y ? o.y : acc.y; acc.z = o.z > acc.z ? o.z : acc.z; acc.w = o.w > acc.w ? o.w : acc.w;
  } else if (any_o) acc = o;
  if (hh == 0 && hl < d4) {
    const int ch = hl;
    if (!IMG || out) ((float4*)(out + w * ldo))[ch] = acc;
    if (IMG) {
      const int b4 = ch * 4;
      const float e0 = b4 < d ? acc.x : 0.f, e1 = b4 + 1 < d ? acc.y : 0.f, e2 = b4 + 2 < d ? acc.z : 0.f, e3 = b4 + 3 < d ? acc.w : 0.f;
      unsigned h0, m0, l0, h1, m1, l1;
      split3(e0, e1, h0, m0, l0); split3(e2, e3, h1, m1, l1);
      unsigned char* rq = img + w * img_row_bytes + (int64_t)(ch >> 3) * 192 + (ch & 1) * 8;
      const int pc = (ch & 7) >> 1;
      *(uint2*)(rq + x3_piece(pc, 0) * 16) = make_uint2(h0, h1);
      *(uint2*)(rq + x3_piece(pc, 1) * 16) = make_uint2(m0, m1);
      *(uint2*)(rq + x3_piece(pc, 2) * 16) = make_uint2(l0, l1);
      if (w == n_dst - 1) {                               // the image's all-zero row sits behind the last destination
        rq += img_row_bytes;
        *(uint2*)(rq + x3_piece(pc, 0) * 16) = make_uint2(0u, 0u);
        *(uint2*)(rq + x3_piece(pc, 1) * 16) = make_uint2(0u, 0u);
        *(uint2*)(rq + x3_piece(pc, 2) * 16) = make_uint2(0u, 0u);
      }
    }
  }
  if (IMG) {
    const int kpad4 = (int)(img_row_bytes / 192) * 8;       // float4 columns of the padded image row
    for (int ch = d4 + lane; ch < kpad4; ch += 64) {
      for (int rr = 0; rr < (w == n_dst - 1 ? 2 : 1); ++rr) {
        unsigned char* rq = img + (w + rr) * img_row_bytes + (int64_t)(ch >> 3) * 192 + (ch & 1) * 8;
        const int pc = (ch & 7) >> 1;
        *(uint2*)(rq + x3_piece(pc, 0) * 16) = make_uint2(0u, 0u);
        *(uint2*)(rq + x3_piece(pc, 1) * 16) = make_uint2(0u, 0u);
        *(uint2*)(rq + x3_piece(pc, 2) * 16) = make_uint2(0u, 0u);
      }
    }
  }
}

// ---- generic forward (any d / alignment): one wave per destination, dword accesses ------------
template <int OP, typename IdxT, bool ARG>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK)
k_reduce_fwd_generic(const float* __restrict__ src, int64_t lds, int64_t n_src, const IdxT* __restrict__ idx,
                     int64_t n_dst, int S, int d, float* __restrict__ out, int64_t ldo,
                     int32_t* __restrict__ argmax) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (w >= n_dst) return;
  for (int c = lane; c < d; c += 64) {
    float acc = 0.f;
    int arg = -1;
    bool any = false;
    for (int j = 0; j < S; ++j) {
      int64_t r = (int64_t)idx[w * S + j];
      if (r < 0 || r >= n_src) continue;
      float v = src[r * lds + c];
      if (OP == OGL_REDUCE_MAX) {
        if (!any || v > acc) { acc = v; arg = (int)r; }
      } else {
        acc += v;
      }
      any = true;
    }
    if (OP == OGL_REDUCE_MEAN && any) acc /= (float)S;
    out[w * ldo + c] = acc;
    if (ARG) argmax[w * (int64_t)d + c] = arg;
  }
}

static int g_reduce_half = 1;      // (ogl_debug_set(OGL_KNOB_REDUCE_HALF): tests and A/B runs pin the one-row-per-wave form)
int oglx_knob_reduce_half(int on, int* prev) { *prev = g_reduce_half; g_reduce_half = on ? 1 : 0; return OGL_OK; }

template <int OP, typename IdxT, bool ARG>
static int launch_reduce_fwd(const float* src, int64_t lds, int64_t n_src, const IdxT* idx, int64_t n_dst,
                             int S, int d, float* out, int64_t ldo, int32_t* argmax, hipStream_t stream,
                             unsigned char* img = nullptr, const int64_t* rows = nullptr, int64_t n_rows = 0) {
  dim3 grid((unsigned)ogl_cdiv(n_dst, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
  const int d4 = (d + 3) / 4;
  const bool vec = (lds % 4 == 0) && (ldo % 4 == 0) && (lds >= 4 * d4) && (ldo >= 4 * d4) &&
                   (((uintptr_t)src & 15) == 0) && (((uintptr_t)out & 15) == 0) && d4 <= 256 && n_src > 0;
  // column slices per destination: two when one wave per destination would not even fill two rounds of the chip
  // column slices per destination: the fewest (at most 4, at least 32 float4 columns each) that put two full rounds
  // of waves on the chip.  Measured (MI355X): 7 054 x 604 floats, 25 rows each: 1 slice 0.0607 ms, 2 slices 0.0595 ms,
  // 3 slices 0.0586 ms; 512 x 600: 1 slice 0.0168 ms, 2 slices 0.0148 ms, 4 slices 0.0137 ms.
  int parts = 1;
  while (vec && parts < 4 && n_dst * parts < 2 * 256 * 20 && d4 / (parts + 1) >= 32) ++parts;
  const int cper = (d4 + parts - 1) / parts;
  if (vec) grid.x = (unsigned)ogl_cdiv(n_dst * parts, WAVES_PER_BLOCK);
  // rows of <= 512 bytes, max without argmax (the inference passes): two rows per wave-instruction (k_reduce_fwd_max_half)
  const bool half = vec && OP == OGL_REDUCE_MAX && !ARG && d4 <= 32 && S >= 2 && g_reduce_half;
  if (half) {
    grid.x = (unsigned)ogl_cdiv(n_dst, WAVES_PER_BLOCK);
    const int64_t irb = ogl_cdiv(d, 32) * 192;
    if (img) hipLaunchKernelGGL((k_reduce_fwd_max_half<IdxT, true>), grid, block, 0, stream, src, lds, n_src, idx, n_dst, S, d, out, ldo, img, irb, rows, n_rows);
    else if (rows) return OGL_EINVAL;
    else hipLaunchKernelGGL((k_reduce_fwd_max_half<IdxT, false>), grid, block, 0, stream, src, lds, n_src, idx, n_dst, S, d, out, ldo, (unsigned char*)nullptr, (int64_t)0, (const int64_t*)nullptr, (int64_t)0);
    OGL_CHECK_LAUNCH();
    return OGL_OK;
  }
  if (img) {
    if (!vec) return OGL_EINVAL;     // the image form exists for the vectorised kernels (the 'pool' / 'meanpool' layers)
    const int64_t irb = ogl_cdiv(d, 32) * 192;
    if (cper <= 64) hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 1, true>), grid, block, 0, stream, src, lds, n_src, idx, n_dst, S, d, out, ldo, argmax, parts, img, irb, rows, n_rows);
    else if (cper <= 128) hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 2, true>), grid, block, 0, stream, src, lds, n_src, idx, n_dst, S, d, out, ldo, argmax, parts, img, irb, rows, n_rows);
    else if (cper <= 192) hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 3, true>), grid, block, 0, stream, src, lds, n_src, idx, n_dst, S, d, out, ldo, argmax, parts, img, irb, rows, n_rows);
    else hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 4, true>), grid, block, 0, stream, src, lds, n_src, idx, n_dst, S, d, out, ldo, argmax, parts, img, irb, rows, n_rows);
    OGL_CHECK_LAUNCH();
    return OGL_OK;
  }
  if (rows) return OGL_EINVAL;       // (the indirection exists in the image form)
  if (!vec) {
    hipLaunchKernelGGL((k_reduce_fwd_generic<OP, IdxT, ARG>), grid, block, 0, stream, src, lds, n_src, idx,
                       n_dst, S, d, out, ldo, argmax);
  } else if (cper <= 64) {
    hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 1>), grid, block, 0, stream, src, lds, n_src, idx,
                       n_dst, S, d, out, ldo, argmax, parts);
  } else if (cper <= 128) {
    hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 2>), grid, block, 0, stream, src, lds, n_src, idx,
                       n_dst, S, d, out, ldo, argmax, parts);
  } else if (cper <= 192) {
    hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 3>), grid, block, 0, stream, src, lds, n_src, idx,
                       n_dst, S, d, out, ldo, argmax, parts);
  } else {
    hipLaunchKernelGGL((k_reduce_fwd_v4<OP, IdxT, ARG, 4>), grid, block, 0, stream, src, lds, n_src, idx,
                       n_dst, S, d, out, ldo, argmax, parts);
  }
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

template <typename IdxT>
static int dispatch_reduce_fwd(const float* src, int64_t lds, int64_t n_src, const IdxT* idx, int64_t n_dst,
                               int S, int d, int op, float* out, int64_t ldo, int32_t* argmax,
                               hipStream_t stream) {
  switch (op) {
    case OGL_REDUCE_MAX:
      return argmax ? launch_reduce_fwd<OGL_REDUCE_MAX, IdxT, true>(src, lds, n_src, idx, n_dst, S, d, out, ldo, argmax, stream)
                    : launch_reduce_fwd<OGL_REDUCE_MAX, IdxT, false>(src, lds, n_src, idx, n_dst, S, d, out, ldo, nullptr, stream);
    case OGL_REDUCE_MEAN:
      return launch_reduce_fwd<OGL_REDUCE_MEAN, IdxT, false>(src, lds, n_src, idx, n_dst, S, d, out, ldo, nullptr, stream);
    case OGL_REDUCE_SUM:
      return launch_reduce_fwd<OGL_REDUCE_SUM, IdxT, false>(src, lds, n_src, idx, n_dst, S, d, out, ldo, nullptr, stream);
    default:
      return OGL_EINVAL;
  }
}

extern "C" int ogl_reduce_fwd(const float* src, int64_t lds, int64_t n_src, const int32_t* idx32,
                              const int64_t* idx64, int64_t n_dst, int fanout, int d, int op, float* out,
                              int64_t ldo, int32_t* argmax, ogl_stream_t stream) {
  if (n_dst < 0 || fanout < 0 || d < 0 || n_src < 0 || lds < d || ldo < d) return OGL_EINVAL;
  if ((idx32 != nullptr) == (idx64 != nullptr) && fanout > 0 && n_dst > 0) return OGL_EINVAL;
  if (n_dst == 0 || d == 0) return OGL_OK;
  if (!out || (!src && n_src > 0)) return OGL_EINVAL;
  if (idx32) return dispatch_reduce_fwd<int32_t>(src, lds, n_src, idx32, n_dst, fanout, d, op, out, ldo, argmax, (hipStream_t)stream);
  return dispatch_reduce_fwd<int64_t>(src, lds, n_src, idx64, n_dst, fanout, d, op, out, ldo, argmax, (hipStream_t)stream);
}

// ogl_reduce_fwd(OGL_REDUCE_MAX) that ALSO writes the bf16x3 image of `out` (what ogl_x3_split(out) would build; n_dst + 1 image
// rows over a reduction of d): the A operand of the projection that consumes the pooled rows.  16-byte-aligned, 4-float-strided
// operands only (what this package allocates).
extern "C" int ogl_reduce_fwd_img(const float* src, int64_t lds, int64_t n_src, const int32_t* idx32, const int64_t* idx64,
                                  int64_t n_dst, int fanout, int d, float* out, int64_t ldo, int32_t* argmax, void* image,
                                  ogl_stream_t stream) {
  if (n_dst <= 0 || fanout <= 0 || d <= 0 || n_src <= 0 || lds < d || (out && ldo < d)) return OGL_EINVAL;
  if ((idx32 != nullptr) == (idx64 != nullptr)) return OGL_EINVAL;
  if (!src || !image || ((uintptr_t)image & 15)) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* im = (unsigned char*)image;
  if (idx32) return argmax ? launch_reduce_fwd<OGL_REDUCE_MAX, int32_t, true>(src, lds, n_src, idx32, n_dst, fanout, d, out, ldo, argmax, st, im)
                           : launch_reduce_fwd<OGL_REDUCE_MAX, int32_t, false>(src, lds, n_src, idx32, n_dst, fanout, d, out, ldo, nullptr, st, im);
  return argmax ? launch_reduce_fwd<OGL_REDUCE_MAX, int64_t, true>(src, lds, n_src, idx64, n_dst, fanout, d, out, ldo, argmax, st, im)
                : launch_reduce_fwd<OGL_REDUCE_MAX, int64_t, false>(src, lds, n_src, idx64, n_dst, fanout, d, out, ldo, nullptr, st, im);
}

// The same for the MEAN (the 'meanpool' / 'mean' layers of the in-repo SAGEConv, R/train/graphsage/pytorch/aggregator_dgl.py:156-159,
// 181-185): out = mailbox.mean(axis=1) in slot order + the bf16x3 image of it for the combine product that consumes it.
extern "C" int ogl_reduce_fwd_mean_img(const float* src, int64_t lds, int64_t n_src, const int32_t* idx32, const int64_t* idx64,
                                       int64_t n_dst, int fanout, int d, float* out, int64_t ldo, void* image, ogl_stream_t stream) {
  if (n_dst <= 0 || fanout <= 0 || d <= 0 || n_src <= 0 || lds < d || (out && ldo < d)) return OGL_EINVAL;
  if ((idx32 != nullptr) == (idx64 != nullptr)) return OGL_EINVAL;
  if (!src || !image || ((uintptr_t)image & 15)) return OGL_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* im = (unsigned char*)image;
  if (idx32) return launch_reduce_fwd<OGL_REDUCE_MEAN, int32_t, false>(src, lds, n_src, idx32, n_dst, fanout, d, out, ldo, nullptr, st, im);
  return launch_reduce_fwd<OGL_REDUCE_MEAN, int64_t, false>(src, lds, n_src, idx64, n_dst, fanout, d, out, ldo, nullptr, st, im);
}

// The MEAN over rows of a resident TABLE through a block's local indices: out[d] = mean_j table[rows[idx32[d, j]]] + its image — the
// aggregator of the in-repo 'mean' mode's first layer (aggregator_dgl.py:156-159) without materialising feat[input_nodes]
// (R/train/graphsage/pytorch/model.py:88): the gathered copy (n0 x 2.4 KB written and read back) never exists.  Table ids < 2^31.
extern "C" int ogl_reduce_fwd_rows_mean_img(const float* table, int64_t ldt, int64_t n_table, const int32_t* idx32, const int64_t* rows,
                                            int64_t n_rows, int64_t n_dst, int fanout, int d, float* out, int64_t ldo, void* image,
                                            ogl_stream_t stream) {
  if (n_dst <= 0 || fanout <= 0 || d <= 0 || n_table <= 0 || n_table >= (1ll << 31) || n_rows < 0 || ldt < d || (out && ldo < d)) return OGL_EINVAL;
  if (!table || !idx32 || !rows || !image || ((uintptr_t)image & 15)) return OGL_EINVAL;
  return launch_reduce_fwd<OGL_REDUCE_MEAN, int32_t, false>(table, ldt, n_table, idx32, n_dst, fanout, d, out, ldo, nullptr, (hipStream_t)stream,
                                                            (unsigned char*)image, rows, n_rows);
}

// ---- backward ----------------------------------------------------------------------------------
// max: only the winning source row receives the gradient (n_dst*d float atomics).
// relu_out (nullable) = the forward max output: when the reduced rows were ReLU outputs, out[d,c] is the
// winner's value, so (out > 0) IS the winner's ReLU mask and the projection's backward needs no mask pass.
__global__ void __launch_bounds__(256) k_reduce_bwd_max(const float* __restrict__ dout, int64_t ldo,
                                                        const int32_t* __restrict__ argmax,
                                                        const float* __restrict__ relu_out, int64_t ldr,
                                                        int64_t n_dst, int d, int64_t n_src,
                                                        float* __restrict__ dsrc, int64_t lds) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_dst) return;
  for (int c = lane; c < d; c += 64) {
    int a = argmax[w * (int64_t)d + c];
    if (a < 0 || a >= n_src) continue;
    if (relu_out && !(relu_out[w * ldr + c] > 0.f)) continue;
    atomicAdd(&dsrc[(int64_t)a * lds + c], dout[w * ldo + c]);
  }
}

// mean/sum: every sampled slot receives dout (scaled by 1/S for mean); one 256-B contiguous
// atomic wave-instruction per (slot, 64 columns) — the shape the float-atomic path runs at full rate.
__global__ void __launch_bounds__(256) k_reduce_bwd_sum(const float* __restrict__ dout, int64_t ldo,
                                                        const int32_t* __restrict__ idx, int64_t n_dst, int S, int d,
                                                        float scale, int64_t n_src, float* __restrict__ dsrc,
                                                        int64_t lds) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_dst) return;
  for (int c0 = 0; c0 < d; c0 += 64) {
    int c = c0 + lane;
    float g = c < d ? dout[w * ldo + c] * scale : 0.f;
    for (int j = 0; j < S; ++j) {
      int r = idx[w * S + j];
      if (r < 0 || r >= n_src) continue;
      if (c < d) atomicAdd(&dsrc[(int64_t)r * lds + c], g);
    }
  }
}

extern "C" int ogl_reduce_bwd(const float* dout, int64_t ldo, const int32_t* idx32, const int32_t* argmax,
                              const float* relu_out, int64_t ldr, int64_t n_dst, int fanout, int d, int op,
                              int64_t n_src, float* dsrc, int64_t lds, ogl_stream_t stream) {
  if (n_dst < 0 || fanout < 0 || d < 0 || n_src < 0 || lds < d || ldo < d) return OGL_EINVAL;
  if (n_dst == 0 || d == 0 || fanout == 0) return OGL_OK;
  if (!dout || !dsrc) return OGL_EINVAL;
  dim3 grid((unsigned)ogl_cdiv(n_dst, 4)), block(256);
  if (op == OGL_REDUCE_MAX) {
    if (!argmax || (relu_out && ldr < d)) return OGL_EINVAL;
    hipLaunchKernelGGL(k_reduce_bwd_max, grid, block, 0, (hipStream_t)stream, dout, ldo, argmax, relu_out, ldr, n_dst, d,
                       n_src, dsrc, lds);
  } else if (op == OGL_REDUCE_MEAN || op == OGL_REDUCE_SUM) {
    if (!idx32 || relu_out) return OGL_EINVAL;
    float scale = op == OGL_REDUCE_MEAN ? 1.0f / (float)fanout : 1.0f;
    hipLaunchKernelGGL(k_reduce_bwd_sum, grid, block, 0, (hipStream_t)stream, dout, ldo, idx32, n_dst, fanout, d, scale, n_src, dsrc, lds);
  } else {
    return OGL_EINVAL;
  }
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- row gather ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_rows_v4(const float* __restrict__ table, int64_t ld, int64_t n_rows,
                                                        const int64_t* __restrict__ ids, int64_t n, int d4,
                                                        float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n) return;
  int64_t r = ids[w];
  const bool ok = r >= 0 && r < n_rows;
  const float4* rp = (const float4*)(table + (ok ? r : 0) * ld);
  float4* op = (float4*)(out + w * ldo);
  for (int c = lane; c < d4; c += 64) op[c] = ok ? rp[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(256) k_gather_rows_generic(const float* __restrict__ table, int64_t ld, int64_t n_rows,
                                                             const int64_t* __restrict__ ids, int64_t n, int d,
                                                             float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n) return;
  int64_t r = ids[w];
  const bool ok = r >= 0 && r < n_rows;
  for (int c = lane; c < d; c += 64) out[w * ldo + c] = ok ? table[r * ld + c] : 0.f;
}

extern "C" int ogl_gather_rows(const float* table, int64_t ld, int64_t n_rows, const int64_t* ids, int64_t n,
                               int d, float* out, int64_t ldo, ogl_stream_t stream) {
  if (n < 0 || d < 0 || ld < d || ldo < d || n_rows < 0) return OGL_EINVAL;
  if (n == 0 || d == 0) return OGL_OK;
  if (!table || !ids || !out) return OGL_EINVAL;
  const int d4 = (d + 3) / 4;
  const bool vec = (ld % 4 == 0) && (ldo % 4 == 0) && ld >= 4 * d4 && ldo >= 4 * d4 &&
                   (((uintptr_t)table & 15) == 0) && (((uintptr_t)out & 15) == 0);
  dim3 grid((unsigned)ogl_cdiv(n, 4)), block(256);
  if (vec)
    hipLaunchKernelGGL(k_gather_rows_v4, grid, block, 0, (hipStream_t)stream, table, ld, n_rows, ids, n, d4, out, ldo);
  else
    hipLaunchKernelGGL(k_gather_rows_generic, grid, block, 0, (hipStream_t)stream, table, ld, n_rows, ids, n, d, out, ldo);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

__global__ void __launch_bounds__(256) k_gather_i64(const int64_t* __restrict__ table, int64_t n_rows,
                                                    const int64_t* __restrict__ ids, int64_t n,
                                                    int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t r = ids[i];
  out[i] = (r >= 0 && r < n_rows) ? table[r] : -1;
}

extern "C" int ogl_gather_i64(const int64_t* table, int64_t n_rows, const int64_t* ids, int64_t n, int64_t* out,
                              ogl_stream_t stream) {
  if (n < 0 || n_rows < 0) return OGL_EINVAL;
  if (n == 0) return OGL_OK;
  if (!table || !ids || !out) return OGL_EINVAL;
  hipLaunchKernelGGL(k_gather_i64, dim3((unsigned)ogl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, table,
                     n_rows, ids, n, out);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- zero fill (the scatter targets of the atomic backward kernels) ------------------------------------------------
// A kernel, not hipMemsetAsync: a memset node recorded into a captured hipGraph re-runs on only 1/16 of its range from the second
// replay on (ROCm 7.2; tools/graph_probe.py), and not an ATen fill: the step's launches are this library's.
__global__ void __launch_bounds__(256) k_zero16(uint4* __restrict__ p, int64_t n16, unsigned char* __restrict__ tail, int ntail) {
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (t0 < ntail) tail[t0] = 0;
}

extern "C" int ogl_fill_zero(void* ptr, int64_t bytes, ogl_stream_t stream) {
  if (bytes < 0 || (bytes > 0 && !ptr)) return OGL_EINVAL;
  if (bytes == 0) return OGL_OK;
  unsigned char* b = (unsigned char*)ptr;
  const int64_t head = std::min<int64_t>(bytes, (16 - ((uintptr_t)b & 15)) & 15);      // bytes up to the first 16-byte boundary
  const int64_t n16 = (bytes - head) / 16, ntail = bytes - head - n16 * 16;
  if (head > 0) {
    hipLaunchKernelGGL(k_zero16, dim3(1), dim3(256), 0, (hipStream_t)stream, (uint4*)nullptr, (int64_t)0, b, (int)head);
    OGL_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_zero16, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(ogl_cdiv(n16, 256), 2048))), dim3(256), 0, (hipStream_t)stream,
                     (uint4*)(b + head), n16, b + head + n16 * 16, (int)ntail);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- stream copy: the box's own streaming rate (measurement only: bench.py's `hbm_copy_measured`) ----
// ONE float4 per thread, no loop: of the forms tools/micro/stream_copy.hip tries on 1 GiB -> 1 GiB (profiles/r06_stream_copy.txt) this is
// the fastest — 6.17 TB/s of read + write bytes, what MI355X_MICROARCH.md quotes as achievable (6.29) — against 5.3-5.7 for a block-contiguous
// loop, 4.1-5.2 for a grid-stride loop, 5.14 for hipMemcpyAsync and 4.7 for torch's copy_.
__global__ void __launch_bounds__(256) k_stream_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

extern "C" int ogl_stream_copy(const void* src, void* dst, int64_t bytes, ogl_stream_t stream) {
  if (bytes < 0 || (bytes & 15) || (bytes > 0 && (!src || !dst)) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return OGL_EINVAL;
  if (bytes == 0) return OGL_OK;
  const int64_t n16 = bytes / 16;
  if (ogl_cdiv(n16, 256) > 0x7FFFFFFFll) return OGL_EINVAL;
  hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)ogl_cdiv(n16, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, n16);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}

// ---- feat_drop: counter-based dropout, optionally fused with the row gather ------------------------------------
// One thread per 4 consecutive columns of one output row = one Philox4x32-10 block (the sampler's keying with the
// output row in place of the vertex id); keep iff draw >= thr; kept values are divided by keep_prob (one IEEE division,
// what the oracle's numpy statement does).
__global__ void __launch_bounds__(256) k_dropout_rows(const float* __restrict__ src, int64_t ld,
                                                      const int64_t* __restrict__ rows, int64_t nrows, int64_t M, int N,
                                                      int n4, uint32_t thr, float keep_prob, uint32_t k0, uint32_t k1,
                                                      uint32_t c3, float* __restrict__ out, int64_t ldo) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * (int64_t)n4) return;
  const int64_t i = t / n4;
  const int q = (int)(t - i * n4);
  const int64_t r = rows ? rows[i] : i;
  const bool ok = r >= 0 && r < nrows;
  const philox4 d = philox4x32_10((uint32_t)q, (uint32_t)((uint64_t)i & 0xFFFFFFFFu), (uint32_t)((uint64_t)i >> 32), c3, k0, k1);
  const uint32_t draw[4] = {d.x, d.y, d.z, d.w};
  const float* sp = src + (ok ? r : 0) * ld;
  float* op = out + i * ldo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * q + e;
    if (j < N) {
      const float v = ok ? sp[j] : 0.f;
      op[j] = draw[e] >= thr ? v / keep_prob : 0.f;
    }
  }
}

extern "C" int ogl_dropout_rows(const float* src, int64_t ld, const int64_t* rows, int64_t nrows, int64_t M, int N,
                                double p, uint64_t seed, uint64_t ctr, float* out, int64_t ldo, ogl_stream_t stream) {
  if (M < 0 || N < 0 || ld < N || ldo < N || nrows < 0 || !(p >= 0.0) || !(p < 1.0)) return OGL_EINVAL;
  if (M == 0 || N == 0) return OGL_OK;
  if (!src || !out) return OGL_EINVAL;
  const double scaled = p * 4294967296.0;
  const uint32_t thr = scaled >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)scaled;
  const float keep_prob = (float)(1.0 - p);
  const int n4 = (N + 3) / 4;
  const uint32_t k0 = (uint32_t)(seed & 0xFFFFFFFFu), k1 = (uint32_t)((seed >> 32) ^ (ctr >> 32));
  hipLaunchKernelGGL(k_dropout_rows, dim3((unsigned)ogl_cdiv(M * (int64_t)n4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     ld, rows, nrows, M, N, n4, thr, keep_prob, k0, k1, (uint32_t)(ctr & 0xFFFFFFFFu), out, ldo);
  OGL_CHECK_LAUNCH();
  return OGL_OK;
}
