// Split-bf16 ("x6") arithmetic shared by the projection GEMMs (linear.hip: operands split on the fly;
// linear_x3.hip: operands pre-split into bf16x3 images).  gfx950 only.
#pragma once
#include "ogl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef v4i16 __attribute__((address_space(3))) lds_v4i16;

// ---- split-bf16 ("x6") arithmetic -------------------------------------------------------------------------------
// An fp32 value is split EXACTLY into three bf16 terms x = x1 + x2 + x3 (8 + 8 + 8 significand bits, each
// residual is exact in fp32); a*b = sum_{i,j} a_i*b_j and the six terms with i + j <= 4 are accumulated in fp32
// by v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in fp32).  The dropped terms are <= 2^-23 |a*b|,
// i.e. at the level of one fp32 rounding of the product, so the result carries fp32-GEMM accuracy while the matrix
// pipe runs at 16x the fp32-MFMA rate for 6x the instructions.
__device__ __forceinline__ unsigned pk_bf16(float x, float y) {
  f32x2 v = {x, y};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void split3(float x, float y, unsigned& h, unsigned& m, unsigned& l) {
  h = pk_bf16(x, y);
  const float rx = x - __builtin_bit_cast(float, h << 16), ry = y - __builtin_bit_cast(float, h & 0xFFFF0000u);
  m = pk_bf16(rx, ry);
  const float sx = rx - __builtin_bit_cast(float, m << 16), sy = ry - __builtin_bit_cast(float, m & 0xFFFF0000u);
  l = pk_bf16(sx, sy);
}

// Byte order inside one 192-byte group of a bf16x3 image (32 reduction elements): three 64-byte planes (hi, mid, lo
// term of the split).  x3_piece(c, p) = index of the 16-byte piece holding elements 8c .. 8c+7 of plane p (its byte offset
// is 16 x that).  (A half-major order — [2 halves][3 planes][16], for 16-deep GEMM steps — was tried and measured 14 %
// slower in the image producers: their 16-byte stores then come in 32-byte runs instead of 64-byte ones.)
__host__ __device__ __forceinline__ constexpr int x3_piece(int c, int p) { return p * 4 + c; }

// The TRANSPOSED, group-major image of a gradient matrix dP [n_src, D] (what the layer-0 weight-gradient product reads as its A operand:
// pool_bwd_x3.hip, reduce_seg.hip): sources dealt round-robin over G = ceil(n_src / 32) groups (source s = lane * G + group), group b
// one contiguous slab of (D + 1) rows x 192 bytes — row f = the split of dP[the group's 32 sources, f], row D all zero.
// emit: image row f, group b = split(T[0..31][f]); the whole slab is one contiguous (D + 1) * 192-byte run
__device__ __forceinline__ void pb_emit(const float* T, int D, int DP, int b, unsigned char* __restrict__ img, int64_t gstride, int tid, int nthreads) {
  for (int u = tid; u < D * 4; u += nthreads) {                   // unit = (row f, 8-source chunk c): one split, three stores
    const int fo = u >> 2, c = u & 3;
    const float* col = T + (8 * c) * DP + fo;
    uint4 o[3];
    split3(col[0], col[DP], o[0].x, o[1].x, o[2].x);
    split3(col[2 * DP], col[3 * DP], o[0].y, o[1].y, o[2].y);
    split3(col[4 * DP], col[5 * DP], o[0].z, o[1].z, o[2].z);
    split3(col[6 * DP], col[7 * DP], o[0].w, o[1].w, o[2].w);
    unsigned char* d = img + (int64_t)b * gstride + (int64_t)fo * 192;
#pragma unroll
    for (int p = 0; p < 3; ++p) *(uint4*)(d + x3_piece(c, p) * 16) = o[p];
  }
  if (tid < 12) *(uint4*)(img + (int64_t)b * gstride + (int64_t)D * 192 + tid * 16) = make_uint4(0, 0, 0, 0);   // the zero row
}

// 16-byte load from a 4-byte-aligned address: gfx950 under HSA runs in unaligned-access mode, the
// compiler emits one global_load_dwordx4 (rows such as K = 602 floats are only 8-B aligned).
__device__ __forceinline__ float4 ld16(const float* p) {
  float4 t;
  __builtin_memcpy(&t, p, 16);
  return t;
}

