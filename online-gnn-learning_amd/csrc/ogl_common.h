// Shared device/host helpers for libogl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ogl_hip.h"

extern thread_local int g_ogl_last_hip_error;

#define OGL_CHECK_HIP(expr)                         \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) {                         \
      g_ogl_last_hip_error = (int)_e;               \
      return OGL_EHIP;                              \
    }                                               \
  } while (0)

#define OGL_CHECK_LAUNCH() OGL_CHECK_HIP(hipGetLastError())

// the diagnostic knobs behind ogl_debug_set (csrc/graph.hip), each next to the kernel it pins; library-internal
#define OGL_INTERNAL __attribute__((visibility("hidden")))
OGL_INTERNAL int oglx_knob_x3_tile(int cfg, int* prev);          // linear_x3.hip
OGL_INTERNAL int oglx_knob_x3_stagger(int on, int* prev);        // linear_x3.hip
OGL_INTERNAL int oglx_knob_block_min_lds(int on, int* prev);     // block.hip
OGL_INTERNAL int oglx_knob_reduce_half(int on, int* prev);       // aggregate.hip
OGL_INTERNAL int oglx_knob_seg_rows(int on, int* prev);          // reduce_seg.hip

static inline int64_t ogl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t ogl_round_up(int64_t a, int64_t b) { return ogl_cdiv(a, b) * b; }
__device__ static inline int64_t ogl_cdiv_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Descriptor of up to OGL_MAX_BATCH independent (destinations -> picks -> block) problems, passed by value in the kernarg.
#define OGL_MAX_BATCH 64
struct ogl_batch_desc {
  int64_t dst_start[OGL_MAX_BATCH];     // batch b's destinations: dst_base[dst_start[b] ..)
  int64_t row_off[OGL_MAX_BATCH + 1];   // packed output rows: running sum of the destination counts
  uint64_t ctr[OGL_MAX_BATCH];          // Philox batch counters (sampler only)
};

struct ogl_graph {
  const int64_t* indptr;
  const int32_t* indices;
  const int32_t* keys;
  int64_t n;
  int64_t nnz;
  int32_t* deg;  // owned, int32[n]
  int64_t n_present;
  int64_t cut;
};

// Philox4x32-10 (Salmon et al. SC'11); pinned by the Random123 KAT in tests/test_oracle.py via
// the oracle, and HIP-vs-oracle bit-exact in tests/test_gpu_sampler.py.
struct philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ static inline philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                        uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  philox4 o = {c0, c1, c2, c3};
  return o;
}
