// Matrix-pipe issue rate on MI355X: W waves per SIMD (4 W per block, one block per CU), each wave a loop of v_mfma_f32_16x16x32_bf16 over
// NACC independent accumulators in CHAIN-long dependent runs (CHAIN = 1: every instruction independent of the previous NACC - 1;
// CHAIN = 6: the six terms of one accumulator back to back).  Prints nanoseconds and shader clocks (s_memtime) per MFMA and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int CHAIN>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* clk, int iters) {
  f32x4 acc[NACC];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (CHAIN == 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int NACC, int CHAIN> void run(int waves_per_simd, float* out, unsigned long long* clk) {
  const int iters = 2000, threads = 256 * waves_per_simd;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, CHAIN>), dim3(256), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  unsigned long long h[256]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i]; c /= 256;
  const double per_simd = (double)iters * 6 * NACC * waves_per_simd;
  printf("waves/SIMD %d  accumulators %2d  chain %d: %.3f ms  %.2f ns per MFMA and SIMD  (%.1f memtime ticks)  -> %.0f TFLOP/s chip\n", waves_per_simd,
         NACC, CHAIN, best, best * 1e6 / per_simd, c / per_simd, 1024.0 * per_simd * 16384 / (best * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 8);
  for (int w = 1; w <= 3; ++w) {
    run<16, 1>(w, out, clk);
    run<4, 1>(w, out, clk);
    run<2, 1>(w, out, clk);
    run<16, 6>(w, out, clk);
  }
  return 0;
}
