// Streaming-copy rate of this box (the HBM denominator SURVEY.md section 8(d) asks to re-measure): 1 GiB -> 1 GiB, read + write bytes over
// HIP-event time, several forms of a float4 copy (what ogl_stream_copy could be) beside hipMemcpyAsync device-to-device.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_copy.hip -o tools/micro/stream_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stride(const u32x4* __restrict__ s, u32x4* __restrict__ d, int64_t n) {
  const int64_t st = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * st < n; i += U * st) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * st) : s[i + u * st];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * st); else d[i + u * st] = v[u]; }
  }
  for (; i < n; i += st) d[i] = s[i];
}
// every block copies ONE contiguous chunk (block-contiguous instead of grid-strided)
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_chunk(const u32x4* __restrict__ s, u32x4* __restrict__ d, int64_t n) {
  const int64_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  int64_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
  }
  for (; i < hi; i += 256) d[i] = s[i];
}
template <class F> static float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 10;
}
int main() {
  const int64_t bytes = 1ll << 30, n = bytes / 16;
  u32x4 *s, *d;
  hipMalloc(&s, bytes); hipMalloc(&d, bytes);
  hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  auto rep = [&](const char* name, float ms) { printf("%-58s %7.3f ms  %7.1f GB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
  rep("hipMemcpyAsync device-to-device", timeit([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }));
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    char nm[96];
    snprintf(nm, 96, "grid-stride float4 x4, %d blocks", blocks); rep(nm, timeit([&] { hipLaunchKernelGGL((k_stride<4, false>), dim3(blocks), dim3(256), 0, 0, s, d, n); }));
    snprintf(nm, 96, "grid-stride float4 x4 nontemporal, %d blocks", blocks); rep(nm, timeit([&] { hipLaunchKernelGGL((k_stride<4, true>), dim3(blocks), dim3(256), 0, 0, s, d, n); }));
    snprintf(nm, 96, "grid-stride float4 x8, %d blocks", blocks); rep(nm, timeit([&] { hipLaunchKernelGGL((k_stride<8, false>), dim3(blocks), dim3(256), 0, 0, s, d, n); }));
    snprintf(nm, 96, "block-contiguous float4 x4, %d blocks", blocks); rep(nm, timeit([&] { hipLaunchKernelGGL((k_chunk<4, false>), dim3(blocks), dim3(256), 0, 0, s, d, n); }));
    snprintf(nm, 96, "block-contiguous float4 x8 nontemporal, %d blocks", blocks); rep(nm, timeit([&] { hipLaunchKernelGGL((k_chunk<8, true>), dim3(blocks), dim3(256), 0, 0, s, d, n); }));
  }
  // one element per thread, no loop (what a plain elementwise launch does)
  rep("one float4 per thread (262 144 blocks)", timeit([&] { hipLaunchKernelGGL((k_stride<1, false>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, s, d, n); }));
  // read-only and write-only rates (a copy's two halves)
  return 0;
}
