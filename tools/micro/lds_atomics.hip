// LDS atomic throughput on MI355X: one block per CU, 640 threads, every thread issues R atomics to pseudo-random slab cells.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(640) k(const unsigned* __restrict__ addr, int n, int rounds, float* out) {
  extern __shared__ unsigned char smem[];
  float* f = (float*)smem; unsigned* u = (unsigned*)smem; unsigned long long* q = (unsigned long long*)smem;
  const int cells = MODE == 2 || MODE == 5 ? 9632 : 19264;     // 77 KB of fp32 / u32, or of u64
  for (int i = threadIdx.x; i < 19264; i += 640) u[i] = 0;
  __syncthreads();
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    const unsigned a = addr[(r * 640 + threadIdx.x) % n] % cells;
    if (MODE == 0) __hip_atomic_fetch_add(&f[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 1) __hip_atomic_fetch_add(&u[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 2) __hip_atomic_fetch_add(&q[a], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 3) acc += __hip_atomic_fetch_add(&f[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 4) acc += (float)__hip_atomic_fetch_add(&u[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 5) acc += (float)__hip_atomic_fetch_add(&q[a], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 6) { f[a] = f[a] + 1.0f; }                      // plain read-modify-write (racy: rate only)
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc + f[1];
}
template <int MODE> void run(const char* name, unsigned* addr, int n, float* out) {
  const int rounds = 2000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(640), 78 * 1024, 0, addr, n, rounds, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double ops = 640.0 * rounds;       // lane-operations per CU
  printf("%-28s %.3f ms: %.2f lane-ops per ns per CU (~%.2f per clock at 2.1 GHz)\n", name, best, ops / (best * 1e6), ops / (best * 1e6) / 2.1);
}
int main() {
  const int n = 1 << 20;
  unsigned* h = (unsigned*)malloc(n * 4);
  unsigned x = 12345;
  for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = x >> 8; }
  unsigned* addr; float* out;
  hipMalloc(&addr, n * 4); hipMalloc(&out, 4096);
  hipMemcpy(addr, h, n * 4, hipMemcpyHostToDevice);
  run<0>("ds_add_f32 (no return)", addr, n, out);
  run<1>("ds_add_u32 (no return)", addr, n, out);
  run<2>("ds_add_u64 (no return)", addr, n, out);
  run<3>("ds_add_rtn_f32", addr, n, out);
  run<4>("ds_add_rtn_u32", addr, n, out);
  run<5>("ds_add_rtn_u64", addr, n, out);
  run<6>("plain ds_read + ds_write", addr, n, out);
  return 0;
}
