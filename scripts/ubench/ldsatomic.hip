// Micro-benchmark: LDS atomic-add throughput per CU for the counter layouts the histogram kernel
// can use (C lane-indexed copies of 768 counters), random vs all-equal bins, atomics vs plain stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int C, int MODE, int T>  // MODE 0 random, 1 all-equal, 2 random + plain store
__global__ __launch_bounds__(T) void k(int iters, unsigned* out) {
  extern __shared__ unsigned sh[];
  for (int i = threadIdx.x; i < 768 * C; i += T) sh[i] = 0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const unsigned copy = threadIdx.x & (C - 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x = x * 1664525u + 1013904223u;
      unsigned bin = MODE == 1 ? 7u : (x >> 24);
      unsigned ch = (u % 3) * 256;
      unsigned* p = sh + (ch + bin) * C + copy;
      if (MODE == 2) *(volatile unsigned*)p = x;
      else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  unsigned s = 0;
  for (int i = threadIdx.x; i < 768 * C; i += T) s += sh[i];
  if (s == 0x12345u) out[0] = s + x;
}

template <int C, int MODE, int T>
int run(const char* name, unsigned* out) {
  const int iters = 2000;
  const size_t lds = 768 * C * 4;
  CK(hipFuncSetAttribute((const void*)k<C, MODE, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)(160 * 1024 / lds);
  int by_threads = 2048 / T;
  if (per_cu > by_threads) per_cu = by_threads;
  const int blocks = 256 * per_cu;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<C, MODE, T>), dim3(blocks), dim3(T), lds, 0, 50, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<C, MODE, T>), dim3(blocks), dim3(T), lds, 0, iters, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double ops = (double)blocks * T * iters * 16;
  printf("%-28s C=%2d T=%4d wg/cu=%d  %.3f ms  %.2f ops/clk/CU (at 2.4 GHz)  => %.2f TB/s of bytes\n", name, C, T, per_cu, ms,
         ops / (ms * 1e-3) / 2.4e9 / 256, ops / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  unsigned* out; CK(hipMalloc(&out, 64));
  run<8, 0, 256>("random atomics", out);
  run<16, 0, 256>("random atomics", out);
  run<32, 0, 256>("random atomics", out);
  run<32, 0, 512>("random atomics", out);
  run<32, 0, 1024>("random atomics", out);
  run<16, 0, 512>("random atomics", out);
  run<8, 1, 256>("all-equal atomics", out);
  run<16, 1, 256>("all-equal atomics", out);
  run<32, 1, 1024>("all-equal atomics", out);
  run<8, 2, 256>("random plain stores", out);
  run<32, 2, 1024>("random plain stores", out);
  return 0;
}
