// Micro-benchmark: VALU issue rate of ONE wave per SIMD against several, by the number of independent
// dependency chains in the instruction stream (a small launch leaves one wave per SIMD: does the SIMD then
// still issue one wave-instruction per 4 clocks?).  Clocks per wave-instruction per WAVE, 2.4 GHz assumed.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int MODE, int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float f[8]; double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = seed + i + threadIdx.x; d[i] = seed * 3 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8 / CH; ++rep)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
      if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[7]));
      if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[7]));
      if (MODE == 3) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i])); }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
  if (s == 1234.5f) out[0] = s;
}

template <int MODE, int CH>
int run(const char* name, float* o, int wg_per_cu) {
  const int iters = 4000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, CH>), dim3(blocks), dim3(256), 0, 0, o, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, CH>), dim3(blocks), dim3(256), 0, 0, o, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)iters * 8 * (MODE == 3 ? 2 : 1);
  printf("%-14s chains %d  waves/SIMD %d: %.3f ms  %.2f clk per instruction per wave, %.2f per SIMD\n", name, CH, wg_per_cu, ms,
         ms * 1e-3 * 2.4e9 / n, ms * 1e-3 * 2.4e9 / n / wg_per_cu);
  return 0;
}

template <int MODE>
int sweep(const char* name, float* o) {
  for (int w : {1, 2, 4}) {
    run<MODE, 1>(name, o, w); run<MODE, 2>(name, o, w); run<MODE, 4>(name, o, w); run<MODE, 8>(name, o, w);
  }
  return 0;
}

int main() {
  float* o; CK(hipMalloc(&o, 4));
  sweep<0>("v_add_f32", o); sweep<1>("v_add_f64", o); sweep<2>("v_fma_f64", o); sweep<3>("cvt f64<->f32", o);
  return 0;
}
