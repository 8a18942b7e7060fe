// Micro-benchmark: issue rate of the VALU instructions the polynomial-expansion kernel is made of
// (clocks per wave-instruction per SIMD, 2.4 GHz assumed; 4 = full rate for a wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  // 8 independent chains per thread hide the instruction latency
  float f[8]; double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = seed + i + threadIdx.x; d[i] = seed * 3 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
      if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
      if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
      if (MODE == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      if (MODE == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      if (MODE == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
      if (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
      if (MODE == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
  if (s == 1234.5f) out[0] = s;
}

template <int MODE>
int run(const char* name, float* o) {
  const int iters = 20000, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double winstr_per_simd = (double)blocks * 4 * iters * 8 / (256.0 * 4);
  printf("%-16s %.3f ms  %.2f clk per wave-instruction per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / winstr_per_simd);
  return 0;
}

int main() {
  float* o; CK(hipMalloc(&o, 4));
  run<0>("v_add_f32", o); run<7>("v_mul_f32", o); run<6>("v_pk_add_f32", o);
  run<1>("v_add_f64", o); run<5>("v_mul_f64", o); run<2>("v_fma_f64", o);
  run<3>("v_cvt_f64_f32", o); run<4>("v_cvt_f32_f64", o);
  return 0;
}
