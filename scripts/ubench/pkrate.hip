// Micro-benchmark: issue cost of packed fp32 (v_pk_mul_f32 / v_pk_add_f32, two floats per lane per instruction) against the
// scalar forms, and of the f64 conversions, at 2 waves per SIMD with 8 independent chains (the steady state of k_polyexp /
// k_flow_iter3).  Clocks per wave-instruction per SIMD, 2.4 GHz assumed.  hipcc --offload-arch=gfx950 -O3 pkrate.hip -o pkrate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float f[8]; v2f p[8]; double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { f[i] = seed + i + threadIdx.x; p[i].x = f[i]; p[i].y = f[i] * 0.5f; d[i] = seed * 3 + i; }
  v2f ps; ps.x = seed; ps.y = seed * 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
      if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ps));
      if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ps));
      if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1 neg_hi:[0,1]" : "+v"(p[i]) : "v"(ps));
      if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(ps));
      if (MODE == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      if (MODE == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[7]));
      if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(ps));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i] + p[i].x + p[i].y + (float)d[i];
  if (s == 1234.5f) out[0] = s;
}

template <int MODE>
int run(const char* name, float* o, int wg_per_cu) {
  const int iters = 4000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, o, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, o, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)iters * 8;
  printf("%-34s waves/SIMD %d: %.3f ms  %.2f clk per instruction per SIMD\n", name, wg_per_cu, ms, ms * 1e-3 * 2.4e9 / n / wg_per_cu);
  return 0;
}

int main() {
  float* o; CK(hipMalloc(&o, 4));
  for (int w : {2, 4}) {
    run<0>("v_mul_f32", o, w); run<1>("v_pk_mul_f32", o, w); run<2>("v_pk_add_f32", o, w); run<3>("v_pk_add_f32 neg_hi", o, w);
    run<4>("v_pk_mul_f32 op_sel_hi broadcast", o, w); run<5>("v_cvt_f64_f32", o, w); run<6>("v_add_f64", o, w); run<7>("v_pk_fma_f32", o, w);
  }
  return 0;
}
