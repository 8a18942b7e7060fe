// Micro-benchmark: cost of per-lane loads of 40 contiguous bytes at a 20-byte lane stride
// (AoS 5-float pixels, two neighbours) issued as 10 dwords, 5 dwordx2 or 2 dwordx4 + 1 dwordx2,
// versus the planar equivalent (10 dwordx2 at 4-byte lane stride from 5 planes x 2 rows).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE>
__global__ void k(const float* __restrict__ a, float* __restrict__ out, int n_px, int iters) {
  float s = 0;
  int px = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    int p = (px + it * 7919 * 64) % (n_px - 2);
    if (MODE == 0) {  // AoS, 10 dword loads
      const float* q = a + 5 * (size_t)p;
#pragma unroll
      for (int i = 0; i < 10; ++i) s += q[i];
    } else if (MODE == 1) {  // AoS, 5 dwordx2
      const float* q = a + 5 * (size_t)p;
#pragma unroll
      for (int i = 0; i < 5; ++i) { f2u v = *(const f2u*)(q + 2 * i); s += v.x + v.y; }
    } else if (MODE == 2) {  // AoS, 2 dwordx4 + 1 dwordx2
      const float* q = a + 5 * (size_t)p;
      f4u v0 = *(const f4u*)q, v1 = *(const f4u*)(q + 4); f2u v2 = *(const f2u*)(q + 8);
      s += v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w + v2.x + v2.y;
    } else {  // planar: 5 planes, one dwordx2 each (x, x+1)
#pragma unroll
      for (int c = 0; c < 5; ++c) { f2u v = *(const f2u*)(a + (size_t)c * n_px + p); s += v.x + v.y; }
    }
  }
  if (s == 1234.5f) out[0] = s;
}

int main() {
  int n_px = 1920 * 1080;  // 41 MB of planes: L2/MALL resident like the R fields of one frame
  float *a, *o;
  hipMalloc(&a, (size_t)n_px * 5 * 4 + 64); hipMalloc(&o, 4);
  hipMemset(a, 0, (size_t)n_px * 5 * 4 + 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 64, blocks = 2048;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, a, o, n_px, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, a, o, n_px, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, a, o, n_px, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, a, o, n_px, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lanes = (double)blocks * 256 * iters;
    const char* names[] = {"AoS 10 x dword", "AoS 5 x dwordx2", "AoS 2 x dwordx4 + dwordx2", "planar 5 x dwordx2"};
    printf("%-28s %.3f ms  %.1f ps per 40-byte (20 for planar) fetch per lane, %.0f GB/s requested\n", names[mode], ms,
           ms * 1e9 / lanes, lanes * (mode == 3 ? 40 : 40) / (ms * 1e-3) / 1e9);
  }
  return 0;
}
