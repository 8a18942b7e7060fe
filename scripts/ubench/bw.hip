// Micro-benchmark: achievable HBM bandwidth on MI355X for the access shapes the flow kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void copy1(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read1(const float* __restrict__ a, float* __restrict__ out, size_t n) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void read4(const float4* __restrict__ a, float* __restrict__ out, size_t n4) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
// read1 with 4 independent loads per iteration (ILP)
__global__ void read1x4(const float* __restrict__ a, float* __restrict__ out, size_t n) {
  float s = 0;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) { float v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride]; s += v0 + v1 + v2 + v3; }
  if (s == 12345.678f) out[0] = s;
}
// 3 reads : 1 write per element, planar (like the fused blur: M, R0, R1 in; M' out), dword per lane
__global__ void rw31_1(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = a[i] + b[i] * c[i];
}
__global__ void rw31_4(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float4* __restrict__ d, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 x = a[i], y = b[i], z = c[i];
    d[i] = make_float4(x.x + y.x * z.x, x.y + y.y * z.y, x.z + y.z * z.z, x.w + y.w * z.w);
  }
}
__global__ void write1(float* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = 1.0f;
}
__global__ void write4(float4* __restrict__ b, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1, 2, 3, 4);
}

template <typename F> double timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  size_t n = (size_t)1 << 30;  // 4 GiB per float array
  float *a, *b, *c, *d;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
  CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4)); CK(hipMemset(c, 3, n * 4));
  for (int blocks : {2048, 4096, 16384}) {
    printf("grid %d x 256\n", blocks);
    double t;
    t = timeit([&] { hipLaunchKernelGGL(copy1, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
    printf("  copy  dword/lane : %7.1f GB/s\n", 2.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(copy4, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); }, 5);
    printf("  copy  16B/lane   : %7.1f GB/s\n", 2.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(read1, dim3(blocks), dim3(256), 0, 0, a, d, n); }, 5);
    printf("  read  dword/lane : %7.1f GB/s\n", 1.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(read1x4, dim3(blocks), dim3(256), 0, 0, a, d, n); }, 5);
    printf("  read  dword x4ILP: %7.1f GB/s\n", 1.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(read4, dim3(blocks), dim3(256), 0, 0, (const float4*)a, d, n / 4); }, 5);
    printf("  read  16B/lane   : %7.1f GB/s\n", 1.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(write1, dim3(blocks), dim3(256), 0, 0, b, n); }, 5);
    printf("  write dword/lane : %7.1f GB/s\n", 1.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(write4, dim3(blocks), dim3(256), 0, 0, (float4*)b, n / 4); }, 5);
    printf("  write 16B/lane   : %7.1f GB/s\n", 1.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(rw31_1, dim3(blocks), dim3(256), 0, 0, a, b, c, d, n); }, 5);
    printf("  3r:1w dword/lane : %7.1f GB/s\n", 4.0 * n * 4 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(rw31_4, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (const float4*)b, (const float4*)c, (float4*)d, n / 4); }, 5);
    printf("  3r:1w 16B/lane   : %7.1f GB/s\n", 4.0 * n * 4 / t / 1e6);
  }
  return 0;
}
