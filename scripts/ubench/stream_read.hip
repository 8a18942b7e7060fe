// Micro-benchmark: how long does the chip take to READ S bytes once (the Histogram kernel's memory side without its LDS work)?
// Launches of `wg` workgroups of 1024 threads, non-temporal 16-byte loads, six in flight per thread as in k_hist_u8c3_v2;
// R launches back to back between one event pair (per-launch figure includes the ~1.5 us launch boundary).
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/stream_read.hip -o /tmp/stream_read && /tmp/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef unsigned u4nt __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(1024) void k_read(const u4nt* __restrict__ p, long long nvec, unsigned* out) {
  const long long per = (nvec + gridDim.x - 1) / gridDim.x;
  const long long v0 = per * blockIdx.x;
  long long v1 = v0 + per; if (v1 > nvec) v1 = nvec;
  unsigned acc = 0;
  long long i = v0 + threadIdx.x;
  for (; i + 5 * 1024 < v1; i += 6 * 1024) {
    u4nt a, b, c, d, e, f;
    if (NT) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + 1024); c = __builtin_nontemporal_load(p + i + 2048);
              d = __builtin_nontemporal_load(p + i + 3072); e = __builtin_nontemporal_load(p + i + 4096); f = __builtin_nontemporal_load(p + i + 5120); }
    else { a = p[i]; b = p[i + 1024]; c = p[i + 2048]; d = p[i + 3072]; e = p[i + 4096]; f = p[i + 5120]; }
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w ^ e.x ^ e.y ^ e.z ^ e.w ^ f.x ^ f.y ^ f.z ^ f.w;
  }
  for (; i < v1; i += 1024) { u4nt a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x12345679u) out[0] = acc;
}

int main() {
  const size_t frame = 3ull * 1080 * 1920;
  unsigned char* buf; unsigned* out;
  CK(hipMalloc(&buf, frame * 256)); CK(hipMalloc(&out, 4));
  CK(hipMemset(buf, 1, frame * 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nf : {8, 16, 32, 64, 128, 256})
    for (int wg : {256, 512, 1024})
      for (int nt = 0; nt < 2; ++nt) {
        const long long nvec = (long long)(frame * nf / 16);
        const int R = 50;
        for (int w = 0; w < 3; ++w) {
          if (nt) hipLaunchKernelGGL(k_read<1>, dim3(wg), dim3(1024), 0, 0, (const u4nt*)buf, nvec, out);
          else hipLaunchKernelGGL(k_read<0>, dim3(wg), dim3(1024), 0, 0, (const u4nt*)buf, nvec, out);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < R; ++r) {
          if (nt) hipLaunchKernelGGL(k_read<1>, dim3(wg), dim3(1024), 0, 0, (const u4nt*)buf, nvec, out);
          else hipLaunchKernelGGL(k_read<0>, dim3(wg), dim3(1024), 0, 0, (const u4nt*)buf, nvec, out);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / R;
        printf("frames %3d  wg %4d  nt %d : %7.2f us per launch  %6.0f GB/s  (%.3f of 8 TB/s)\n", nf, wg, nt, us, frame * nf / us / 1e3, frame * nf / us / 1e3 / 8000);
      }
  return 0;
}
