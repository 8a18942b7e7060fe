// Micro-benchmark: what does a vector-memory instruction cost the texture-address unit when only SOME of its lanes are
// active?  (Round 5: a texel-reuse scheme for k_flow_iter3 -- take the neighbour lane's / previous row's R1 texel when
// the gather geometry allows, load it otherwise -- pays on near-integer flows only if the fallback loads of the few
// failing lanes are cheaper than full-wave loads.)  dwordx4 and dword loads at the kernel's occupancy (8 waves per
// CU), L2-resident footprint, under exec masks: all lanes, one lane, 16 contiguous lanes, every 8th lane, a random
// ~12 % of lanes, and "wave-uniform skip" (the instruction is branched over in 7 of 8 iterations).  Also checks the
// semantics of the DPP wave shifts used for lane sharing and times a DPP move against a plain VALU op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// PAT 0: all lanes; 1: lane 63 only; 2: lanes 0..15; 3: lane % 8 == 0; 4: hashed ~12 %; 5: lanes 0..31; 6: hashed ~50 %
template <int PAT>
__device__ __forceinline__ bool lane_on(int lane, int it) {
  if (PAT == 0) return true;
  if (PAT == 1) return lane == 63;
  if (PAT == 2) return lane < 16;
  if (PAT == 3) return (lane & 7) == 0;
  if (PAT == 4) return (((unsigned)(lane * 2654435761u + it * 40503u) >> 11) & 7) == 0;
  if (PAT == 5) return lane < 32;
  return (((unsigned)(lane * 2654435761u + it * 40503u) >> 11) & 1) == 0;
}

template <int PAT, int WIDE, int U>
__global__ __launch_bounds__(256) void k(const float* __restrict__ a, float* __restrict__ out, int row_floats, int rows, int iters) {
  float s = 0;
  const int lane = threadIdx.x & 63;
  int r = (blockIdx.x * 7 + (threadIdx.x >> 6)) % rows;
  for (int it = 0; it < iters; ++it) {
    const bool on = lane_on<PAT>(lane, it);
    if (on) {  // one masked region per iteration: the U loads are in flight together
      f4u v[U];
      float d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* row = a + (size_t)((r + u * 3) & (rows - 1)) * row_floats;
        if (WIDE) v[u] = *(const f4u*)(row + 4 * lane); else d[u] = row[lane];
      }
      __builtin_amdgcn_sched_barrier(0);  // all U loads issued before the first is consumed
#pragma unroll
      for (int u = 0; u < U; ++u) s += WIDE ? v[u].x + v[u].y + v[u].z + v[u].w : d[u];
    }
    r = (r + U * 3 + 1) & (rows - 1);
  }
  if (s == 1234.5f) out[0] = s;
}

template <int PAT, int WIDE, int U = 16>
int run(const char* name, const float* a, float* o, int row_floats, int rows) {
  const int iters = 400, blocks = 512;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<PAT, WIDE, U>), dim3(blocks), dim3(256), 0, 0, a, o, row_floats, rows, 20);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<PAT, WIDE, U>), dim3(blocks), dim3(256), 0, 0, a, o, row_floats, rows, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double winstr = (double)blocks * 4 * iters * U;
  printf("%-8s %-34s %.3f ms  %.1f clk/wave-instr/CU (2.4 GHz assumed)\n", WIDE ? "dwordx4" : "dword", name, ms, ms * 1e-3 * 2.4e9 * 256 / winstr);
  return 0;
}

// DPP semantics: out[i] = what lane i sees
__global__ void k_dpp(int* out) {
  const int lane = threadIdx.x;
  int v = 1000 + lane;
  // wave_shl:1 = 0x130, wave_rol:1 = 0x134, wave_shr:1 = 0x138, wave_ror:1 = 0x13c; row_shl:1 = 0x101, row_shr:1 = 0x111
  out[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
  out[192 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xf, 0xf, false);
  // wave_shl:1 with half the lanes disabled: do active lanes read disabled neighbours?
  int w = -2;
  if (lane & 1) w = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);
  out[256 + lane] = w;
}

template <int DPP>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int it = 0; it < iters; ++it) {
#define STEP(x) if (DPP) x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false)) + 1.0f; else x = x * 1.0001f + 1.0f;
    STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
  const int row_floats = 2048, rows = 1024;  // 8 MB: L2 resident
  float *a, *o;
  CK(hipMalloc(&a, (size_t)row_floats * rows * 4 + 4096)); CK(hipMalloc(&o, 4 * 256 * 2048));
  CK(hipMemset(a, 0, (size_t)row_floats * rows * 4 + 4096));
#define ALL(W) \
  run<0, W>("all 64 lanes", a, o, row_floats, rows); \
  run<1, W>("lane 63 only", a, o, row_floats, rows); \
  run<2, W>("lanes 0..15", a, o, row_floats, rows); \
  run<5, W>("lanes 0..31", a, o, row_floats, rows); \
  run<3, W>("every 8th lane", a, o, row_floats, rows); \
  run<4, W>("hashed ~12 % of lanes", a, o, row_floats, rows); \
  run<6, W>("hashed ~50 % of lanes", a, o, row_floats, rows);
  ALL(1)
  ALL(0)
  int* d; CK(hipMalloc(&d, 4 * 320));
  hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d);
  int hsts[320]; CK(hipMemcpy(hsts, d, sizeof(hsts), hipMemcpyDeviceToHost));
  const char* nm[5] = {"wave_shl:1", "wave_rol:1", "wave_shr:1", "row_shl:1", "wave_shl:1, odd lanes active"};
  for (int t = 0; t < 5; ++t) {
    printf("%s: lane0 %d lane1 %d lane15 %d lane16 %d lane62 %d lane63 %d\n", nm[t], hsts[t * 64], hsts[t * 64 + 1], hsts[t * 64 + 15], hsts[t * 64 + 16], hsts[t * 64 + 62], hsts[t * 64 + 63]);
  }
  for (int dpp = 0; dpp < 2; ++dpp) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 2048;
    if (dpp) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, o, 10); else hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, o, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    if (dpp) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, o, iters); else hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, o, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: blocks * 4 waves / (256 CUs * 4 SIMDs) waves, each iters * 8 steps (DPP: mov_dpp + add = 2 instr; else fma = 1)
    const double winstr = (double)blocks * 4 / 1024 * iters * 8;
    printf("%s: %.3f ms, %.2f clk per step per SIMD (2.4 GHz assumed)\n", dpp ? "mov_dpp wave_shl:1 + add" : "fma", ms, ms * 1e-3 * 2.4e9 / winstr);
  }
  return 0;
}
