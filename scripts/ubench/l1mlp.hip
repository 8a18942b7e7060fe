// Micro-benchmark: how many 128-byte lines can one CU keep in flight from its vector L1?
// A workgroup streams a private, never re-read region with U independent 16-byte loads per lane in flight (every wave
// load = 8 whole lines).  Swept over U and waves per CU, for a footprint that lives in HBM (4 GiB) and one that stays in
// the XCD's L2 (2 MiB per CU group).  From bytes / time and the latency of a dependent chain of the same loads:
//   lines in flight per CU (Little) = (B/clk/CU) x (latency in clk) / 128.
// DESIGN.md 4.5 reads k_flow_iter3's counters against the plateau this prints.   hipcc -O3 --offload-arch=gfx950 l1mlp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_stream(const f4* __restrict__ a, float* out, size_t per_wg_vec, int iters) {
  // workgroup w reads its own slice [w * per_wg_vec, +per_wg_vec) in steps of 256 * U vectors
  const f4* p = a + (size_t)blockIdx.x * per_wg_vec + threadIdx.x;
  float s = 0;
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[off + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    off += (size_t)U * 256;
    if (off + (size_t)U * 256 > per_wg_vec) off = 0;
  }
  if (s == 1234.5f) out[0] = s;
}

// dependent chain: each load's address comes from the previous one's data (one lane per wave active)
__global__ void k_chase(const unsigned* __restrict__ a, unsigned* out, int steps) {
  unsigned i = threadIdx.x + blockIdx.x * 977u;
  for (int s = 0; s < steps; ++s) i = a[i];
  out[blockIdx.x] = i;
}

template <int U>
int run(const f4* a, float* o, size_t total_vec, int wg_per_cu, const char* what, double lat_clk, double ghz) {
  const int blocks = 256 * wg_per_cu;
  const size_t per = total_vec / blocks / (256 * U) * (256 * U);
  const int iters = 3000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_stream<U>), dim3(blocks), dim3(256), 0, 0, a, o, per, 50);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_stream<U>), dim3(blocks), dim3(256), 0, 0, a, o, per, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * iters * U * 256 * 16;
  const double bpc = bytes / (ms * 1e-3 * ghz * 1e9 * 256);
  printf("%-4s %d waves/CU x %2d loads in flight: %6.2f TB/s  %5.1f B/clk/CU  -> %5.1f lines in flight per CU at %4.0f clk\n", what, 4 * wg_per_cu, U,
         bytes / (ms * 1e-3) / 1e12, bpc, bpc * lat_clk / 128, lat_clk);
  return 0;
}

int main() {
  const double ghz = 2.4;
  const size_t hbm_bytes = (size_t)4 << 30, l2_bytes = (size_t)24 << 20;   // 24 MiB < 8 x 4 MiB of L2
  f4* a; float* o; CK(hipMalloc(&a, hbm_bytes)); CK(hipMalloc(&o, 1 << 20));
  CK(hipMemset(a, 0, hbm_bytes));
  // latency of a dependent load chain: random permutation over the footprint (one 4-byte index per 128-byte line)
  double lat[2];
  for (int f = 0; f < 2; ++f) {
    const size_t bytes = f ? l2_bytes / 8 : (size_t)1 << 30;   // L2 case: one XCD's share
    const size_t lines = bytes / 128;
    unsigned* h = (unsigned*)malloc(bytes);
    for (size_t i = 0; i < bytes / 4; ++i) h[i] = 0;
    // a single cycle through all lines (Sattolo), index stored in the first dword of each line
    unsigned* perm = (unsigned*)malloc(lines * 4);
    for (size_t i = 0; i < lines; ++i) perm[i] = (unsigned)i;
    srand(1);
    for (size_t i = lines - 1; i > 0; --i) { size_t j = (size_t)rand() % i; unsigned t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    for (size_t i = 0; i < lines; ++i) h[(size_t)perm[i] * 32] = perm[(i + 1) % lines] * 32;
    CK(hipMemcpy(a, h, bytes, hipMemcpyHostToDevice));
    unsigned* out = (unsigned*)o;
    const int steps = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, (const unsigned*)a, out, f ? (int)lines : 2000);   // warm (L2 case: touch every line)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, (const unsigned*)a, out, steps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    lat[f] = ms * 1e-3 * ghz * 1e9 / steps;
    printf("dependent-load latency, %s footprint: %.0f ns = %.0f clk at %.1f GHz (idle chip)\n", f ? "L2-resident" : "HBM", ms * 1e6 / steps, lat[f], ghz);
    free(h); free(perm);
  }
  CK(hipMemset(a, 0, hbm_bytes));
  for (int wg = 1; wg <= 4; wg *= 2) {
    run<1>(a, o, hbm_bytes / 16, wg, "HBM", lat[0], ghz); run<2>(a, o, hbm_bytes / 16, wg, "HBM", lat[0], ghz);
    run<4>(a, o, hbm_bytes / 16, wg, "HBM", lat[0], ghz); run<8>(a, o, hbm_bytes / 16, wg, "HBM", lat[0], ghz);
    run<16>(a, o, hbm_bytes / 16, wg, "HBM", lat[0], ghz);
  }
  for (int wg = 1; wg <= 4; wg *= 2) {
    run<1>(a, o, l2_bytes / 16, wg, "L2", lat[1], ghz); run<2>(a, o, l2_bytes / 16, wg, "L2", lat[1], ghz);
    run<4>(a, o, l2_bytes / 16, wg, "L2", lat[1], ghz); run<8>(a, o, l2_bytes / 16, wg, "L2", lat[1], ghz);
    run<16>(a, o, l2_bytes / 16, wg, "L2", lat[1], ghz);
  }
  return 0;
}
