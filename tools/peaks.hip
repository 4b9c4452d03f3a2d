// Measures the two ceilings K1 is priced against on this GPU:
//   (a) VALU: register-resident v_xor_b32 + v_bcnt_u32_b32 (accumulating) chains, lane-ops/s
//   (b) HBM : streaming uint4 read (sum-reduced so it is not elided) and uint4 copy, GB/s
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/peaks tools/peaks.hip ; prints one JSON line.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int CHAINS>
__global__ __launch_bounds__(256) void k_valu(uint32_t *out, uint32_t seed, int iters) {
  uint32_t a[CHAINS], acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { a[c] = seed * (threadIdx.x + 1) + c * 0x9E3779B9u; acc[c] = 0; }
  uint32_t q = seed ^ 0x12345u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] += __builtin_popcount(a[c] ^ q);
      q += 0x9E3779B9u;  // keeps the xor from being hoisted: 1 extra v_add per 8 useful ops
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ in, size_t n, uint32_t *out) {
  uint32_t s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = in[i];
    s += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (s == 0x12345678u) out[0] = s;
}

__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ outp, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) outp[i] = in[i];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t *d_out; CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4 * 4));
  double best_valu = 0; int best_waves = 0;
  for (int wg_per_cu : {4, 8}) {
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_valu<4>, dim3(cus * wg_per_cu), dim3(256), 0, 0, d_out, 7u + rep, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      // per lane per inner k: 4 xor + 4 bcnt VALU ops (q is wave-uniform and lives in SGPRs)
      double ops = (double)cus * wg_per_cu * 256 * iters * 16 * 8;
      double tops = ops / (ms * 1e-3) / 1e12;
      if (tops > best_valu) { best_valu = tops; best_waves = wg_per_cu * 4; }
    }
  }
  const size_t bytes = 2ull << 30; const size_t n = bytes / 16;
  uint4 *d_a, *d_b; CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes)); CK(hipMemset(d_a, 1, bytes)); CK(hipMemset(d_b, 2, bytes));
  double best_read = 0, best_copy = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_read, dim3(cus * 8), dim3(256), 0, 0, d_a, n, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double g = bytes / (ms * 1e-3) / 1e9; if (g > best_read) best_read = g;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, d_a, d_b, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    g = 2.0 * bytes / (ms * 1e-3) / 1e9; if (g > best_copy) best_copy = g;
  }
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"valu_xor_bcnt_tops\": %.2f, \"valu_waves_per_cu\": %d, "
         "\"hbm_read_gbs\": %.1f, \"hbm_copy_gbs\": %.1f}\n",
         p.name, cus, p.clockRate / 1000, best_valu, best_waves, best_read, best_copy);
  return 0;
}
