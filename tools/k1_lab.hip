// Laboratory for the K1 inner loop (not part of the library): times variants of the Hamming top-2 loop on a
// synthetic bank so that the structure of the production kernel can be chosen from measurements.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ void top2_push(uint32_t &b0, uint32_t &b1, uint32_t key) { b1 = umed3(b0, b1, key); b0 = min(b0, key); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(4))) *cptr4;

// MODE 0: query slice in LDS, broadcast ds_read_b128 (production structure)
// MODE 1: query rows through the scalar cache (s_load_dwordx16), no LDS
// MODE 2: LDS with explicit register double buffering of the query row
// ACC2: two popcount accumulators per bank row (breaks the dependent v_bcnt chain)
template <int MODE, int R, int WAVES, bool ACC2>
__global__ __launch_bounds__(WAVES * 64) void k_lab(const uint4 *__restrict__ bank, uint32_t n_blocks,
                                                    const uint4 *__restrict__ qdesc, uint32_t nq, uint32_t lds_rows,
                                                    uint2 *__restrict__ part) {
  extern __shared__ uint4 qs[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t w0 = (blockIdx.x * WAVES + wave) * R;
  uint32_t b[R][16], best0[R], best1[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t blk = w0 + r;
    best0[r] = best1[r] = 0xFFFFFFFFu;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (blk < n_blocks) v = bank[((uint64_t)blk * 4 + c) * 64 + lane];
      b[r][4 * c] = v.x; b[r][4 * c + 1] = v.y; b[r][4 * c + 2] = v.z; b[r][4 * c + 3] = v.w;
    }
  }
  auto pairs = [&](const uint32_t *q, uint32_t j) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t acc;
      if (ACC2) {
        uint32_t a0 = 0, a1 = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) { a0 += __builtin_popcount(b[r][k] ^ q[k]); a1 += __builtin_popcount(b[r][k + 1] ^ q[k + 1]); }
        acc = a0 + a1;
      } else {
        acc = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += __builtin_popcount(b[r][k] ^ q[k]);
      }
      top2_push(best0[r], best1[r], (acc << 16) | j);
    }
  };
  if (MODE == 1) {
    cptr4 qc = (cptr4)qdesc;
#pragma unroll 2
    for (uint32_t j = 0; j < nq; ++j) {
      const u32x4 q0 = qc[j * 4 + 0], q1 = qc[j * 4 + 1], q2 = qc[j * 4 + 2], q3 = qc[j * 4 + 3];
      const uint32_t q[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      pairs(q, j);
    }
  } else {
    for (uint32_t j0 = 0; j0 < nq; j0 += lds_rows) {
      const uint32_t cnt = min(lds_rows, nq - j0);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < cnt * 4; i += WAVES * 64) qs[i] = qdesc[(uint64_t)j0 * 4 + i];
      __syncthreads();
      if (MODE == 0) {
#pragma unroll 2
        for (uint32_t jj = 0; jj < cnt; ++jj) {
          const uint4 q0 = qs[jj * 4 + 0], q1 = qs[jj * 4 + 1], q2 = qs[jj * 4 + 2], q3 = qs[jj * 4 + 3];
          const uint32_t q[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
          pairs(q, j0 + jj);
        }
      } else {
        uint4 n0 = qs[0], n1 = qs[1], n2 = qs[2], n3 = qs[3];
        for (uint32_t jj = 0; jj < cnt; ++jj) {
          const uint4 q0 = n0, q1 = n1, q2 = n2, q3 = n3;
          const uint32_t nx = (jj + 1 < cnt) ? jj + 1 : jj;
          n0 = qs[nx * 4 + 0]; n1 = qs[nx * 4 + 1]; n2 = qs[nx * 4 + 2]; n3 = qs[nx * 4 + 3];
          const uint32_t q[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
          pairs(q, j0 + jj);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (w0 + r < n_blocks) part[((uint64_t)(w0 + r)) * 64 + lane] = make_uint2(best0[r], best1[r]);
}

static std::vector<uint2> g_ref;
template <int MODE, int R, int WAVES, bool ACC2>
void run(const char *name, const uint4 *d_bank, uint32_t n_blocks, const uint4 *d_q, uint32_t nq, uint32_t lds_rows, uint2 *d_part) {
  auto kern = k_lab<MODE, R, WAVES, ACC2>;
  const size_t lds = MODE == 1 ? 0 : (size_t)std::min(lds_rows, nq) * 64;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((n_blocks + WAVES * R - 1) / (WAVES * R));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, 0, d_bank, n_blocks, d_q, nq, lds_rows, d_part);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  std::vector<uint2> h((size_t)n_blocks * 64);
  CK(hipMemcpy(h.data(), d_part, h.size() * 8, hipMemcpyDeviceToHost));
  bool same = true;
  if (g_ref.empty()) g_ref = h; else for (size_t i = 0; i < h.size(); ++i) if (h[i].x != g_ref[i].x || h[i].y != g_ref[i].y) { same = false; break; }
  const double pair_waves_per_simd = (double)n_blocks * nq / 1024.0;
  printf("%-28s R=%d W=%2d lds=%4u  %.3f ms  %.1f cyc/pair-wave@2.2GHz  %s\n", name, R, WAVES, lds_rows, best,
         best * 1e-3 * 2.2e9 / pair_waves_per_simd, same ? "ok" : "MISMATCH");
}

int main() {
  const uint32_t n_rows = 2000000, nq = 2000;
  const uint32_t n_blocks = (n_rows + 63) / 64;
  std::vector<uint32_t> hb((size_t)n_blocks * 64 * 16), hq((size_t)nq * 16);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (auto &v : hb) v = rnd();
  for (auto &v : hq) v = rnd();
  uint4 *d_bank, *d_q; uint2 *d_part;
  CK(hipMalloc(&d_bank, hb.size() * 4)); CK(hipMalloc(&d_q, hq.size() * 4 + 4096)); CK(hipMalloc(&d_part, (size_t)n_blocks * 64 * 8));
  CK(hipMemcpy(d_bank, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  run<0, 1, 16, false>("lds", d_bank, n_blocks, d_q, nq, 1024, d_part);
  run<0, 1, 4, false>("lds", d_bank, n_blocks, d_q, nq, 512, d_part);
  run<0, 2, 16, false>("lds", d_bank, n_blocks, d_q, nq, 1024, d_part);
  run<0, 1, 16, true>("lds acc2", d_bank, n_blocks, d_q, nq, 1024, d_part);
  run<0, 2, 16, true>("lds acc2", d_bank, n_blocks, d_q, nq, 1024, d_part);
  run<0, 1, 4, true>("lds acc2", d_bank, n_blocks, d_q, nq, 512, d_part);
  run<2, 1, 16, false>("lds prefetch", d_bank, n_blocks, d_q, nq, 1024, d_part);
  run<2, 2, 16, false>("lds prefetch", d_bank, n_blocks, d_q, nq, 1024, d_part);
  run<2, 1, 4, false>("lds prefetch", d_bank, n_blocks, d_q, nq, 512, d_part);
  run<2, 2, 16, true>("lds prefetch acc2", d_bank, n_blocks, d_q, nq, 1024, d_part);
  run<1, 1, 4, false>("smem", d_bank, n_blocks, d_q, nq, 0, d_part);
  run<1, 2, 4, false>("smem", d_bank, n_blocks, d_q, nq, 0, d_part);
  run<1, 4, 4, false>("smem", d_bank, n_blocks, d_q, nq, 0, d_part);
  run<1, 2, 8, false>("smem", d_bank, n_blocks, d_q, nq, 0, d_part);
  run<1, 2, 4, true>("smem acc2", d_bank, n_blocks, d_q, nq, 0, d_part);
  run<1, 4, 4, true>("smem acc2", d_bank, n_blocks, d_q, nq, 0, d_part);
  run<1, 1, 4, true>("smem acc2", d_bank, n_blocks, d_q, nq, 0, d_part);
  return 0;
}
