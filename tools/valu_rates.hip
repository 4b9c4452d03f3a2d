// Per-instruction issue cost of the integer VALU ops K1 is made of, on gfx950.
// One kernel per instruction mix; each wave runs a long unrolled loop of independent chains and stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around it.  Prints cycles per wave-instruction at
// 1, 2, 4 and 8 waves per SIMD and the clock the chip held.  The grid is persistent and exactly balanced: every
// workgroup (4 waves = one per SIMD) asks for 160 KiB / w of LDS, so a CU takes exactly w of them and the grid is
// w x CUs -- in-kernel stamps and wall time then describe the same thing (round 1 measured at <= 4 waves per SIMD on
// whatever placement the dispatcher chose, and its wall-time column under-read the ceiling).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MIX>
__global__ __launch_bounds__(256) void k_mix(uint32_t *out, unsigned long long *stamps, uint32_t seed, int iters) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  uint32_t q = seed ^ 0x5555u;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (MIX == 0) {  // v_xor_b32 only (8 independent chains)
      REP8(asm volatile("v_xor_b32 %0, %8, %0\n\tv_xor_b32 %1, %8, %1\n\tv_xor_b32 %2, %8, %2\n\tv_xor_b32 %3, %8, %3\n\t"
                        "v_xor_b32 %4, %8, %4\n\tv_xor_b32 %5, %8, %5\n\tv_xor_b32 %6, %8, %6\n\tv_xor_b32 %7, %8, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 1) {  // v_bcnt_u32_b32 only
      REP8(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n\tv_bcnt_u32_b32 %1, %8, %1\n\tv_bcnt_u32_b32 %2, %8, %2\n\tv_bcnt_u32_b32 %3, %8, %3\n\t"
                        "v_bcnt_u32_b32 %4, %8, %4\n\tv_bcnt_u32_b32 %5, %8, %5\n\tv_bcnt_u32_b32 %6, %8, %6\n\tv_bcnt_u32_b32 %7, %8, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 2) {  // xor then bcnt (the K1 pair), 4 chains
      REP8(asm volatile("v_xor_b32 %4, %8, %4\n\tv_bcnt_u32_b32 %0, %4, %0\n\tv_xor_b32 %5, %8, %5\n\tv_bcnt_u32_b32 %1, %5, %1\n\t"
                        "v_xor_b32 %6, %8, %6\n\tv_bcnt_u32_b32 %2, %6, %2\n\tv_xor_b32 %7, %8, %7\n\tv_bcnt_u32_b32 %3, %7, %3"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 10) {  // xor + bcnt, software pipelined: a bcnt never reads the xor issued right before it
      REP8(asm volatile("v_xor_b32 %4, %8, %4\n\tv_xor_b32 %5, %8, %5\n\tv_bcnt_u32_b32 %0, %6, %0\n\tv_bcnt_u32_b32 %1, %7, %1\n\t"
                        "v_xor_b32 %6, %8, %6\n\tv_xor_b32 %7, %8, %7\n\tv_bcnt_u32_b32 %2, %4, %2\n\tv_bcnt_u32_b32 %3, %5, %3"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 11) {  // 6 xors, then 6 bcnt (two accumulators): does grouping by instruction class help the issue rate?
      REP8(asm volatile("v_xor_b32 %2, %8, %2\n\tv_xor_b32 %3, %8, %3\n\tv_xor_b32 %4, %8, %4\n\tv_xor_b32 %5, %8, %5\n\t"
                        "v_xor_b32 %6, %8, %6\n\tv_xor_b32 %7, %8, %7\n\t"
                        "v_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %0, %4, %0\n\t"
                        "v_bcnt_u32_b32 %1, %5, %1\n\tv_bcnt_u32_b32 %0, %6, %0\n\tv_bcnt_u32_b32 %1, %7, %1"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 3) {  // v_med3_u32
      REP8(asm volatile("v_med3_u32 %0, %8, %0, %1\n\tv_med3_u32 %1, %8, %1, %2\n\tv_med3_u32 %2, %8, %2, %3\n\tv_med3_u32 %3, %8, %3, %4\n\t"
                        "v_med3_u32 %4, %8, %4, %5\n\tv_med3_u32 %5, %8, %5, %6\n\tv_med3_u32 %6, %8, %6, %7\n\tv_med3_u32 %7, %8, %7, %0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 4) {  // v_min_u32 (VOP2)
      REP8(asm volatile("v_min_u32 %0, %8, %0\n\tv_min_u32 %1, %8, %1\n\tv_min_u32 %2, %8, %2\n\tv_min_u32 %3, %8, %3\n\t"
                        "v_min_u32 %4, %8, %4\n\tv_min_u32 %5, %8, %5\n\tv_min_u32 %6, %8, %6\n\tv_min_u32 %7, %8, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 5) {  // v_lshl_or_b32 (VOP3)
      REP8(asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\t"
                        "v_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 6) {  // xor with SGPR operand + bcnt
      uint32_t sq = __builtin_amdgcn_readfirstlane(q);
      REP8(asm volatile("v_xor_b32 %4, %8, %4\n\tv_bcnt_u32_b32 %0, %4, %0\n\tv_xor_b32 %5, %8, %5\n\tv_bcnt_u32_b32 %1, %5, %1\n\t"
                        "v_xor_b32 %6, %8, %6\n\tv_bcnt_u32_b32 %2, %6, %2\n\tv_xor_b32 %7, %8, %7\n\tv_bcnt_u32_b32 %3, %7, %3"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sq));)
    } else if (MIX == 7) {  // v_add_u32 VOP2
      REP8(asm volatile("v_add_u32 %0, %8, %0\n\tv_add_u32 %1, %8, %1\n\tv_add_u32 %2, %8, %2\n\tv_add_u32 %3, %8, %3\n\t"
                        "v_add_u32 %4, %8, %4\n\tv_add_u32 %5, %8, %5\n\tv_add_u32 %6, %8, %6\n\tv_add_u32 %7, %8, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 8) {  // v_and_or_b32 (VOP3, 3 operands)
      REP8(asm volatile("v_and_or_b32 %0, %0, %8, %1\n\tv_and_or_b32 %1, %1, %8, %2\n\tv_and_or_b32 %2, %2, %8, %3\n\tv_and_or_b32 %3, %3, %8, %4\n\t"
                        "v_and_or_b32 %4, %4, %8, %5\n\tv_and_or_b32 %5, %5, %8, %6\n\tv_and_or_b32 %6, %6, %8, %7\n\tv_and_or_b32 %7, %7, %8, %0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    } else if (MIX == 9) {  // v_pk_add_u16 (packed)
      REP8(asm volatile("v_pk_add_u16 %0, %8, %0\n\tv_pk_add_u16 %1, %8, %1\n\tv_pk_add_u16 %2, %8, %2\n\tv_pk_add_u16 %3, %8, %3\n\t"
                        "v_pk_add_u16 %4, %8, %4\n\tv_pk_add_u16 %5, %8, %5\n\tv_pk_add_u16 %6, %8, %6\n\tv_pk_add_u16 %7, %8, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    stamps[4 * w] = t1 - t0;
    stamps[4 * w + 1] = r1 - r0;
    stamps[4 * w + 2] = r0;  // absolute 100 MHz time at the wave's start: do the waves of a launch overlap?
    stamps[4 * w + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32) | __builtin_amdgcn_s_getreg(63492);  // XCC_ID | HW_ID
  }
}

template <int MIX>
void run(const char *name, int cus, uint32_t *d_out, unsigned long long *d_st) {
  const int iters = 2000;
  const int insts_per_iter = MIX == 11 ? 96 : 64;
  for (int wps : {1, 2, 4, 8}) {
    const int wg_per_cu = wps;  // 256 threads = 4 waves = 1 per SIMD
    const int n_waves = cus * wg_per_cu * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = (size_t)(160 * 1024 / wps) & ~(size_t)1023;  // exactly wps workgroups fit a CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mix<MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_mix<MIX>, dim3(cus * wg_per_cu), dim3(256), lds, 0, d_out, d_st, 1u, 200);  // warm
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mix<MIX>, dim3(cus * wg_per_cu), dim3(256), lds, 0, d_out, d_st, 7u, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(4 * (size_t)n_waves);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    unsigned long long s_min = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0;
    std::vector<unsigned long long> cu_keys;
    for (int w = 0; w < n_waves; ++w) {
      cyc.push_back((double)st[4 * w]); clk.push_back((double)st[4 * w] / ((double)st[4 * w + 1] * 10.0));
      const unsigned long long b = st[4 * w + 2], e = b + st[4 * w + 1];
      s_min = std::min(s_min, b); s_max = std::max(s_max, b); e_min = std::min(e_min, e); e_max = std::max(e_max, e);
      const unsigned long long hw = st[4 * w + 3];
      cu_keys.push_back(((hw >> 32) << 16) | ((hw >> 8) & 0x7F) | (((hw >> 4) & 3) << 12));  // xcc | se,sh,cu | simd
    }
    std::sort(cu_keys.begin(), cu_keys.end());
    int simds = 0, w_min = 1 << 30, w_max = 0;
    for (size_t i = 0; i < cu_keys.size();) { size_t j = i; while (j < cu_keys.size() && cu_keys[j] == cu_keys[i]) ++j; ++simds; w_min = std::min(w_min, (int)(j - i)); w_max = std::max(w_max, (int)(j - i)); i = j; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double c_med = cyc[cyc.size() / 2] / ((double)iters * insts_per_iter);
    const double lane_ops = (double)n_waves * 64 * iters * insts_per_iter / (ms * 1e-3) / 1e12;
    printf("{\"mix\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_inst\": %.3f, \"cycles_per_inst_per_simd\": %.3f, \"clock_ghz\": %.3f, \"tera_lane_ops\": %.2f, \"ms\": %.3f, "
           "\"wave_ms_median\": %.3f, \"start_spread_us\": %.1f, \"end_spread_us\": %.1f, \"first_start_to_last_end_ms\": %.3f, \"simds_seen\": %d, \"waves_per_simd_census\": [%d, %d]}\n",
           name, wps, c_med, c_med / wps, clk[clk.size() / 2], lane_ops, ms, cyc[cyc.size() / 2] / (clk[clk.size() / 2] * 1e6),
           (double)(s_max - s_min) / 100.0, (double)(e_max - e_min) / 100.0, (double)(e_max - s_min) / 1e5, simds, w_min, w_max);
  }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  uint32_t *d_out; unsigned long long *d_st;
  CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc(&d_st, (size_t)cus * 8 * 4 * 4 * 8));
  run<0>("v_xor_b32", cus, d_out, d_st);
  run<1>("v_bcnt_u32_b32", cus, d_out, d_st);
  run<2>("xor+bcnt", cus, d_out, d_st);
  run<6>("xor(sgpr)+bcnt", cus, d_out, d_st);
  run<10>("xor+bcnt (pipelined)", cus, d_out, d_st);
  run<11>("6 xor then 6 bcnt", cus, d_out, d_st);
  run<3>("v_med3_u32", cus, d_out, d_st);
  run<4>("v_min_u32", cus, d_out, d_st);
  run<5>("v_lshl_or_b32", cus, d_out, d_st);
  run<7>("v_add_u32", cus, d_out, d_st);
  run<8>("v_and_or_b32", cus, d_out, d_st);
  run<9>("v_pk_add_u16", cus, d_out, d_st);
  return 0;
}
