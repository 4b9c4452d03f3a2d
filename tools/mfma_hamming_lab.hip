// LAB (not adopted: north_star reserves MFMA for J^T J) -- would the matrix cores beat the popcount
// kernel at exact Hamming 2-NN?  Hamming(a, b) over the 486 valid bits = (486 - <a', b'>) / 2 with a', b' the bits as
// +-1 bytes (the query's padding bits as 0), so a 32 x 32 tile of distances is 16 x v_mfma_i32_32x32x32_i8 (K = 512)
// and the running top-2 per bank row works on the accumulators: lane (l & 31) holds the bank row's column, its 16
// registers are 16 query rows.  Per pair that leaves five VALU lane-ops -- key = ((486 << 15) | q) - (dot << 15)
// (shift + subtract), then the running top-2 (min, max, min) -- five lane-ops against 22 in k_hamming_screen.
//   functional: one wave per 32 bank rows, operands expanded from the bit rows in the kernel; (best, second) keys of every
//               bank row compared with the CPU's exact popcount top-2 (lower query index wins ties)
//   ceiling:    the same inner loop (16 MFMAs + the 48-instruction epilogue per 32 x 32 tile) on register operands, 1 / 2
//               / 4 waves per SIMD on every SIMD: pairs/s when operand delivery is free -- what a finished kernel could
//               at most reach; delivery (LDS-staged query tiles, a 2 x 2 register block per wave) is not written
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_hamming_lab tools/mfma_hamming_lab.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// 4 bits -> 4 bytes: bit set -> +1, clear -> -1
__device__ __forceinline__ uint32_t pm1_bytes(uint32_t nib) {
  const uint32_t spread = ((nib & 0xFu) * 0x00204081u) & 0x01010101u;  // bit i -> byte i (0 / 1)
  return 0xFFFFFFFFu ^ (spread * 0xFEu);                               // 1 -> 0x01, 0 -> 0xFF
}
// 16 bits -> 16 bytes of +-1; valid: mask of the bits that count (others -> 0)
__device__ __forceinline__ v4i expand16(uint32_t bits16, uint32_t valid16) {
  v4i r;
  for (int g = 0; g < 4; ++g) {
    const uint32_t b = pm1_bytes(bits16 >> (4 * g));
    const uint32_t m = (((valid16 >> (4 * g)) & 0xFu) * 0x00204081u) & 0x01010101u;
    r[g] = (int)(b & (m * 0xFFu));
  }
  return r;
}

// running top-2 on the key: b1 = med3(b0, b1, key); b0 = min(b0, key)
__device__ __forceinline__ void push(uint32_t &b0, uint32_t &b1, uint32_t key) {
  const uint32_t lo = min(b0, key), hi = max(b0, key);
  b1 = min(b1, hi);
  b0 = lo;
}

// bank [n_bank x 16 dwords], query [n_q x 16 dwords] (rows of 64 bytes, padding zero) -> out[n_bank] = (best, second) keys
__global__ __launch_bounds__(64) void k_functional(const uint32_t *__restrict__ bank, const uint32_t *__restrict__ query,
                                                   int n_q, uint2 *__restrict__ out) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int brow = blockIdx.x * 32 + r;
  v4i B[16];
  for (int s = 0; s < 16; ++s) {
    const uint32_t w = bank[(size_t)brow * 16 + s];
    B[s] = expand16((w >> (16 * h)) & 0xFFFFu, 0xFFFFu);  // the bank's padding bits may be anything: the query's are 0
  }
  uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
  for (int q0 = 0; q0 < n_q; q0 += 32) {
    v16i acc = {0};
    const int qrow = min(q0 + r, n_q - 1);
    for (int s = 0; s < 16; ++s) {
      const uint32_t w = query[(size_t)qrow * 16 + s];
      // valid bits: 486 = 15 dwords + 6 bits
      const uint32_t vmask = s < 15 ? 0xFFFFFFFFu : 0x3Fu;
      const v4i A = expand16((w >> (16 * h)) & 0xFFFFu, (vmask >> (16 * h)) & 0xFFFFu);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[s], acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) {
      const int m = (reg & 3) + 8 * (reg >> 2) + 4 * h;  // the query row of this accumulator
      const int q = q0 + m;
      if (q < n_q) {
        const uint32_t key = (uint32_t)(((486 << 15) | q) - (acc[reg] << 15));
        push(b0, b1, key);
      }
    }
  }
  // lanes l and l + 32 hold the same bank row: merge
  const uint32_t o0 = __shfl_xor(b0, 32, 64), o1 = __shfl_xor(b1, 32, 64);
  push(b0, b1, o0);
  push(b0, b1, o1);
  if (h == 0) out[brow] = make_uint2(b0, b1);
}

// the inner loop on register operands: `tiles` 32 x 32 tiles per wave
template <int EPI>
__global__ __launch_bounds__(256) void k_ceiling(int tiles, uint32_t seed, uint2 *__restrict__ sink) {
  v4i A[4], B[4];
  for (int i = 0; i < 4; ++i)
    for (int g = 0; g < 4; ++g) {
      A[i][g] = (int)pm1_bytes(seed * 2654435761u + threadIdx.x * 97u + i * 13u + g);
      B[i][g] = (int)pm1_bytes(seed * 40503u + threadIdx.x * 31u + i * 7u + g);
    }
  uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
  int qbase = (486 << 15);
  for (int t = 0; t < tiles; ++t) {
    v16i acc = {0};
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s & 3], B[(s >> 2) & 3], acc, 0, 0, 0);
    if (EPI) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const uint32_t key = (uint32_t)((qbase + reg) - (acc[reg] << 15));
        push(b0, b1, key);
      }
    } else {
      b0 ^= (uint32_t)acc[0];
    }
    qbase += 32;
    A[t & 3][0] ^= (int)b0 & 0x01010101;  // (keeps the loop from being hoisted; one VALU per tile)
  }
  if (b0 == 0x12345678u) sink[0] = make_uint2(b0, b1);
}

static uint32_t rnd(uint64_t &s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(s >> 33);
}

int main() {
  // ---- functional check ----
  const int n_bank = 4096, n_q = 1000;
  std::vector<uint32_t> bank((size_t)n_bank * 16), query((size_t)n_q * 16);
  uint64_t s = 7;
  auto fill = [&](std::vector<uint32_t> &v, int rows) {
    for (int i = 0; i < rows; ++i)
      for (int k = 0; k < 16; ++k) {
        uint32_t w = (rnd(s) << 16) ^ rnd(s);
        if (k == 15) w &= 0x3Fu;  // 486 valid bits
        v[(size_t)i * 16 + k] = w;
      }
  };
  fill(bank, n_bank);
  fill(query, n_q);
  for (int i = 0; i < n_bank; i += 7) {  // planted near-duplicates and exact ties
    const int j = (i * 31) % n_q;
    memcpy(&bank[(size_t)i * 16], &query[(size_t)j * 16], 64);
    bank[(size_t)i * 16 + (i % 15)] ^= (1u << (i % 29)) | (1u << ((i + 5) % 31));
    if (i % 21 == 0 && j + 1 < n_q) memcpy(&query[(size_t)(j + 1) * 16], &query[(size_t)j * 16], 64);  // a tie: lower index wins
  }
  uint32_t *d_bank, *d_query;
  uint2 *d_out;
  CK(hipMalloc(&d_bank, bank.size() * 4));
  CK(hipMalloc(&d_query, query.size() * 4));
  CK(hipMalloc(&d_out, n_bank * sizeof(uint2)));
  CK(hipMemcpy(d_bank, bank.data(), bank.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_query, query.data(), query.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_functional, dim3(n_bank / 32), dim3(64), 0, 0, d_bank, d_query, n_q, d_out);
  CK(hipDeviceSynchronize());
  std::vector<uint2> out(n_bank);
  CK(hipMemcpy(out.data(), d_out, n_bank * sizeof(uint2), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n_bank; ++i) {
    uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
    for (int j = 0; j < n_q; ++j) {
      int d = 0;
      for (int k = 0; k < 16; ++k) d += __builtin_popcount(bank[(size_t)i * 16 + k] ^ query[(size_t)j * 16 + k]);
      const uint32_t key = ((uint32_t)d << 16) | (uint32_t)j;
      const uint32_t lo = key < b0 ? key : b0, hi = key < b0 ? b0 : key;
      b1 = hi < b1 ? hi : b1;
      b0 = lo;
    }
    if (out[i].x != b0 || out[i].y != b1) {
      if (bad < 5) printf("row %d: mfma (%08x, %08x) popcount (%08x, %08x)\n", i, out[i].x, out[i].y, b0, b1);
      ++bad;
    }
  }
  printf("{\"functional\": \"%d bank rows x %d query rows, top-2 keys by i8 MFMA vs popcount\", \"rows_differing\": %d}\n",
         n_bank, n_q, bad);
  // ---- ceiling ----
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint2 *d_sink;
  CK(hipMalloc(&d_sink, 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int epi = 0; epi < 2; ++epi)
    for (int wps = 1; wps <= 4; wps *= 2) {
      const int tiles = 4000;
      const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD; wps blocks per CU
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (epi)
          hipLaunchKernelGGL(k_ceiling<1>, dim3(blocks), dim3(256), 0, 0, tiles, 11u, d_sink);
        else
          hipLaunchKernelGGL(k_ceiling<0>, dim3(blocks), dim3(256), 0, 0, tiles, 11u, d_sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
      }
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double pairs = (double)blocks * 4 * tiles * 1024.0;
      printf("{\"ceiling\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"tera_pairs_per_s\": %.3f, "
             "\"i8_tera_ops_per_s\": %.1f}\n",
             epi ? "16 MFMA + top-2 epilogue (3 VALU per pair) per 32x32 tile" : "16 MFMA per tile, no epilogue", wps, ms,
             pairs / (ms * 1e-3) / 1e12, pairs * 1024.0 / (ms * 1e-3) / 1e12);
    }
  return bad ? 1 : 0;
}
