// K3 / K4 / K5 of the query-localisation path for gfx950.
//
//   K3  k_fmatrix_filter     per surviving (view, query) pair: AC-RANSAC with the 7-point solver
//                            [reference: hulo::geometricMatch -> GeometricFilter_FMatrix_AC, MatchUtils.cpp:372-420]
//   K4  k_emit_candidates /  matchProviderToMatchSet: per query feature keep the landmark whose match has the
//       k_select_candidates  smallest descriptor distance, first one on ties   [SfMDataUtils.cpp:59-125]
//   K5  k_p3p_*              SfM_Localizer::Localize: AC-RANSAC with Kneip P3P on K^-1-normalised points,
//                            then KRt_From_P                                    [localization.cpp:504-509,544-547]
//
// AC-RANSAC is sequential by definition (the first meaningful model redirects sampling to its inliers and
// cuts the iteration budget to the reserved 10 %).  The kernels keep those semantics exactly and get their
// parallelism from speculation: hypotheses whose samples come from the CURRENT index set are evaluated in
// parallel, one workgroup per hypothesis (residuals -> LDS -> bitonic sort -> NFA scan -> block argmin), and
// a selection step replays the sequential rule over them in iteration order, discarding everything after
// the first hypothesis that changes the index set.  Samples come from a counter-based generator
// (Philox4x32-10 keyed by seed/stage/stream/iteration), so iteration i draws the same sample no matter
// which round evaluates it.  All arithmetic is f64, in the operation order of geom_device.h.
#include <string.h>
#include <float.h>

#include <type_traits>

#include "geom_device.h"
#include "sfmloc_internal.h"

namespace sfmloc {
using namespace geom;

namespace {

constexpr int kThreads = 256;

// Diagnostic build only (make EXTRA=-DSFMLOC_STAMPS OUT=../lib/libsfmloc_hip_stamps.so, tools/k35_stamps.py): thread 0
// of a workgroup records wall_clock64() (100 MHz) at stage boundaries of K3 and K5.  Compiles to nothing otherwise.
#ifdef SFMLOC_STAMPS
__device__ unsigned long long g_stamps_p3p[16 * 256 * 8];  // [round][hypothesis][point]
__device__ unsigned long long g_stamps_sel[16 * 8];        // [round][point]
__device__ unsigned long long g_stamps_f[256 * 64];        // [workgroup][event]: tag << 48 | time
#define STAMP_P3P(rnd, b, k)                                                                      \
  do {                                                                                            \
    if (threadIdx.x == 0 && (rnd) < 16 && (b) < 256) g_stamps_p3p[(((rnd)*256) + (b)) * 8 + (k)] = wall_clock64(); \
  } while (0)
#define STAMP_SEL(rnd, k)                                                              \
  do {                                                                                 \
    if (threadIdx.x == 0 && (rnd) < 16) g_stamps_sel[(rnd)*8 + (k)] = wall_clock64(); \
  } while (0)
#define STAMP_F_DECL int stamp_f_n = 0
#define STAMP_F(tag)                                                                                         \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 256 && stamp_f_n < 64)                                              \
      g_stamps_f[blockIdx.x * 64 + stamp_f_n++] = ((unsigned long long)(tag) << 48) | (wall_clock64() & 0xFFFFFFFFFFFFull); \
  } while (0)
#else
#define STAMP_P3P(rnd, b, k) do {} while (0)
#define STAMP_SEL(rnd, k) do {} while (0)
#define STAMP_F_DECL do {} while (0)
#define STAMP_F(tag) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------
// block-wide helpers (256 threads)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
  return ka < kb || (ka == kb && ia < ib);
}

// ascending bitonic sort of (key, idx) pairs, P a power of two
__device__ void bitonic_sort(uint64_t *key, uint32_t *idx, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += kThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i + j;
        const bool up = (i & k) == 0;
        const uint64_t ka = key[i], kb = key[l];
        const uint32_t ia = idx[i], ib = idx[l];
        const bool a_gt_b = pair_less(kb, ib, ka, ia);
        if (a_gt_b == up) {
          key[i] = kb;
          key[l] = ka;
          idx[i] = ib;
          idx[l] = ia;
        }
      }
      __syncthreads();
    }
  }
}

struct NfaBest {
  double nfa;
  int k;
};

// bestNFA of OpenMVG over the sorted residuals key[0..n): min over k in (s, n] with e_k <= max_thr of
//   loge0 + logalpha(e_k) * (k - s) + logc_n[k] + logc_k[k];  first k on ties.  All threads return the result.
__device__ NfaBest best_nfa_block(const uint64_t *key, int n, int s, double max_thr, double logalpha0, double mult,
                                  double loge0, const float *logc_n, const float *logc_k, double *red_nfa,
                                  int *red_k) {
  double lb = pos_inf();
  int lk = 0x7FFFFFFF;
  for (int kk = s + 1 + (int)threadIdx.x; kk <= n; kk += kThreads) {
    const double ek = u2d(key[kk - 1]);
    if (ek <= max_thr) {
      const double logalpha = logalpha0 + mult * det_log10(ek + (double)FLT_EPSILON);
      const double nfa = loge0 + logalpha * (double)(kk - s) + (double)logc_n[kk] + (double)logc_k[kk];
      if (nfa < lb) {
        lb = nfa;
        lk = kk;
      }
    }
  }
  // wave reduce (64 lanes), then across the 4 waves through LDS
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_down(lb, off, 64);
    const int ok = __shfl_down(lk, off, 64);
    if (ob < lb || (ob == lb && ok < lk)) {
      lb = ob;
      lk = ok;
    }
  }
  __syncthreads();  // red_* may still be read by the previous call's consumers
  if ((threadIdx.x & 63) == 0) {
    red_nfa[threadIdx.x >> 6] = lb;
    red_k[threadIdx.x >> 6] = lk;
  }
  __syncthreads();
  NfaBest r{red_nfa[0], red_k[0]};
  for (int w = 1; w < kThreads / 64; ++w) {
    if (red_nfa[w] < r.nfa || (red_nfa[w] == r.nfa && red_k[w] < r.k)) {
      r.nfa = red_nfa[w];
      r.k = red_k[w];
    }
  }
  return r;
}

// bestNFA over one wave's sorted segment; every lane of the wave returns the result
__device__ NfaBest best_nfa_wave(const uint64_t *key, int n, int s, double max_thr, double logalpha0, double mult,
                                 double loge0, const float *logc_n, const float *logc_k) {
  double lb = pos_inf();
  int lk = 0x7FFFFFFF;
  for (int kk = s + 1 + (int)(threadIdx.x & 63); kk <= n; kk += 64) {
    const double ek = u2d(key[kk - 1]);
    if (ek <= max_thr) {
      const double logalpha = logalpha0 + mult * det_log10(ek + (double)FLT_EPSILON);
      const double nfa = loge0 + logalpha * (double)(kk - s) + (double)logc_n[kk] + (double)logc_k[kk];
      if (nfa < lb) {
        lb = nfa;
        lk = kk;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(lb, off, 64);
    const int ok = __shfl_xor(lk, off, 64);
    if (ob < lb || (ob == lb && ok < lk)) {
      lb = ob;
      lk = ok;
    }
  }
  return NfaBest{lb, lk};
}

// ---------------------------------------------------------------------------------------------------
// register-resident wave sort: 64*E (key, idx) pairs, element e = r*64 + lane in register r.  The same bitonic
// network as bitonic_sort: strides >= 64 are compare-exchanges between two registers of one lane, strides < 64
// exchange with lane ^ j through ds_bpermute.  The order by (key, idx) is total, so any correct sort gives the
// same array.
// ---------------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void wave_sort_regs(uint64_t (&key)[E], uint32_t (&idx)[E]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 2; k <= 64 * E; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int jr = j >> 6;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          if ((r & jr) == 0) {
            const int r2 = r | jr;
            const bool up = ((r << 6) & k) == 0;  // k >= 128: only the register index decides the direction
            const bool a_gt_b = pair_less(key[r2], idx[r2], key[r], idx[r]);
            if (a_gt_b == up) {
              const uint64_t tk = key[r];
              key[r] = key[r2];
              key[r2] = tk;
              const uint32_t ti = idx[r];
              idx[r] = idx[r2];
              idx[r2] = ti;
            }
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const uint64_t ok = __shfl_xor(key[r], j, 64);
          const uint32_t oi = __shfl_xor(idx[r], j, 64);
          const bool up = ((((r << 6) | lane) & k) == 0);
          const bool want_min = (((lane & j) == 0) == up);  // lower position of an ascending pair, or upper of a descending one
          if (pair_less(ok, oi, key[r], idx[r]) == want_min) {
            key[r] = ok;
            idx[r] = oi;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same sort without LDS round trips (wave_sort_regs spends most of its time waiting for ds_bpermute: 33 dependent
// exchange stages for 256 elements).  Two changes:
//  * the element index rides in the low kPackBits bits of the 64-bit key, so a compare-exchange is one 64-bit compare
//    and two selects.  The 10 bits it displaces are the last bits of the residual's mantissa; they are parked in LDS
//    (`low`, one u32 per element, private to the wave) and put back after the sort.  Two keys that agree in everything
//    BUT those bits are ordered by index instead of by value; if that was wrong the restored keys do not ascend, which
//    is checked, and such a wave sorts again with wave_sort_regs.  The result is therefore always the order by
//    (key, idx) of the exact sort.
//  * lane ^ j exchanges are DPP moves (j = 1, 2, 4, 8) and gfx950's v_permlane16_swap / v_permlane32_swap (j = 16,
//    32); which lanes keep the minimum at a stage is a compile-time lane mask, applied with s_xor + v_cndmask.
// ---------------------------------------------------------------------------------------------------
constexpr int kPackBits = 10;  // 64 * 16 elements at most
constexpr uint64_t kPackMask = (1ull << kPackBits) - 1;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_perm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, false);  // every lane is written: no `old` operand
}
// the value lane ^ J holds
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, bool odd_half) {
  if constexpr (J == 1) return dpp_perm<0xB1>(v);   // quad_perm:[1,0,3,2]
  if constexpr (J == 2) return dpp_perm<0x4E>(v);   // quad_perm:[2,3,0,1]
  if constexpr (J == 4) {                           // banks 0, 2 read lane + 4 (row_shl:4), banks 1, 3 lane - 4 (row_shr:4)
    const int t = __builtin_amdgcn_mov_dpp((int)v, 0x104, 0xF, 0x5, false);  // (banks 1, 3 are written next)
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
  }
  if constexpr (J == 8) return dpp_perm<0x128>(v);  // row_ror:8
  if constexpr (J == 16) {  // rows 1, 3 of the first operand <-> rows 0, 2 of the second
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return odd_half ? r[0] : r[1];
  }
  if constexpr (J == 32) {  // upper half of the first operand <-> lower half of the second
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return odd_half ? r[0] : r[1];
  }
  return v;
}
// lane bit set in m -> b, else a (m is wave-uniform: an SGPR pair)
__device__ __forceinline__ uint32_t select_by_mask(uint32_t a, uint32_t b, uint64_t m) {
  uint32_t r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
  return r;
}
// lanes l of register r that keep the smaller key at stage (k, j < 64) of the ascending bitonic network
constexpr uint64_t want_min_mask(int k, int j, int r) {
  uint64_t m = 0;
  for (int l = 0; l < 64; ++l) {
    const bool up = ((((r << 6) | l) & k) == 0);
    if (((l & j) == 0) == up) m |= 1ull << l;
  }
  return m;
}

template <int E, int K, int J>
__device__ __forceinline__ void packed_stage(uint64_t (&key)[E], int lane) {
  if constexpr (J >= 64) {
    constexpr int jr = J >> 6;
#pragma unroll
    for (int r = 0; r < E; ++r) {
      if ((r & jr) == 0) {
        const int r2 = r | jr;
        const bool up = ((r << 6) & K) == 0;
        const uint64_t a = key[r], b = key[r2];
        const bool swap = (a > b) == up;
        key[r] = swap ? b : a;
        key[r2] = swap ? a : b;
      }
    }
  } else {
    const bool odd_half = (lane & J) != 0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const uint32_t lo = (uint32_t)key[r], hi = (uint32_t)(key[r] >> 32);
      const uint32_t olo = lane_xor_u32<J>(lo, odd_half), ohi = lane_xor_u32<J>(hi, odd_half);
      const uint64_t o = ((uint64_t)ohi << 32) | olo;
      // take the partner's key where (partner < mine) == (this lane keeps the minimum)
      const uint64_t less = __builtin_amdgcn_ballot_w64(o < key[r]);
      const uint64_t take = ~(less ^ want_min_mask(K, J, r));
      key[r] = ((uint64_t)select_by_mask(hi, ohi, take) << 32) | select_by_mask(lo, olo, take);
    }
  }
}
template <int E, int K, int J>
__device__ __forceinline__ void packed_merge(uint64_t (&key)[E], int lane) {
  packed_stage<E, K, J>(key, lane);
  if constexpr (J > 1) packed_merge<E, K, (J >> 1)>(key, lane);
}
template <int E, int K>
__device__ __forceinline__ void packed_network(uint64_t (&key)[E], int lane) {
  if constexpr (K > 2) packed_network<E, (K >> 1)>(key, lane);
  packed_merge<E, K, (K >> 1)>(key, lane);
}

// key[r] / idx[r]: element r*64 + lane on entry (idx is overwritten), sorted position r*64 + lane on return;
// low: 64*E u32 of LDS owned by this wave (free on return).
template <int E>
__device__ __forceinline__ void wave_sort_fast(uint64_t (&key)[E], uint32_t (&idx)[E], uint32_t *low) {
  const int lane = threadIdx.x & 63;
  constexpr uint64_t kMask = E > 16 ? ((1ull << (kPackBits + 1)) - 1) : kPackMask;  // (the index of 64 E elements)
  static_assert(E <= 32, "the packed index holds 11 bits at most");
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const uint32_t p = (uint32_t)((r << 6) + lane);
    low[p] = (uint32_t)(key[r] & kMask);
    key[r] = (key[r] & ~kMask) | p;
  }
  packed_network<E, 64 * E>(key, lane);
#pragma unroll
  for (int r = 0; r < E; ++r) {
    idx[r] = (uint32_t)(key[r] & kMask);
    key[r] = (key[r] & ~kMask) | low[idx[r]];
  }
  // Keys that agree above the packed bits were ordered by index; that is the exact order unless their displaced bits
  // descend somewhere, i.e. unless the restored keys are not ascending.  (Equal keys -- the +inf residuals of a
  // degenerate model, the padding -- are in index order, which is the exact order.)
  bool ambiguous = false;
  uint64_t prev_last = 0ull;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    uint64_t before = __shfl_up(key[r], 1, 64);
    if (lane == 0) before = prev_last;
    if (before > key[r]) ambiguous = true;
    if (r + 1 < E) prev_last = __shfl(key[r], 63, 64);
  }
  if (__builtin_amdgcn_ballot_w64(ambiguous) != 0ull) wave_sort_regs<E>(key, idx);  // any order is a valid input
}

// bestNFA over the sorted registers of wave_sort_regs (position r*64 + lane holds e_{position+1}); same
// candidates, same per-lane visiting order and same reduction as best_nfa_wave
// the two table entries of sorted position r*64 + lane (k = position + 1), fetched ahead of the sort
template <int E>
__device__ __forceinline__ void nfa_tables_fetch(float (&cn)[E], float (&ck)[E], int n, const float *logc_n,
                                                 const float *logc_k) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int kk = (r << 6) + lane + 1;
    cn[r] = kk <= n ? logc_n[kk] : 0.0f;
    ck[r] = kk <= n ? logc_k[kk] : 0.0f;
  }
}
// wave_reduce_nfa: the same first-minimum reduction for both forms below
__device__ __forceinline__ NfaBest wave_reduce_nfa(double lb, int lk);

// K5's form (one wave per SIMD, latency-bound): the logarithms of all E positions first, branch-free, so that the E
// dependent chains interleave (values of positions that are not candidates are computed and dropped); table entries
// fetched ahead of the sort
template <int E>
__device__ __forceinline__ NfaBest best_nfa_regs_ilp(const uint64_t (&key)[E], int n, int s, double max_thr,
                                                     double logalpha0, double mult, double loge0, const float (&cn)[E],
                                                     const float (&ck)[E]) {
  const int lane = threadIdx.x & 63;
  double lb = pos_inf();
  int lk = 0x7FFFFFFF;
  double l10[E];
#pragma unroll
  for (int r = 0; r < E; ++r) l10[r] = det_log10_inline(u2d(key[r]) + (double)FLT_EPSILON);
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int kk = (r << 6) + lane + 1;
    if (kk > s && kk <= n) {
      const double ek = u2d(key[r]);
      if (ek <= max_thr) {
        const double logalpha = logalpha0 + mult * l10[r];
        const double nfa = loge0 + logalpha * (double)(kk - s) + (double)cn[r] + (double)ck[r];
        if (nfa < lb) {
          lb = nfa;
          lk = kk;
        }
      }
    }
  }
  return wave_reduce_nfa(lb, lk);
}

// K3's form (sixteen waves share a compute unit, issue-bound: fewer instructions and registers matter, latency does not)
template <int E>
__device__ __forceinline__ NfaBest best_nfa_regs(const uint64_t (&key)[E], int n, int s, double max_thr,
                                                 double logalpha0, double mult, double loge0, const float *logc_n,
                                                 const float *logc_k) {
  const int lane = threadIdx.x & 63;
  double lb = pos_inf();
  int lk = 0x7FFFFFFF;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int kk = (r << 6) + lane + 1;
    if (kk > s && kk <= n) {
      const double ek = u2d(key[r]);
      if (ek <= max_thr) {
        const double logalpha = logalpha0 + mult * det_log10(ek + (double)FLT_EPSILON);
        const double nfa = loge0 + logalpha * (double)(kk - s) + (double)logc_n[kk] + (double)logc_k[kk];
        if (nfa < lb) {
          lb = nfa;
          lk = kk;
        }
      }
    }
  }
  return wave_reduce_nfa(lb, lk);
}
__device__ __forceinline__ NfaBest wave_reduce_nfa(double lb, int lk) {
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(lb, off, 64);
    const int ok = __shfl_xor(lk, off, 64);
    if (ob < lb || (ob == lb && ok < lk)) {
      lb = ob;
      lk = ok;
    }
  }
  return NfaBest{lb, lk};
}

// the key at sorted position pos (every lane returns it)
template <int E>
__device__ __forceinline__ uint64_t sorted_key_at(const uint64_t (&key)[E], int pos) {
  uint64_t v = 0;
#pragma unroll
  for (int r = 0; r < E; ++r)
    if ((pos >> 6) == r) v = key[r];
  return __shfl(v, pos & 63, 64);
}

// logcombi tables of OpenMVG (float): logc_n[k] = log10 C(n,k), logc_k[m] = log10 C(m,s); L10[i] = log10(i).
// logcombi(k,n) = sum_{i=1..min(k,n-k)} (L10[n-i+1] - L10[i]) accumulated in double in that order, so the
// values for k = 0..n/2 are the running sums of one sequential pass; the rest is symmetry.  The pass itself cannot be
// split (the rounding of every partial sum is part of the result), but its terms can be fetched by the whole
// workgroup: `terms` (n / 2 + 1 doubles of LDS) takes L10[n-k+1] - L10[k], then thread 0 adds them up out of LDS with
// eight loads in flight -- the same sums as a loop over global memory, which cost ~50 us of dependent L2 latency per
// call on the critical path of K3 and K5.
__device__ void logc_n_block(int n, const double *__restrict__ L10, double *terms, float *logc_n, int n_threads) {
  const int kmax = n / 2;  // 2 k <= n
  for (int k = 1 + (int)threadIdx.x; k <= kmax; k += n_threads) terms[k] = L10[n - k + 1] - L10[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0;
    logc_n[0] = 0.0f;
    logc_n[n] = 0.0f;
    int k = 1;
    for (; k + 7 <= kmax; k += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = terms[k + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        r += t[u];
        logc_n[k + u] = (float)r;
        logc_n[n - k - u] = (float)r;
      }
    }
    for (; k <= kmax; ++k) {
      r += terms[k];
      logc_n[k] = (float)r;
      logc_n[n - k] = (float)r;
    }
  }
}

__device__ void logcombi_tables_block(int s, int n, const double *__restrict__ L10, double *terms, float *logc_n,
                                      float *logc_k, int n_threads = kThreads) {
  logc_n_block(n, L10, terms, logc_n, n_threads);
  for (int m = threadIdx.x; m <= n; m += n_threads) {
    float v = 0.0f;
    if (s < m) {
      int k = s;
      if (m - k < k) k = m - k;
      double r = 0.0;
      for (int i = 1; i <= k; ++i) r += L10[m - i + 1] - L10[i];
      v = (float)r;
    }
    logc_k[m] = v;
  }
  __syncthreads();
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 64;
  while (p < n) p <<= 1;
  return p;
}

// ---------------------------------------------------------------------------------------------------
// K3: F-matrix AC-RANSAC, one workgroup per selected view
// ---------------------------------------------------------------------------------------------------
constexpr int kFMaxM = 2048;  // putative matches per view the LDS sort holds
constexpr int kFPre = 64;     // hypotheses solved speculatively per batch while sampling is still uniform

// (a write-through store: see "A round's results go from the workgroup that computed them ..." at K5)
template <typename T>
__device__ __forceinline__ void store_through(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what a workgroup of k_fmatrix_fast's wide form delivers for its iteration of the first batch: the models and their
// (NFA, inlier count)
struct K3Spec {
  double models[27];
  double nfa[3];
  int k[3];
  int nm;
};
struct FFilterArgs {
  const uint32_t *view_sel;
  uint32_t n_sel;
  const uint32_t *view_off, *view_id, *view_wh;
  const uint32_t *put_count, *match_i, *match_key;
  const float2 *map_kpt, *q_kpt6;
  uint32_t qw, qh;
  double precision;
  int n_iter;
  uint64_t seed;
  int min_putative;
  const double *L10;
  uint32_t *geo_count, *geo_idx;
  double *geo_model;  // [view][10]: the winning F (normalised frame) and its errorMax, for guided matching; or null
  uint32_t *large_count, *large_list;  // views with more than kFMaxM matches, queued for k_fmatrix_large (or null)
  int *status;
  int skip_le;  // k_fmatrix_filter leaves views with at most this many putative matches to k_fmatrix_fast
  int fast_min;  // k_fmatrix_fast<W, M> takes the views with fast_min < matches <= M
  MergeMaskedArgs merge;  // enabled: k_fmatrix_fast first builds its view's putative list (K2 left to it)
  K3Spec *spec;           // wide form: [view slot][kF2Batch] results of the first batch; per view slot the arrivals
  unsigned int *spec_arrive;
};
// How the list reaches a kernel.  As kernel arguments it is 280 bytes, and a gang launch -- the members' lists side by
// side in the 4 KB argument segment -- held 14 of them: a session of 32 queries went out as 14 + 14 + 4, three launches one
// after the other for each of K3's forms.  Most of the list never changes for a context (the map's tables, the context's
// own buffers, the parameters): that part lives on the device (FFilterStatic, written when it changes -- in practice
// once), the kernels get a pointer to it and what does change per launch (84 bytes: 32 members a launch), and put the list
// together again first thing.
struct FFilterStatic {
  const uint32_t *view_off, *view_id, *view_wh;
  const uint32_t *put_count, *match_i, *match_key;
  const float2 *map_kpt;
  double precision;
  int n_iter;
  uint64_t seed;
  const double *L10;
  uint32_t *geo_count, *geo_idx;
  double *geo_model;
  uint32_t *large_count, *large_list;
  int *status;
  MergeMaskedArgs merge;  // (enabled, view_sel, view_widx0: per launch, below)
};
struct FFilterArgsPacked {
  const FFilterStatic *st;
  const uint32_t *view_sel;
  const float2 *q_kpt6;
  const uint32_t *merge_view_sel, *merge_view_widx0;
  K3Spec *spec;
  unsigned int *spec_arrive;
  uint32_t n_sel, qw, qh;
  int min_putative, skip_le, fast_min, merge_enabled;
};
__device__ __forceinline__ FFilterArgs ffilter_expand(const FFilterArgsPacked &P) {
  const FFilterStatic &T = *P.st;
  FFilterArgs A;
  A.view_sel = P.view_sel;
  A.n_sel = P.n_sel;
  A.view_off = T.view_off;
  A.view_id = T.view_id;
  A.view_wh = T.view_wh;
  A.put_count = T.put_count;
  A.match_i = T.match_i;
  A.match_key = T.match_key;
  A.map_kpt = T.map_kpt;
  A.q_kpt6 = P.q_kpt6;
  A.qw = P.qw;
  A.qh = P.qh;
  A.precision = T.precision;
  A.n_iter = T.n_iter;
  A.seed = T.seed;
  A.min_putative = P.min_putative;
  A.L10 = T.L10;
  A.geo_count = T.geo_count;
  A.geo_idx = T.geo_idx;
  A.geo_model = T.geo_model;
  A.large_count = T.large_count;
  A.large_list = T.large_list;
  A.status = T.status;
  A.skip_le = P.skip_le;
  A.fast_min = P.fast_min;
  A.merge = T.merge;
  A.merge.enabled = P.merge_enabled;
  A.merge.view_sel = P.merge_view_sel;
  A.merge.view_widx0 = P.merge_view_widx0;
  A.spec = P.spec;
  A.spec_arrive = P.spec_arrive;
  return A;
}

// small per-view state of the block-wide form (always in LDS)
struct FSmall {
  double pre_models[kFPre][27];
  int pre_nm[kFPre];
  double cur_models[27];
  int cur_nm;
  double best_model[9];
  double red_nfa[kThreads / 64];
  int red_k[kThreads / 64];
};

// where the per-match arrays of one view live: LDS for up to kFMaxM putative matches ...
struct FShared {
  uint64_t key[kFMaxM];
  uint32_t idx[kFMaxM];
  int32_t vec_index[kFMaxM];
  int32_t best_inl[kFMaxM];
  float logc_n[kFMaxM + 1];
  float logc_k[kFMaxM + 1];
  FSmall small;
};
struct FStoreLds {
  FShared *S;
  __device__ __forceinline__ uint64_t *key() const { return S->key; }
  __device__ __forceinline__ uint32_t *idx() const { return S->idx; }
  __device__ __forceinline__ int32_t *vec_index() const { return S->vec_index; }
  __device__ __forceinline__ int32_t *best_inl() const { return S->best_inl; }
  __device__ __forceinline__ float *logc_n() const { return S->logc_n; }
  __device__ __forceinline__ float *logc_k() const { return S->logc_k; }
};
// ... and one of a few global-memory slots for a view with more (k_fmatrix_large): the reference has no limit on the
// matches of a pair (MatchUtils.cpp:346-355), a near-duplicate of a map frame reaches several thousand
constexpr int kFLargeMaxM = 65536;  // matches per view the slots hold (the log10 table and 16-bit query indices end there)
constexpr int kFLargeSlots = 8;
struct FStoreGlobal {
  uint64_t *k;
  uint32_t *i;
  int32_t *vi, *bi;
  float *ln, *lk;
  __device__ __forceinline__ uint64_t *key() const { return k; }
  __device__ __forceinline__ uint32_t *idx() const { return i; }
  __device__ __forceinline__ int32_t *vec_index() const { return vi; }
  __device__ __forceinline__ int32_t *best_inl() const { return bi; }
  __device__ __forceinline__ float *logc_n() const { return ln; }
  __device__ __forceinline__ float *logc_k() const { return lk; }
};

// F-matrix AC-RANSAC of ONE view by the whole workgroup (block-wide sort); `st` says where the per-match arrays are,
// `max_m` how many matches they hold.  Returns false when the view has more matches than that.
template <typename Store>
__device__ bool fmatrix_filter_view(const FFilterArgs &A, uint32_t v, Store st, FSmall &S, int max_m) {
  const int tid = threadIdx.x;
  const int m = (int)A.put_count[v];
  const uint32_t off = A.view_off[v];
  constexpr int s = 7;
  if (m < A.min_putative || m <= s) {  // localization.cpp:408-415 ; ACRANSAC: nData <= sizeSample
    if (tid == 0) A.geo_count[v] = 0;
    return true;
  }
  if (m > max_m) return false;
  // NormalizePoints(x, w, h) for both images
  const int w1 = (int)A.view_wh[2 * v], h1 = (int)A.view_wh[2 * v + 1];
  const int w2 = (int)A.qw, h2 = (int)A.qh;
  const double s1 = 1.0 / sqrt((double)(w1 * h1)), s2 = 1.0 / sqrt((double)(w2 * h2));
  const double t1x = -0.5 * (double)w1 * s1, t1y = -0.5 * (double)h1 * s1;
  const double t2x = -0.5 * (double)w2 * s2, t2y = -0.5 * (double)h2 * s2;
  const double Dg = sqrt((double)w2 * (double)w2 + (double)h2 * (double)h2);
  const double Ar = (double)w2 * (double)h2;
  const double logalpha0 = det_log10(2.0 * Dg / Ar / s2);
  const double max_thr = (A.precision * A.precision) * s2 * s2;
  const double loge0 = det_log10(3.0 * (double)(m - s));
  const uint32_t stream = A.view_id[v];
  const int P = next_pow2(m);

  auto point = [&](int p, double &x, double &y, double &u, double &w) {
    const uint32_t i = A.match_i[off + p];
    const uint32_t j = A.match_key[off + p] & 0xFFFFu;
    const float2 a = A.map_kpt[off + i];
    const float2 b = A.q_kpt6[j];
    x = s1 * (double)a.x + t1x;
    y = s1 * (double)a.y + t1y;
    u = s2 * (double)b.x + t2x;
    w = s2 * (double)b.y + t2y;
  };
  auto solve = [&](const int32_t *smp, double *models) -> int {
    double x1[14], x2[14];
    for (int i = 0; i < 7; ++i) point(smp[i], x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]);
    return seven_point(x1, x2, models);
  };

  logcombi_tables_block(s, m, A.L10, reinterpret_cast<double *>(st.key()), st.logc_n(), st.logc_k());  // the key array is free until the first sort

  double min_nfa = pos_inf();
  int n_in = 0;
  long n_iter = A.n_iter;
  long n_reserve = n_iter / 10;
  n_iter -= n_reserve;
  bool identity = true;
  int n_index = m;
  long pre_base = -1;

  for (long iter = 0; iter < n_iter; ++iter) {
    const double *models;
    int nm;
    if (identity) {
      if (pre_base < 0 || iter >= pre_base + kFPre) {
        __syncthreads();
        pre_base = iter;
        if (tid < kFPre) {
          int32_t smp[7];
          double mm[27];
          ac_sample<7>(nullptr, m, A.seed, STAGE_FMATRIX, stream, (uint32_t)(pre_base + tid), smp);
          const int k = solve(smp, mm);
          S.pre_nm[tid] = k;
          for (int q = 0; q < 9 * k; ++q) S.pre_models[tid][q] = mm[q];
        }
        __syncthreads();
      }
      models = S.pre_models[iter - pre_base];
      nm = S.pre_nm[iter - pre_base];
    } else {
      __syncthreads();
      if (tid == 0) {
        int32_t smp[7];
        double mm[27];
        ac_sample<7>(st.vec_index(), n_index, A.seed, STAGE_FMATRIX, stream, (uint32_t)iter, smp);
        const int k = solve(smp, mm);
        S.cur_nm = k;
        for (int q = 0; q < 9 * k; ++q) S.cur_models[q] = mm[q];
      }
      __syncthreads();
      models = S.cur_models;
      nm = S.cur_nm;
    }
    bool better = false;
    for (int k = 0; k < nm; ++k) {
      double M[9];
      for (int q = 0; q < 9; ++q) M[q] = models[9 * k + q];
      __syncthreads();  // key/idx of the previous model are no longer read
      for (int p = tid; p < P; p += kThreads) {
        uint64_t kv = ~0ull;
        if (p < m) {
          double x, y, u, w;
          point(p, x, y, u, w);
          kv = d2u(err_fmatrix(M, x, y, u, w));
        }
        st.key()[p] = kv;
        st.idx()[p] = (uint32_t)p;
      }
      __syncthreads();
      bitonic_sort(st.key(), st.idx(), P);
      const NfaBest b = best_nfa_block(st.key(), m, s, max_thr, logalpha0, 0.5, loge0, st.logc_n(), st.logc_k(), S.red_nfa,
                                       S.red_k);
      if (b.nfa < min_nfa) {
        better = true;
        min_nfa = b.nfa;
        n_in = b.k;
        for (int p = tid; p < n_in; p += kThreads) st.best_inl()[p] = (int32_t)st.idx()[p];
        if (tid < 9) S.best_model[tid] = M[tid];
      }
    }
    if ((better && min_nfa < 0.0) || (iter + 1 == n_iter && n_reserve)) {
      if (n_in == 0) {
        n_iter++;
        n_reserve--;
      } else {
        __syncthreads();
        for (int p = tid; p < n_in; p += kThreads) st.vec_index()[p] = st.best_inl()[p];
        n_index = n_in;
        identity = false;
        if (n_reserve) {
          n_iter = iter + 1 + n_reserve;
          n_reserve = 0;
        }
      }
    }
  }
  __syncthreads();
  if (min_nfa >= 0.0) n_in = 0;
  if ((double)n_in > 7 * 2.5) {
    for (int p = tid; p < n_in; p += kThreads) A.geo_idx[off + p] = (uint32_t)st.best_inl()[p];
    if (tid == 0) A.geo_count[v] = (uint32_t)n_in;
    if (A.geo_model && tid == 0) {  // errorMax = the residual of the last inlier (the list ascends by residual)
      double x, y, u, w;
      point(st.best_inl()[n_in - 1], x, y, u, w);
      double *gm = A.geo_model + 10 * (size_t)v;
      for (int q = 0; q < 9; ++q) gm[q] = S.best_model[q];
      gm[9] = err_fmatrix(S.best_model, x, y, u, w);
    }
  } else if (tid == 0) {
    A.geo_count[v] = 0;
  }
  return true;
}

struct FmatrixFilterBody {
  static constexpr int kGangThreads = kThreads;
  static __device__ __forceinline__ void run(FFilterArgsPacked P) {
    const FFilterArgs A = ffilter_expand(P);
    extern __shared__ unsigned char smem_raw[];
    FShared &S = *reinterpret_cast<FShared *>(smem_raw);
    const uint32_t v = A.view_sel ? A.view_sel[blockIdx.x] : blockIdx.x;
    const int m = (int)A.put_count[v];
    if (m <= A.skip_le) return;
    if (!fmatrix_filter_view(A, v, FStoreLds{&S}, S.small, kFMaxM)) {
      // more matches than the LDS form holds: queue the view for k_fmatrix_large
      if (threadIdx.x == 0) {
        A.geo_count[v] = 0;
        if (A.large_list) {
          const uint32_t slot = atomicAdd(A.large_count, 1u);
          A.large_list[slot] = v;
        } else {
          atomicOr(A.status, 1);
        }
      }
    }
  }
};
__global__ __launch_bounds__(kThreads) void k_fmatrix_filter(FFilterArgsPacked P) {
  FmatrixFilterBody::run(P);
}

// The views k_fmatrix_filter queued (more than kFMaxM putative matches): kFLargeSlots persistent workgroups, each
// owning one global-memory slot, take them in turn.  Same algorithm, same arithmetic, same results; the sort runs
// through L2 instead of LDS, which costs milliseconds for such a view instead of failing the query.
struct FLargeArgs {
  uint64_t *key;
  uint32_t *idx;
  int32_t *vec_index, *best_inl;
  float *logc_n, *logc_k;
  int slot_m;  // matches one slot holds: a power of two >= the map's longest view, at most kFLargeMaxM
};
struct FmatrixLargeBody {
  static constexpr int kGangThreads = kThreads;
  static __device__ __forceinline__ void run(FFilterArgsPacked P, FLargeArgs W) {
    const FFilterArgs A = ffilter_expand(P);
    __shared__ FSmall S;
    const uint32_t n = *A.large_count;
    const size_t o = (size_t)blockIdx.x * W.slot_m;
    FStoreGlobal st{W.key + o, W.idx + o, W.vec_index + o, W.best_inl + o, W.logc_n + (size_t)blockIdx.x * (W.slot_m + 1),
                    W.logc_k + (size_t)blockIdx.x * (W.slot_m + 1)};
    for (uint32_t t = blockIdx.x; t < n; t += gridDim.x) {
      const uint32_t v = A.large_list[t];
      if (!fmatrix_filter_view(A, v, st, S, W.slot_m) && threadIdx.x == 0) {
        A.geo_count[v] = 0;
        atomicOr(A.status, 1);  // more than 65 536 matches in one view
      }
      __syncthreads();
    }
  }
};
__global__ __launch_bounds__(kThreads) void k_fmatrix_large(FFilterArgsPacked P, FLargeArgs W) {
  FmatrixLargeBody::run(P, W);
}

// ---------------------------------------------------------------------------------------------------
// K3, fast form (views with at most kF2MaxM putative matches -- in practice all of them): the same AC-RANSAC,
// restructured so that nothing waits on a single lane.
//   * the 7-point solver runs wave-wide (wave_seven_point), 8 hypotheses at a time, one per wave;
//   * while sampling is still uniform the iterations are independent, so a batch of them is solved and their
//     models are evaluated speculatively, one MODEL per wave (residuals -> per-wave bitonic sort -> bestNFA), eight
//     side by side; the sequential acceptance rule of ACRANSAC is then replayed over the results in iteration
//     order, and the winner's inlier list is rebuilt by evaluating that one model again;
//   * after the switch to inlier sampling each iteration depends on the previous one: wave 0 solves, waves 0..2
//     evaluate the (up to 3) models side by side.
// Results are bit-identical to k_fmatrix_filter (same samples, same arithmetic per value, same tie rules).
// ---------------------------------------------------------------------------------------------------
constexpr int kF2MaxM = 512;   // putative matches per view (one wave sorts one model's residuals)
constexpr int kF2Batch = 32;   // uniform iterations solved speculatively per batch
constexpr int kK3WideViews = 256;  // view lists the wide form takes (its result slots: 2 MB per context)

// (W = waves per view: 16 for a query alone on the GPU -- a view's latency --, 4 when the GPU is shared: a workgroup of 16
// waves at 124 VGPRs is a compute unit's whole register file, i.e. it starts only on a compute unit nothing else runs on
// and nothing else runs beside it; at 8 the Hamming scans of the other queries keep three of their waves per SIMD, at
// 4 all of them.  Measured at 20 queries in flight: 16 / 8 / 4 waves 3 259 / 3 413 / 3 460 queries/s -- K3 costs the
// others 13 us per query instead of 36.)
// (M = putative matches per view the form holds: 512 -- at most 8 residuals per lane in the register sort, 124 VGPRs --
// or, round 3, 1 024 for the views of a query that nearly duplicates a map frame: 16 per lane, more registers, launched
// only while such queries come, Map::k3_big_credit; they took the block-wide LDS form before, 0.4 ms per frame)
template <int W, int MaxM = 512>
struct F2SharedT {
  static constexpr int kF2MaxM = MaxM;
  uint32_t idx[W][kF2MaxM];  // sorted match indices of the model each wave evaluated last
  double pts[4][kF2MaxM];  // normalised x, y of the map keypoint, u, w of the query keypoint (one plane each: a
                           // wave reading element p of 64 consecutive matches hits 64 different banks)
  int32_t vec_index[kF2MaxM];
  int32_t best_inl[kF2MaxM];
  float logc_n[kF2MaxM + 1];
  float logc_k[kF2MaxM + 1];
  double pre_models[kF2Batch][27];
  int pre_nm[kF2Batch];
  double res_nfa[kF2Batch][3];
  int res_k[kF2Batch][3];
  double best_model[9];
  int iter_end[kF2Batch];  // flattened index one past the last model of each iteration of the batch
  uint8_t flat_b[kF2Batch * 3], flat_k[kF2Batch * 3];
  unsigned int wide_ticket;  // the wide form: this workgroup's arrival number
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ascending bitonic sort of one wave's own (key, idx) segment; no workgroup barrier involved
__device__ void bitonic_sort_wave(uint64_t *kw, uint32_t *iw, int P) {
  const int lane = threadIdx.x & 63;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (P >> 1); t += 64) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i + j;
        const bool up = (i & k) == 0;
        const uint64_t ka = kw[i], kb = kw[l];
        const uint32_t ia = iw[i], ib = iw[l];
        const bool a_gt_b = pair_less(kb, ib, ka, ia);
        if (a_gt_b == up) {
          kw[i] = kb;
          kw[l] = ka;
          iw[i] = ib;
          iw[l] = ia;
        }
      }
      wave_lds_sync();
    }
  }
}

template <int W, int MaxM = 512, bool Wide = false>
struct FmatrixFastBody {
  static constexpr int kGangThreads = W * 64;
  static __device__ __forceinline__ void run(FFilterArgsPacked packed) {
    const FFilterArgs A = ffilter_expand(packed);
    constexpr int kF2Waves = W, kF2Threads = W * 64, kF2MaxM = MaxM;
    constexpr bool kF2Wide = Wide;
    using F2Shared = F2SharedT<W, MaxM>;
#include "fmatrix_fast.body.inc"
  }
};
template <int W, int MaxM = 512, bool Wide = false>
__global__ __launch_bounds__(W * 64) void k_fmatrix_fast(FFilterArgsPacked packed) {
  const FFilterArgs A = ffilter_expand(packed);
  constexpr int kF2Waves = W, kF2Threads = W * 64, kF2MaxM = MaxM;
  constexpr bool kF2Wide = Wide;
  using F2Shared = F2SharedT<W, MaxM>;
#include "fmatrix_fast.body.inc"
}

// ---------------------------------------------------------------------------------------------------
// K4: 2D-3D candidates and their de-duplication
// ---------------------------------------------------------------------------------------------------
// Every geometric match whose map feature has a landmark is a candidate for its query feature, ranked by
//   order key = dist << 48 | view_id << 24 | position in the view's geometric list
// (smaller is better; equal distance -> earlier in std::map iteration order = lower view id, then list order).
// dist = featDist[(v,q)][j] = d0 of the LAST putative match of the view that hit query feature j.
// matchProviderToMatchSet keeps ONE candidate per query feature (the minimum), so that is all a context -- or a shard --
// ever materialises: pass 1 (k_emit_min) computes every candidate's key and keeps the per-feature minimum with
// atomicMin, pass 2 (k_emit_win) turns the candidates that ARE the minimum into the part.  A part therefore holds at
// most one candidate per query feature (<= 65 535): no capacity can overflow however many geometric matches the views
// have, and a shard's exchange shrinks to its winners.
constexpr uint16_t kNoDist = 0xFFFFu;

struct EmitMinBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const uint32_t *view_sel, uint32_t n_sel, const uint32_t *view_off,
                                          const uint32_t *view_id, const uint32_t *put_count,
                                          const uint32_t *match_i, const uint32_t *match_key,
                                          const uint32_t *geo_count, const uint32_t *geo_idx,
                                          const uint32_t *geo_j /*null: geo_idx indexes the putative list; else
                                                                  (geo_idx, geo_j) = (map feature, query feature)
                                                                  of a guided match*/,
                                          const int32_t *row_landmark, unsigned long long *best64,
                                          uint16_t *geo_dist, uint32_t min_putative, uint32_t *view_stats) {
    // one workgroup (four waves) per selected view: the stage is a chain of dependent loads per candidate, so the waves
    // take 64 candidates each side by side instead of one wave walking them 64 at a time
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x;
    if (gw >= n_sel) return;
    const uint32_t v = view_sel ? view_sel[gw] : gw;
    const uint32_t ng = geo_count[v];
    if (threadIdx.x == 0) {  // the counts the reference prints (localization.cpp:416,458)
      if (put_count[v] >= min_putative) atomicAdd(&view_stats[0], 1u);
      if (ng > 0) atomicAdd(&view_stats[1], 1u);
      if (put_count[v] > 512u) atomicMax(&view_stats[2], put_count[v]);  // (the host's hint for K3's launch forms)
    }
    if (ng == 0) return;
    const uint32_t off = view_off[v];
    const uint32_t np = put_count[v];
    // the view's putative keys in LDS (one segment per wave): the "last match with the same query feature" search
    // below is a dependent backward scan, far too slow against L2
    __shared__ uint32_t keys[kFMaxM];
    const bool staged = np <= (uint32_t)kFMaxM;
    if (staged) {
      for (uint32_t k = threadIdx.x; k < np; k += 256) keys[k] = match_key[off + k];
      __syncthreads();
    }
    // Round 3: from 128 putative matches on, "the last putative match with this query feature" is looked up in a hash
    // table (query feature -> largest list position, open addressing in LDS, at most half full) instead of walked to: the
    // walk is np / 2 dependent LDS reads per geometric match, and a frame that nearly duplicates a map view has 1 500 of
    // each -- 270 us per frame in the image-in leg, a few us now.  Same answer: the entry with the largest position.
    constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    __shared__ uint32_t tab[2 * kFMaxM];  // (query feature << 11) | position; positions < kFMaxM = 2 048
    static_assert(kFMaxM <= 2048, "a list position has 11 bits in the table's entries");
    const bool hashed = staged && np >= 128u;
    uint32_t tmask = 0;
    if (hashed) {
      uint32_t tsize = 256;
      while (tsize < 2u * np) tsize <<= 1;
      tmask = tsize - 1u;
      for (uint32_t e = threadIdx.x; e < tsize; e += 256) tab[e] = kEmpty;
      __syncthreads();
      for (uint32_t k = threadIdx.x; k < np; k += 256) {
        const uint32_t jq = keys[k] & 0xFFFFu, val = (jq << 11) | k;
        uint32_t hsh = (jq * 2654435761u) >> 7 & tmask;
        for (;;) {
          uint32_t cur = __hip_atomic_load(&tab[hsh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (cur == kEmpty) {
            cur = atomicCAS(&tab[hsh], kEmpty, val);
            if (cur == kEmpty) break;
          }
          if ((cur >> 11) == jq) {
            atomicMax(&tab[hsh], val);
            break;
          }
          hsh = (hsh + 1u) & tmask;
        }
      }
      __syncthreads();
    }
    const uint64_t vkey = (uint64_t)(view_id[v] & 0xFFFFFFu) << 24;
    for (uint32_t p0 = (threadIdx.x >> 6) * 64; p0 < ng; p0 += 256) {
      const uint32_t p = p0 + lane;
      if (p >= ng) continue;
      uint32_t i, j;
      if (geo_j) {
        i = geo_idx[off + p];
        j = geo_j[off + p];
      } else {
        const uint32_t pp = geo_idx[off + p];
        i = match_i[off + pp];
        j = match_key[off + pp] & 0xFFFFu;
      }
      uint16_t dist16 = kNoDist;
      if (row_landmark[off + i] >= 0 && hashed) {
        uint32_t hsh = (j * 2654435761u) >> 7 & tmask;
        for (;;) {
          const uint32_t cur = tab[hsh];
          if (cur == kEmpty) break;  // (no putative match of this view has this query feature: a guided match)
          if ((cur >> 11) == j) {
            dist16 = (uint16_t)(keys[cur & 2047u] >> 16);
            break;
          }
          hsh = (hsh + 1u) & tmask;
        }
      } else if (row_landmark[off + i] >= 0) {
        for (int32_t k = (int32_t)np - 1; k >= 0; --k) {  // last putative match with the same query feature
          const uint32_t kk = staged ? keys[k] : match_key[off + k];
          if ((kk & 0xFFFFu) == j) {
            dist16 = (uint16_t)(kk >> 16);
            break;
          }
        }
        // featDist has no entry for a query feature no putative match of this view hit (only possible for guided
        // matches): matchProviderToMatchSet then skips the match (SfMDataUtils.cpp:105-106) -> dist16 stays kNoDist
      }
      geo_dist[off + p] = dist16;
      if (dist16 != kNoDist)
        atomicMin(&best64[j], ((unsigned long long)dist16 << 48) | vkey | (unsigned long long)(p & 0xFFFFFFu));
    }
  }
};
__global__ __launch_bounds__(256) void k_emit_min(const uint32_t *view_sel, uint32_t n_sel, const uint32_t *view_off,
                                          const uint32_t *view_id, const uint32_t *put_count,
                                          const uint32_t *match_i, const uint32_t *match_key,
                                          const uint32_t *geo_count, const uint32_t *geo_idx,
                                          const uint32_t *geo_j /*null: geo_idx indexes the putative list; else
                                                                  (geo_idx, geo_j) = (map feature, query feature)
                                                                  of a guided match*/,
                                          const int32_t *row_landmark, unsigned long long *best64,
                                          uint16_t *geo_dist, uint32_t min_putative, uint32_t *view_stats) {
  EmitMinBody::run(view_sel, n_sel, view_off, view_id, put_count, match_i, match_key, geo_count, geo_idx, geo_j, row_landmark,
                   best64, geo_dist, min_putative, view_stats);
}

struct EmitWinBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const uint32_t *view_sel, uint32_t n_sel, const uint32_t *view_off,
                                          const uint32_t *view_id, const uint32_t *match_i,
                                          const uint32_t *match_key, const uint32_t *geo_count,
                                          const uint32_t *geo_idx, const uint32_t *geo_j,
                                          const int32_t *row_landmark, const uint32_t *landmark_id,
                                          const double *landmark_X, const unsigned long long *best64,
                                          const uint16_t *geo_dist, Candidate *cand, uint32_t cap,
                                          uint32_t *n_cand, int *status) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x;
    if (gw >= n_sel) return;
    const uint32_t v = view_sel ? view_sel[gw] : gw;
    const uint32_t ng = geo_count[v];
    if (ng == 0) return;
    const uint32_t off = view_off[v];
    const uint64_t vkey = (uint64_t)(view_id[v] & 0xFFFFFFu) << 24;
    for (uint32_t p0 = (threadIdx.x >> 6) * 64; p0 < ng; p0 += 256) {
      const uint32_t p = p0 + lane;
      bool has = false;
      uint32_t i = 0, j = 0;
      unsigned long long order = 0;
      if (p < ng) {
        const uint16_t d = geo_dist[off + p];
        if (d != kNoDist) {
          if (geo_j) {
            i = geo_idx[off + p];
            j = geo_j[off + p];
          } else {
            const uint32_t pp = geo_idx[off + p];
            i = match_i[off + pp];
            j = match_key[off + pp] & 0xFFFFu;
          }
          order = ((unsigned long long)d << 48) | vkey | (unsigned long long)(p & 0xFFFFFFu);
          has = best64[j] == order;
        }
      }
      // one atomic per wave step instead of one per winner
      const unsigned long long mask = __ballot(has);
      uint32_t slot0 = 0;
      if (lane == 0 && mask) slot0 = atomicAdd(n_cand, (uint32_t)__popcll(mask));
      slot0 = __shfl(slot0, 0, 64);
      if (!has) continue;
      const uint32_t slot = slot0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      if (slot >= cap) {  // cannot happen: at most one winner per query feature and cap >= SFMLOC_MAX_QUERY_ROWS
        atomicOr(status, 2);
        continue;
      }
      const int32_t lm = row_landmark[off + i];
      Candidate c;
      c.order = order;
      c.qfeat = j;
      c.landmark_id = landmark_id[lm];
      c.X[0] = landmark_X[3 * lm];
      c.X[1] = landmark_X[3 * lm + 1];
      c.X[2] = landmark_X[3 * lm + 2];
      cand[slot] = c;
    }
  }
};
__global__ __launch_bounds__(256) void k_emit_win(const uint32_t *view_sel, uint32_t n_sel, const uint32_t *view_off,
                                          const uint32_t *view_id, const uint32_t *match_i,
                                          const uint32_t *match_key, const uint32_t *geo_count,
                                          const uint32_t *geo_idx, const uint32_t *geo_j,
                                          const int32_t *row_landmark, const uint32_t *landmark_id,
                                          const double *landmark_X, const unsigned long long *best64,
                                          const uint16_t *geo_dist, Candidate *cand, uint32_t cap,
                                          uint32_t *n_cand, int *status) {
  EmitWinBody::run(view_sel, n_sel, view_off, view_id, match_i, match_key, geo_count, geo_idx, geo_j, row_landmark, landmark_id, landmark_X, best64, geo_dist, cand, cap, n_cand, status);
}

// A "part" is what one shard contributes for one query: 16-byte header {u32 n_cand, pad} + cap candidates.
// The selection kernels run over n_parts parts laid out back to back (n_parts = 1 on a single GPU; after the
// all-gather it is the number of shards).  Candidate c of part p has the global index p*cap + c.
//
// PACKED parts (the multi-GPU exchange, sfmloc_shard_export_packed): one buffer per shard for a whole batch of B
// queries -- header {u32 total, n_queries, budget, flags}, u32 count[B], u32 offset[B], then (16-byte aligned) the
// candidates of all B queries back to back in arrival order; query i's are [offset[i], offset[i] + count[i]).  The
// kernels address a candidate of part p as p*cap + c with cap = the budget and c counted from the start of the
// buffer's candidate area, so the two layouts differ only in where a part's range lies.
struct PartLayout {
  uint32_t packed_b;  // 0 = plain part; else B of the packed layout
  uint32_t qi;        // query index inside the packed batch
};
__host__ __device__ __forceinline__ uint64_t packed_cands_offset(uint32_t n_queries) {
  return (16ull + 8ull * n_queries + 15ull) & ~15ull;
}
__device__ __forceinline__ const Candidate *part_cands(const unsigned char *parts, uint64_t part_bytes, uint32_t p,
                                                       PartLayout L) {
  return reinterpret_cast<const Candidate *>(parts + (uint64_t)p * part_bytes +
                                             (L.packed_b ? packed_cands_offset(L.packed_b) : (uint64_t)kPartHeaderBytes));
}
// candidates [c0, c1) of part p belong to this query
__device__ __forceinline__ void part_range(const unsigned char *parts, uint64_t part_bytes, uint32_t p, uint32_t cap,
                                           PartLayout L, uint32_t *c0, uint32_t *c1) {
  const uint32_t *h = reinterpret_cast<const uint32_t *>(parts + (uint64_t)p * part_bytes);
  if (L.packed_b) {
    const uint32_t n = h[4 + L.qi], off = h[4 + L.packed_b + L.qi];
    *c0 = min(off, cap);
    *c1 = min(off + n, cap);
  } else {
    *c0 = 0;
    *c1 = min(h[0], cap);
  }
}

struct CandidatesMinBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const unsigned char *parts, uint32_t n_parts,
                                                uint64_t part_bytes, uint32_t cap, uint32_t nq,
                                                unsigned long long *best, int *status, PartLayout L) {
    for (uint32_t p = blockIdx.y; p < n_parts; p += gridDim.y) {
      const Candidate *cand = part_cands(parts, part_bytes, p, L);
      uint32_t c0, c1;
      part_range(parts, part_bytes, p, cap, L, &c0, &c1);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t *h = reinterpret_cast<const uint32_t *>(parts + (uint64_t)p * part_bytes);
        // a shard produced more candidates than its part holds (packed: than the batch's budget, or than a context holds)
        if (L.packed_b ? (h[0] > cap || h[3] != 0) : (h[0] > cap)) atomicOr(status, 2);
      }
      for (uint32_t c = c0 + blockIdx.x * blockDim.x + threadIdx.x; c < c1; c += gridDim.x * blockDim.x)
        if (cand[c].qfeat < nq) atomicMin(&best[cand[c].qfeat], (unsigned long long)cand[c].order);
    }
  }
};
__global__ __launch_bounds__(256) void k_candidates_min(const unsigned char *parts, uint32_t n_parts,
                                                uint64_t part_bytes, uint32_t cap, uint32_t nq,
                                                unsigned long long *best, int *status, PartLayout L) {
  CandidatesMinBody::run(parts, n_parts, part_bytes, cap, nq, best, status, L);
}

__device__ void p3p_init_block(const P3pArgs &A, int n, int n_threads, double *s_terms);  // K5's start, below

struct MatchSetFinishBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(const unsigned char *parts, uint32_t n_parts,
                                                  uint64_t part_bytes,
                                                  uint32_t cap, const unsigned long long *best,
                                                  uint32_t *winner, uint32_t nq, const float2 *q_kpt,
                                                  uint32_t *ms_n, uint32_t *ms_qfeat, uint32_t *ms_landmark,
                                                  double *pt2d, double *pt3d, int radial_k3, double f, double ppx,
                                                  double ppy, double k1, double k2, double k3, PartLayout L,
                                                  P3pArgs init /*K5's start, by this workgroup: one launch less*/) {
    // one workgroup; winners are compacted in query-feature order, 1024 features per pass (a pass is a chain of
    // dependent loads, so fewer, wider passes)
    __shared__ uint32_t wave_cnt[16];
    __shared__ uint32_t base_s;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    // which candidate holds each query feature's minimum (the parts hold winners only -- a few
    // hundred candidates each --, so the one workgroup that compacts them can also find them: one launch less)
    for (uint32_t p = 0; p < n_parts; ++p) {
      const Candidate *cand = part_cands(parts, part_bytes, p, L);
      uint32_t c0, c1;
      part_range(parts, part_bytes, p, cap, L, &c0, &c1);
      for (uint32_t c = c0 + threadIdx.x; c < c1; c += 1024)
        if (cand[c].qfeat < nq && best[cand[c].qfeat] == (unsigned long long)cand[c].order)
          winner[cand[c].qfeat] = p * cap + c;
    }
    __syncthreads();
    for (uint32_t j0 = 0; j0 < nq; j0 += 1024) {
      const uint32_t j = j0 + threadIdx.x;
      const bool has = j < nq && best[j] != ~0ull;
      const unsigned long long mask = __ballot(has);
      if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(mask);
      __syncthreads();
      uint32_t pre = base_s;
      for (uint32_t w = 0; w < wave; ++w) pre += wave_cnt[w];
      if (has) {
        const uint32_t pos = pre + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        const uint32_t w = winner[j];
        const Candidate c = part_cands(parts, part_bytes, w / cap, L)[w % cap];
        ms_qfeat[pos] = j;
        ms_landmark[pos] = c.landmark_id;
        const float2 kp = q_kpt[j];
        double ux = (double)kp.x, uy = (double)kp.y;  // cam_I->get_ud_pixel(qFeatLoc[j])   localization.cpp:484-487
        if (radial_k3) ud_pixel_k3(f, ppx, ppy, k1, k2, k3, ux, uy, &ux, &uy);
        pt2d[2 * pos] = ux;
        pt2d[2 * pos + 1] = uy;
        pt3d[3 * pos] = c.X[0];
        pt3d[3 * pos + 1] = c.X[1];
        pt3d[3 * pos + 2] = c.X[2];
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < 16; ++w) t += wave_cnt[w];
        base_s += t;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) *ms_n = base_s;
    // K5's initial state, normalised points and logcombi tables for the base_s correspondences just written
    __shared__ double s_terms[kP3pMaxN / 2 + 1];
    __syncthreads();
    p3p_init_block(init, (int)base_s, 1024, s_terms);
  }
};
__global__ __launch_bounds__(1024) void k_match_set_finish(const unsigned char *parts, uint32_t n_parts,
                                                  uint64_t part_bytes,
                                                  uint32_t cap, const unsigned long long *best,
                                                  uint32_t *winner, uint32_t nq, const float2 *q_kpt,
                                                  uint32_t *ms_n, uint32_t *ms_qfeat, uint32_t *ms_landmark,
                                                  double *pt2d, double *pt3d, int radial_k3, double f, double ppx,
                                                  double ppy, double k1, double k2, double k3, PartLayout L,
                                                  P3pArgs init /*K5's start, by this workgroup: one launch less*/) {
  MatchSetFinishBody::run(parts, n_parts, part_bytes, cap, best, winner, nq, q_kpt, ms_n, ms_qfeat, ms_landmark, pt2d, pt3d,
                          radial_k3, f, ppx, ppy, k1, k2, k3, L, init);
}

// ---------------------------------------------------------------------------------------------------
// K5: P3P AC-RANSAC as rounds of (evaluate a batch of hypotheses | replay the sequential rule)
// ---------------------------------------------------------------------------------------------------
// the start of K5: the state machine's initial state, the K^-1-normalised image points and the logcombi tables.  A
// workgroup of n_threads threads; n = the number of 2D-3D correspondences (what k_match_set_finish counted).
__device__ void p3p_init_block(const P3pArgs &A, int n, int n_threads, double *s_terms) {
  P3pState &st = *A.state;
  __shared__ int go;
  if (threadIdx.x == 0) {
    st.n = n;
    st.iter = 0;
    const long maxit = A.max_iteration;
    st.n_reserve = (int)(maxit / 10);
    st.n_iter = (int)(maxit - maxit / 10);
    st.n_index = n;
    st.identity = 1;
    st.n_in = 0;
    st.done = 0;
    st.rounds = 0;
    st.status = 0;
    st.arrive = 0u;
    st.finished = 0;
    st.batch_limit = 1 << 30;
    st.switch_iter = 0;
    st.prep_iter = -1;
    st.prep_n = 0;
    st.first_hit = ~0u;
    st.min_nfa = pos_inf();
    st.errmax = pos_inf();
    for (int i = 0; i < 12; ++i) st.model[i] = 0.0;
    go = 1;
    // localization.cpp:506 "cpt > MINUM_NUMBER_OF_POINT_RESECTION"; ACRANSAC: nData <= sizeSample -> nothing
    if (n <= A.min_resection_points || n <= 3) {
      st.done = 1;
      st.finished = 1;  // nothing to estimate: the result record written below is final
      go = 0;
    }
    if (n > A.max_n) {
      st.done = 1;
      st.finished = 1;
      st.status = 4;
      go = 0;
    }
    A.result->ok = 0;
    A.result->n_inliers = 0;
    A.result->n_matches_2d3d = n;
    A.result->iterations = 0;
    A.result->status = st.status;
    // a query that ends here (too few correspondences) must not report the PREVIOUS query's numbers: the record is the
    // context's, and a result is a function of the query alone (bench.py / tests compare results bit for bit)
    A.result->reserved = 0;
    A.result->nfa = 0.0;
    A.result->error_max = 0.0;
    for (int i = 0; i < 12; ++i) A.result->P[i] = 0.0;
    for (int i = 0; i < 9; ++i) {
      A.result->K[i] = 0.0;
      A.result->R[i] = 0.0;
    }
    for (int i = 0; i < 3; ++i) {
      A.result->t[i] = 0.0;
      A.result->center[i] = 0.0;
    }
  }
  __syncthreads();
  if (!go) return;
  // normalise by K^-1: x * (1/f) + (-pp/f)
  const double inv_f = 1.0 / A.focal;
  const double cx = -A.ppx * inv_f, cy = -A.ppy * inv_f;
  for (int i = threadIdx.x; i < n; i += n_threads) {
    A.xn[2 * i] = A.pt2d[2 * i] * inv_f + cx;
    A.xn[2 * i + 1] = A.pt2d[2 * i + 1] * inv_f + cy;
  }
  // (the table pass wants n / 2 + 1 doubles of scratch: LDS up to kP3pMaxN correspondences, global beyond)
  logcombi_tables_block(3, n, A.L10, n > kP3pMaxN ? A.ws_terms : s_terms, A.logc_n, A.logc_k, n_threads);
}
struct P3pInitBody {
  static constexpr int kGangThreads = kThreads;
  static __device__ __forceinline__ void run(P3pArgs A) {
    __shared__ double s_terms[kP3pMaxN / 2 + 1];
    p3p_init_block(A, (int)*A.ms_n, kThreads, s_terms);
  }
};
__global__ __launch_bounds__(kThreads) void k_p3p_init(P3pArgs A) {
  P3pInitBody::run(A);
}

constexpr int kP3pWaveSeg = kP3pMaxN / 4;  // elements one wave sorts when the four models run side by side
// More than kP3pMaxN correspondences (the reference has no limit, localization.cpp:479-509; a near-duplicate of a map
// frame with thousands of landmarks gets there): the residual sort of a hypothesis runs in a global-memory segment
// instead of LDS, and a round evaluates only the first kP3pLargeBatch hypotheses (that many segments exist).  Both
// kernels derive the round's size and the inlier-list stride from the same device-side n.
constexpr int kP3pLargeBatch = 64;
__device__ __forceinline__ int p3p_round_batch(int n, int batch) {
  return (n > kP3pMaxN && batch > kP3pLargeBatch) ? kP3pLargeBatch : batch;
}
// How many hypotheses the next round evaluates when the GPU is shared.  A round's hypotheses all sample from the current
// index set and everything after the first one that improves the model is thrown away.  Once sampling has switched to
// the best model's inliers, an iteration improves on the running minimum about as often as a new record appears in a
// random sequence: t iterations after the switch the next improvement is within the next m with probability m/(t+m).
// Evaluating 3t (at least 64) instead of a full batch finds it three times out of four and otherwise just advances; the
// acceptance rule is replayed exactly for any partition into rounds, so only the cost changes: ~2x fewer hypotheses
// evaluated for ~1.3x the rounds (+8 % queries per second at 12 in flight; the policy's constants hardly matter,
// profiles/r02_p3p_adaptive_policy.txt).  A query alone on the GPU keeps full batches: rounds are what its latency is
// made of.
__device__ __forceinline__ int p3p_next_batch_limit(const P3pArgs &A, int identity, long iter, long switch_iter, int n) {
  // (more than 512 correspondences: a model that passes the NFA filter costs a block-wide sort, and right after the
  // switch nearly every hypothesis passes -- its reference is the FIRST meaningful model -- while everything behind the
  // round's first improvement is thrown away: small rounds first, also for a query alone on the GPU)
  const bool costly = next_pow2(n) >= 1024;
  if ((!A.adaptive_batch && !costly) || identity) return 1 << 30;
  const long t = iter - switch_iter;
  const long m = (A.adapt_quarters * t) / 4;
  const long floor_ = costly && !A.adaptive_batch ? 16 : A.adapt_floor;
  return (int)(m < floor_ ? floor_ : (m > kP3pBatchMax ? kP3pBatchMax : m));
}
__device__ __forceinline__ size_t p3p_inl_stride(int n, int max_n) { return n > kP3pMaxN ? (size_t)max_n : (size_t)kP3pMaxN; }

// (the small members first, then idx, then key: the SMALL form of the round -- below -- allocates only kP3pSmallN entries
// per wave of idx and nothing of key)
struct P3pShared {
  double models[48];
  P3pPrep prep;  // Kneip's intermediates + the quartic's roots (lane 0), read by the four model lanes
  int nm;
  double red_nfa[kThreads / 64];
  int red_k[kThreads / 64];
  double red_err[kThreads / 64];
  uint32_t idx[kP3pMaxN];
  uint64_t key[kP3pMaxN];
};
// The SMALL form of a round (k_p3p_round_small): match sets of at most kP3pSmallN correspondences, whose models are
// sorted in at most kP3pSmallN / 64 registers per lane and never filtered.  k_p3p_round holds every form of the
// evaluation -- up to 16 (key, index, two table values) per lane for the register sort, the LDS and global sorts, the
// filter -- and the compiler gives it 248 VGPRs: ONE of its waves takes half a SIMD's register file, i.e. the place of
// four waves of the Hamming scan (68 VGPRs) that other queries in flight are running, for the 30-90 us a round lasts,
// and a round is 64-256 workgroups.  That, not K5's instructions (a tenth of the scan's) nor its LDS, is what K5 cost
// the other queries.  The small form is the same code with the large cases compiled out.  The host queues a query's
// rounds before it knows the set's size (Map::p3p_small_credit: the map's last queries were all small); a small round
// that finds a larger set returns at once, state untouched -- as if it had not been launched --, and
// ctx_resection_wait queues that query's rounds again in the full form.  The partition into rounds and the form of a
// launch change nothing in the result.
constexpr int kP3pSmallN = 512;

// A round's results go from the workgroup that computed them to the one that replays the round, which may sit on
// another XCD (another L2): they are written through (agent-scope stores), so that delivering them needs no L2
// write-back -- only the stores' completion: every storing wave runs s_waitcnt vmcnt(0) after its last store and the
// workgroup's barrier then orders all of those waits in front of the one lane that adds to the counter
// (p3p_round.body.inc; the barrier alone does not wait on vmcnt).

// ---------------------------------------------------------------------------------------------------
// The NFA filter.  A model of this round can only matter to the replay if its NFA is below the best NFA the round
// started from (st.min_nfa = B: the running minimum only goes down), and finding a model's NFA means sorting its n
// residuals -- 4 us for 256 correspondences in a wave's registers, 100 us for 2 000 in LDS, per model.  Whether a model
// CAN beat B is decided without sorting: with e_(k) the k-th smallest residual,
//     NFA(k) = loge0 + (logalpha0 + log10(e_(k) + eps)) (k - s) + logc_n[k] + logc_k[k]  <  B
//         <=>  e_(k) + eps  <  T_k := 10 ^ ((B - loge0 - logc_n[k] - logc_k[k]) / (k - s) - logalpha0)
//         <=>  #{ i : e_i + eps < T_k }  >=  k,
// so a model beats B only if SOME k has at least k residuals below T_k.  With T'_k = max_{j <= k} T_j (monotone, >= T_k:
// a weaker, still necessary condition) every residual is binned by binary search in the table, the bins are counted and
// prefix-summed, and the model passes iff some bin's running count reaches the bin's lowest k.  A model that fails
// reports NFA = +inf (exactly what the replay does with any NFA >= its running minimum: nothing); a model that passes
// is evaluated exactly as before, so every result the replay acts on has the bits it always had.  The table is
// computed with a safety margin of 1e-7 in the exponent (the exact test's rounding is ~1e-12), i.e. the filter never
// rejects a model whose exactly computed NFA is below B.  For n > 1024 the bins are 2 or 4 values of k wide (threshold
// of the bin's upper end against the bin's lowest k: weaker again, still necessary), which keeps the table at 8 KB.
// ---------------------------------------------------------------------------------------------------
constexpr int kP3pFilterBins = 1024;
struct P3pFilterLds {  // lives in P3pShared::key (32 KB), which is idle until a model is sorted in LDS
  double T[kP3pFilterBins];            // T' at each bin's upper end
  uint32_t hist[4][kP3pFilterBins + 4];
  double wave_max[kThreads / 64];
  int pass[4];
};
static_assert(sizeof(P3pFilterLds) <= sizeof(uint64_t) * kP3pMaxN, "the filter's tables live in the sort's key array");

__device__ __forceinline__ int p3p_filter_shift(int n) { return n <= kP3pFilterBins ? 0 : (n <= 2 * kP3pFilterBins ? 1 : 2); }

// raw thresholds: the maximum of T_k over the k of each bin, bins owned by the calling threads (any subset of the
// workgroup: thread `t` of `nt` takes bins t, t + nt, ...)
__device__ __forceinline__ void p3p_filter_raw(P3pFilterLds &F, int n, int s, double B, double logalpha0, double loge0,
                                               const float *__restrict__ logc_n, const float *__restrict__ logc_k, int t,
                                               int nt) {
  const int sh = p3p_filter_shift(n), w = 1 << sh;
  const int nb = (n >> sh) + 1;
  for (int j = t; j < nb; j += nt) {
    double m = 0.0;
    for (int k = j << sh; k < (j << sh) + w; ++k)
      if (k > s && k <= n) {
        const double x = (B - loge0 - (double)logc_n[k] - (double)logc_k[k]) / (double)(k - s) - logalpha0 + 1e-7;
        const double T = x > 300.0 ? pos_inf() : exp10(x);
        m = T > m ? T : m;
      }
    F.T[j] = m;
  }
}

// T' = running maximum over the bins; every thread of the workgroup calls it (two barriers inside)
__device__ __forceinline__ void p3p_filter_scan(P3pFilterLds &F, int n) {
  const int sh = p3p_filter_shift(n);
  const int nb = (n >> sh) + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int per = (kP3pFilterBins + 4 + kThreads - 1) / kThreads;  // bins per thread, contiguous
  double v[per];
  double run = 0.0;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    const int j = tid * per + i;
    const double x = j < nb ? F.T[j] : 0.0;
    run = x > run ? x : run;
    v[i] = run;
  }
  double inc = run;  // inclusive running maximum across the wave's lanes
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(inc, off, 64);
    if (lane >= off) inc = o > inc ? o : inc;
  }
  double before = __shfl_up(inc, 1, 64);
  if (lane == 0) before = 0.0;
  if (lane == 63) F.wave_max[wv] = inc;
  __syncthreads();
  for (int q = 0; q < wv; ++q) before = F.wave_max[q] > before ? F.wave_max[q] : before;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    const int j = tid * per + i;
    if (j < nb) F.T[j] = v[i] > before ? v[i] : before;
  }
  __syncthreads();
}

// one wave, one model: can it beat B?  (every lane returns the verdict)
__device__ __forceinline__ bool p3p_filter_model(P3pFilterLds &F, int wv, const double (&M)[12], int n, int s,
                                                 const double *__restrict__ pt3d, const double *__restrict__ xn) {
  const int lane = threadIdx.x & 63;
  const int sh = p3p_filter_shift(n);
  const int nb = (n >> sh) + 1;
  uint32_t *h = F.hist[wv];
  for (int j = lane; j < nb; j += 64) h[j] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int steps = 0;
  while ((1 << steps) < nb + 1) ++steps;
  // four elements of a lane at a time: their residuals (two f64 divisions each) and their searches (dependent LDS reads)
  // interleave
  constexpr int U = 4;
  for (int p0 = 0; p0 < n; p0 += 64 * U) {
    double r[U];
    int lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + 64 * u + lane;
      const int pc = p < n ? p : n - 1;
      const double e = err_resection(M, pt3d[3 * pc], pt3d[3 * pc + 1], pt3d[3 * pc + 2], xn[2 * pc], xn[2 * pc + 1]);
      r[u] = e + (double)FLT_EPSILON;
      lo[u] = 0;
      hi[u] = nb;
    }
    // first bin whose threshold exceeds r (nb: none); T is non-decreasing
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mid = (lo[u] + hi[u]) >> 1;
        const bool below = mid < nb && r[u] < F.T[mid < nb ? mid : nb - 1];
        const bool open = lo[u] < hi[u];
        hi[u] = (open && below) ? mid : hi[u];
        lo[u] = (open && !below) ? mid + 1 : lo[u];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (p0 + 64 * u + lane < n && lo[u] < nb) atomicAdd(&h[lo[u]], 1u);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // running counts: each lane a contiguous run of bins, then across the lanes
  constexpr int per = (kP3pFilterBins + 4 + 63) / 64;
  uint32_t c[per];
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    const int j = lane * per + i;
    sum += j < nb ? h[j] : 0u;
    c[i] = sum;
  }
  uint32_t inc = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  const uint32_t before = inc - sum;
  bool any = false;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    const int j = lane * per + i;
    const int k_lo = (j << sh) > s + 1 ? (j << sh) : s + 1;          // lowest k of the bin that NFA ranges over
    const int k_hi = (j << sh) + (1 << sh) - 1;
    if (j < nb && k_hi > s && k_lo <= n && before + c[i] >= (uint32_t)k_lo) any = true;
  }
  return __ballot(any) != 0ull;
}

// one model, the whole workgroup: can it beat B?  (p3p_filter_model with the four waves sharing the elements and ONE
// table of counts; every thread returns the verdict; two barriers inside)
__device__ __forceinline__ bool p3p_filter_model_block(P3pFilterLds &F, const double (&M)[12], int n, int s,
                                                       const double *__restrict__ pt3d, const double *__restrict__ xn) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int sh = p3p_filter_shift(n);
  const int nb = (n >> sh) + 1;
  uint32_t *h = F.hist[0];
  for (int j = tid; j < nb; j += kThreads) h[j] = 0u;
  __syncthreads();
  int steps = 0;
  while ((1 << steps) < nb + 1) ++steps;
  constexpr int U = 4;
  for (int p0 = 0; p0 < n; p0 += kThreads * U) {
    double r[U];
    int lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + kThreads * u + tid;
      const int pc = p < n ? p : n - 1;
      const double e = err_resection(M, pt3d[3 * pc], pt3d[3 * pc + 1], pt3d[3 * pc + 2], xn[2 * pc], xn[2 * pc + 1]);
      r[u] = e + (double)FLT_EPSILON;
      lo[u] = 0;
      hi[u] = nb;
    }
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mid = (lo[u] + hi[u]) >> 1;
        const bool below = mid < nb && r[u] < F.T[mid < nb ? mid : nb - 1];
        const bool open = lo[u] < hi[u];
        hi[u] = (open && below) ? mid : hi[u];
        lo[u] = (open && !below) ? mid + 1 : lo[u];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (p0 + kThreads * u + tid < n && lo[u] < nb) atomicAdd(&h[lo[u]], 1u);
  }
  __syncthreads();
  // every wave walks the same table of counts and reaches the same verdict
  constexpr int per = (kP3pFilterBins + 4 + 63) / 64;
  uint32_t c[per];
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    const int j = lane * per + i;
    sum += j < nb ? h[j] : 0u;
    c[i] = sum;
  }
  uint32_t inc = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  const uint32_t before = inc - sum;
  bool any = false;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    const int j = lane * per + i;
    const int k_lo = (j << sh) > s + 1 ? (j << sh) : s + 1;
    const int k_hi = (j << sh) + (1 << sh) - 1;
    if (j < nb && k_hi > s && k_lo <= n && before + c[i] >= (uint32_t)k_lo) any = true;
  }
  return __ballot(any) != 0ull;
}

// one model evaluated by one wave in its registers: residuals of elements r * 64 + lane, register sort, NFA minimum;
// the sorted indices go to iw[] (LDS).  -> r (NFA, k) and the k-th smallest residual.
// first_hit (or null): P3pState::first_hit -- the wave gives up between its steps once an EARLIER hypothesis of the round
// is known to change the index set (the replay will not look at this one); the value is read by every lane from the same
// address and taken from the first lane, so the wave decides as one.
__device__ __forceinline__ bool p3p_overtaken(const unsigned *first_hit, int b) {
  if (!first_hit) return false;
  const unsigned v = __hip_atomic_load(first_hit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v) < (unsigned)b;
}
template <int E>
__device__ __forceinline__ void p3p_eval_regs(const double (&M)[12], NfaBest &r, double &r_err, const double *__restrict__ pt3d,
                                              const double *__restrict__ xn, const float *__restrict__ logc_n,
                                              const float *__restrict__ logc_k, uint32_t *iw, int lane, int n, double logalpha0,
                                              double loge0, int stamp_round, int b, const unsigned *first_hit = nullptr) {
  uint64_t key[E];
  uint32_t idx[E];
  float cn[E], ck[E];
  if (p3p_overtaken(first_hit, b)) return;  // (r stays "no model")
  nfa_tables_fetch<E>(cn, ck, n, logc_n, logc_k);  // in flight during the residuals and the sort
  // the wave is alone on its SIMD: the E residuals of a lane are computed without branches (clamped index, then a
  // select) so that their dependent f64 chains -- two divisions each -- interleave
#pragma unroll
  for (int rr = 0; rr < E; ++rr) {
    const int p = (rr << 6) + lane;
    const int pc = p < n ? p : n - 1;
    const double e = err_resection(M, pt3d[3 * pc], pt3d[3 * pc + 1], pt3d[3 * pc + 2], xn[2 * pc], xn[2 * pc + 1]);
    key[rr] = p < n ? d2u(e) : ~0ull;
    idx[rr] = (uint32_t)p;
  }
  STAMP_P3P(stamp_round, b, 6);
  if (p3p_overtaken(first_hit, b)) return;
  wave_sort_fast<E>(key, idx, iw);
  STAMP_P3P(stamp_round, b, 7);
  if (p3p_overtaken(first_hit, b)) return;
#pragma unroll
  for (int rr = 0; rr < E; ++rr) iw[(rr << 6) + lane] = idx[rr];
  r = best_nfa_regs_ilp<E>(key, n, 3, pos_inf(), logalpha0, 1.0, loge0, cn, ck);
  if (r.k != 0x7FFFFFFF) r_err = u2d(sorted_key_at<E>(key, r.k - 1));
}

// entries of a sorted run of L keys (L a power of two) that come before x: those < x, or (le) those <= x
template <int L>
__device__ __forceinline__ int run_count_before(const uint64_t *run, uint64_t x, bool le) {
  int lo = 0, hi = L;
  constexpr int kSteps = (L == 64 ? 7 : L == 128 ? 8 : 9);  // L + 1 possible answers
  static_assert(L == 64 || L == 128 || L == 256, "run length");
#pragma unroll
  for (int it = 0; it < kSteps; ++it) {
    const int mid = (lo + hi) >> 1;
    const uint64_t v = run[mid < L ? mid : L - 1];
    const bool open = lo < hi;
    const bool before = le ? (v <= x) : (v < x);
    lo = (open && before) ? mid + 1 : lo;
    hi = (open && !before) ? mid : hi;
  }
  return lo;
}

// One model evaluated by the WHOLE workgroup (four waves; a "single" launch: one workgroup per model): wave j computes and
// sorts the residuals of elements 64 E j .. 64 E (j + 1) - 1 in its registers (wave_sort_fast, E <= 4 per lane), the four
// sorted runs are merged by rank through LDS -- an element's final position is its position in its own run plus, per
// other run, the number of that run's elements that come before it in the (key, index) order: runs of lower element
// indices win ties, which is the order of the one-wave sort -- and every thread takes its share of the NFA scan.  Up to
// 1 024 correspondences; the same residuals, order, candidates and first minimum as p3p_eval_regs, in a quarter of the
// dependent chain: one wave with 16 elements per lane sorts 1 024 residuals in ~16 us, four with 4 each in ~4 + ~2 for the
// merge.  `runs`: 256 E u64 of LDS, `fidx`: 256 E u32 (the sort's parking space, then the merged element indices).
// Every thread of the workgroup calls it (barriers inside) and gets the result.
template <int E>
__device__ __forceinline__ void p3p_eval_coop4(const double (&M)[12], NfaBest &res, double &res_err,
                                               const double *__restrict__ pt3d, const double *__restrict__ xn,
                                               const float *__restrict__ logc_n, const float *__restrict__ logc_k, int n,
                                               double logalpha0, double loge0, uint64_t *runs, uint32_t *fidx, double *red_nfa,
                                               int *red_k) {
  constexpr int L = 64 * E;
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int base = j * L;
  uint64_t key[E];
  uint32_t idx[E];
  float cn[E], ck[E];
#pragma unroll
  for (int rr = 0; rr < E; ++rr) {  // (the table entries of the merged positions this lane scans: in flight meanwhile)
    const int kk = base + (rr << 6) + lane + 1;
    cn[rr] = kk <= n ? logc_n[kk] : 0.0f;
    ck[rr] = kk <= n ? logc_k[kk] : 0.0f;
  }
#pragma unroll
  for (int rr = 0; rr < E; ++rr) {
    const int p = base + (rr << 6) + lane;
    const int pc = p < n ? p : n - 1;
    const double e = err_resection(M, pt3d[3 * pc], pt3d[3 * pc + 1], pt3d[3 * pc + 2], xn[2 * pc], xn[2 * pc + 1]);
    key[rr] = p < n ? d2u(e) : ~0ull;
    idx[rr] = 0u;
  }
  wave_sort_fast<E>(key, idx, fidx + base);
#pragma unroll
  for (int rr = 0; rr < E; ++rr) runs[base + (rr << 6) + lane] = key[rr];
  __syncthreads();
  int rank[E];
#pragma unroll
  for (int rr = 0; rr < E; ++rr) rank[rr] = (rr << 6) + lane;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u == j) continue;  // (uniform over the wave)
#pragma unroll
    for (int rr = 0; rr < E; ++rr) rank[rr] += run_count_before<L>(runs + u * L, key[rr], u < j);
  }
  __syncthreads();  // every rank is known: the runs may be overwritten
#pragma unroll
  for (int rr = 0; rr < E; ++rr) {
    runs[rank[rr]] = key[rr];
    fidx[rank[rr]] = (uint32_t)base + idx[rr];
  }
  __syncthreads();
  constexpr int s = 3;
  double lb = pos_inf();
  int lk = 0x7FFFFFFF;
  double l10[E];
  uint64_t mk[E];
#pragma unroll
  for (int rr = 0; rr < E; ++rr) mk[rr] = runs[base + (rr << 6) + lane];
#pragma unroll
  for (int rr = 0; rr < E; ++rr) l10[rr] = det_log10_inline(u2d(mk[rr]) + (double)FLT_EPSILON);
#pragma unroll
  for (int rr = 0; rr < E; ++rr) {
    const int kk = base + (rr << 6) + lane + 1;
    if (kk > s && kk <= n) {
      const double logalpha = logalpha0 + 1.0 * l10[rr];
      const double nfa = loge0 + logalpha * (double)(kk - s) + (double)cn[rr] + (double)ck[rr];
      if (nfa < lb) {
        lb = nfa;
        lk = kk;
      }
    }
  }
  const NfaBest wr = wave_reduce_nfa(lb, lk);
  if (lane == 0) {
    red_nfa[j] = wr.nfa;
    red_k[j] = wr.k;
  }
  __syncthreads();
  NfaBest r{red_nfa[0], red_k[0]};
#pragma unroll
  for (int u = 1; u < 4; ++u)
    if (red_nfa[u] < r.nfa || (red_nfa[u] == r.nfa && red_k[u] < r.k)) {
      r.nfa = red_nfa[u];
      r.k = red_k[u];
    }
  res = r;
  res_err = r.k != 0x7FFFFFFF ? u2d(runs[r.k - 1]) : pos_inf();
}

// The same for 1 025 .. 4 096 correspondences: NR = P / 1 024 runs of 256 per wave, sorted one after the other in the wave's
// registers and parked in LDS; then every element's rank among the 4 NR runs by binary search (a run of lower element
// indices wins ties), the scatter into the merged order, and the NFA scan, P / 256 positions per thread.  Replaces the
// block-wide bitonic sort in LDS for one-workgroup-per-model launches: 66 (78) barrier-separated stages over 2 048
// (4 096) elements, 110 us (250 us) per model, against ~25 us (~45 us) this way.  runs: P u64 of LDS, fidx: P u32.
template <int NR>
__device__ __forceinline__ void p3p_eval_coop4_multi(const double (&M)[12], NfaBest &res, double &res_err,
                                                     const double *__restrict__ pt3d, const double *__restrict__ xn,
                                                     const float *__restrict__ logc_n, const float *__restrict__ logc_k, int n,
                                                     double logalpha0, double loge0, uint64_t *runs, uint32_t *fidx,
                                                     double *red_nfa, int *red_k) {
  constexpr int E = 4, L = 256, R = 4 * NR;
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  uint64_t key[NR][E];
  uint32_t gidx[NR][E];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int base = (j * NR + q) * L;
    uint32_t idx[E];
#pragma unroll
    for (int rr = 0; rr < E; ++rr) {
      const int p = base + (rr << 6) + lane;
      const int pc = p < n ? p : n - 1;
      const double e = err_resection(M, pt3d[3 * pc], pt3d[3 * pc + 1], pt3d[3 * pc + 2], xn[2 * pc], xn[2 * pc + 1]);
      key[q][rr] = p < n ? d2u(e) : ~0ull;
      idx[rr] = 0u;
    }
    wave_sort_fast<E>(key[q], idx, fidx + base);
#pragma unroll
    for (int rr = 0; rr < E; ++rr) {
      runs[base + (rr << 6) + lane] = key[q][rr];
      gidx[q][rr] = (uint32_t)base + idx[rr];
    }
  }
  __syncthreads();
  int rank[NR][E];
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int rr = 0; rr < E; ++rr) rank[q][rr] = (rr << 6) + lane;
  for (int u = 0; u < R; ++u) {
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int mine = j * NR + q;
      if (u == mine) continue;  // (uniform over the wave)
#pragma unroll
      for (int rr = 0; rr < E; ++rr) rank[q][rr] += run_count_before<L>(runs + u * L, key[q][rr], u < mine);
    }
  }
  __syncthreads();  // every rank is known: the runs may be overwritten
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int rr = 0; rr < E; ++rr) {
      runs[rank[q][rr]] = key[q][rr];
      fidx[rank[q][rr]] = gidx[q][rr];
    }
  __syncthreads();
  constexpr int s = 3;
  double lb = pos_inf();
  int lk = 0x7FFFFFFF;
  // merged positions tid, tid + 256, ...: k = position + 1 (any split of the candidates gives the same first minimum)
  for (int pos = threadIdx.x; pos < n; pos += kThreads) {
    const int kk = pos + 1;
    if (kk > s) {
      const double ek = u2d(runs[pos]);
      const double logalpha = logalpha0 + 1.0 * det_log10_inline(ek + (double)FLT_EPSILON);
      const double nfa = loge0 + logalpha * (double)(kk - s) + (double)logc_n[kk] + (double)logc_k[kk];
      if (nfa < lb || (nfa == lb && kk < lk)) {
        lb = nfa;
        lk = kk;
      }
    }
  }
  const NfaBest wr = wave_reduce_nfa(lb, lk);
  if (lane == 0) {
    red_nfa[j] = wr.nfa;
    red_k[j] = wr.k;
  }
  __syncthreads();
  NfaBest r{red_nfa[0], red_k[0]};
#pragma unroll
  for (int u = 1; u < 4; ++u)
    if (red_nfa[u] < r.nfa || (red_nfa[u] == r.nfa && red_k[u] < r.k)) {
      r.nfa = red_nfa[u];
      r.k = red_k[u];
    }
  res = r;
  res_err = r.k != 0x7FFFFFFF ? u2d(runs[r.k - 1]) : pos_inf();
}

// one hypothesis: sample, solve, evaluate the (up to 4) models, leave the best one's NFA / inliers / model in the
// round's result arrays.  Executed by one workgroup of k_p3p_round -- or, in a WIDE launch (four workgroups per
// hypothesis: model m of hypothesis b is workgroup m * batch + b) and from 513 correspondences on, by four, one model each: a model's residuals are
// two f64 divisions per correspondence and a lone wave issues one instruction every ~5 cycles, so with a wave per model
// 2 000 correspondences are 32 per lane (45 us for the filter alone, 105 us for a model that has to be sorted), and a
// hypothesis's models wait for each other; with a workgroup per model it is 8 per thread, four times as many waves on the
// compute unit, and the four models of a hypothesis side by side.  Results go to slot 4 b + m; the replay takes the
// best model of each hypothesis (the first on ties, as the sequential loop over a hypothesis's models does).
template <int kForm>  // 0: the full form (k_p3p_round), 1: small (k_p3p_round_small)
__device__ __forceinline__ void p3p_eval_hypothesis(const P3pArgs &A, int batch, int wide, unsigned char *smem_raw, int b,
                                                    int mdl) {
  constexpr bool kSmall = kForm == 1;
  const P3pState &st = *A.state;
  const int n = st.n;
  if (b >= p3p_round_batch(n, batch) || b >= st.batch_limit) return;
  const long it = (long)st.iter + b;
  if (it >= st.n_iter) return;
  P3pShared &S = *reinterpret_cast<P3pShared *>(smem_raw);
  const int tid = threadIdx.x;
  constexpr int s = 3;
  const int P = next_pow2(n);
  // An earlier hypothesis of this round is already known to change the index set: whatever this one finds, the replay
  // will not look at it (it stops at the FIRST such hypothesis, which is that one or an earlier one).  Nothing is
  // evaluated, the slot reports "no model".  Only workgroups that start late see this -- i.e. while the GPU is shared.
  // (ONE lane reads the word and the workgroup decides on that value: the waves' own reads could straddle an update and
  // part of the workgroup would leave in front of the barriers below)
  // (the decision is taken behind the barrier that follows the solver / the fetch of the prepared models, below)
  __shared__ unsigned s_first_hit;
  // one model per workgroup: a wide launch with enough correspondences (and the LDS forms: n <= kP3pMaxN)
  const bool single = !kSmall && wide && n <= kP3pMaxN;
  // (a wide launch that is not in single mode runs workgroups 0 .. batch - 1 only -- p3p_round.body.inc -- and is a plain
  // round: slot b, so that the inlier lists of a set above kP3pMaxN, whose stride is max_n, stay inside the 64 lists
  // ctx_p3p_reserve sizes them for)
  const int slot = single ? 4 * b + mdl : b;
#ifdef SFMLOC_STAMPS
  const int stamp_round = st.rounds;
#endif
  STAMP_P3P(stamp_round, b, 0);
  const size_t inl_stride = p3p_inl_stride(n, A.max_n);
  // where the block-wide sort of this hypothesis runs
  uint64_t *const skey = n > kP3pMaxN ? A.ws_key + (size_t)b * A.max_n : S.key;
  uint32_t *const sidx = n > kP3pMaxN ? A.ws_idx + (size_t)b * A.max_n : S.idx;
  // (register path of the full form: up to 256 correspondences, 4 per lane.  The 8- and 16-per-lane instantiations it used
  // to hold for plain launches on 257 .. 1 024 correspondences made the kernel 248 VGPRs for every launch; such sets take
  // the small form, a wide launch -- p3p_eval_coop4 -- or, in a plain launch of this form, the block-wide sort)
  const bool fast = kSmall || P <= 256;
  constexpr int wave_seg = kSmall ? kP3pSmallN : kP3pWaveSeg;  // entries of S.idx a wave owns
  const double logalpha0 = det_log10(3.14159265358979323846);
  const double loge0 = det_log10(4.0 * (double)(n - s));
  // the NFA filter (above): only once a model exists, only while its tables fit the idle part of the LDS
  const double nfa_to_beat = st.min_nfa;
  // (from 257 correspondences on: below that a model's register sort costs about what the filter does)
  // (the small form has no LDS for the tables: its models are sorted, which gives the same result)
  const bool filter = !kSmall && A.nfa_filter && nfa_to_beat < pos_inf() && n <= kP3pMaxN && P >= A.nfa_filter_min_p;
  P3pFilterLds &F = *reinterpret_cast<P3pFilterLds *>(S.key);
  // "Prepared ahead" (p3p_prepare_ahead, below): the models of this hypothesis may be there already
  const bool have = st.prep_iter == st.iter && b < st.prep_n;
  if (filter) {
    if (have) p3p_filter_raw(F, n, s, nfa_to_beat, logalpha0, loge0, A.logc_n, A.logc_k, tid, kThreads);
    else if (tid >= 64)  // (wave 0 is busy: its lane 0 solves the P3P below)
      p3p_filter_raw(F, n, s, nfa_to_beat, logalpha0, loge0, A.logc_n, A.logc_k, tid - 64, kThreads - 64);
  }
  if (have) {
    if (tid < 48) S.models[tid] = A.prep_models[48 * (size_t)b + tid];
    if (tid == 0) S.nm = A.prep_nm[b];
  } else if (tid == 0) {
    int32_t smp[3];
    ac_sample<3>(st.identity ? nullptr : A.vec_index, st.n_index, A.seed, STAGE_P3P, A.stream, (uint32_t)it, smp);
    double x[6], X[9];
    for (int i = 0; i < 3; ++i) {
      x[2 * i] = A.xn[2 * smp[i]];
      x[2 * i + 1] = A.xn[2 * smp[i] + 1];
      X[3 * i] = A.pt3d[3 * smp[i]];
      X[3 * i + 1] = A.pt3d[3 * smp[i] + 1];
      X[3 * i + 2] = A.pt3d[3 * smp[i] + 2];
    }
    STAMP_P3P(stamp_round, b, 1);
    S.nm = p3p_kneip_prepare(x, X, S.prep);
  }
  if (tid == 0)
    s_first_hit = A.skip_overtaken ? __hip_atomic_load(&A.state->first_hit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
  __syncthreads();
  STAMP_P3P(stamp_round, b, 2);
  if (s_first_hit < (unsigned)b) {  // (uniform: one lane's read, seen by all behind the barrier)
    if (tid == 0) store_through(A.hyp_nfa + slot, pos_inf());
    return;
  }
  const int nm = S.nm;
  if (single && mdl >= nm) {  // (uniform over the workgroup) no such root
    if (tid == 0) store_through(A.hyp_nfa + slot, pos_inf());
    return;
  }
  if (!have && tid < nm) p3p_kneip_model(S.prep, tid, S.models + 12 * tid);  // the four roots' models side by side
  if (filter) p3p_filter_scan(F, n);  // (two barriers inside: the models are visible behind them too)
  else __syncthreads();
  STAMP_P3P(stamp_round, b, 3);
  if (!fast || single) {
    STAMP_P3P(stamp_round, b, 6);   // (the register path stamps 6 / 7 around its sort)
  }
  // which models have to be evaluated: bit k = model k (one model per workgroup: bit mdl only)
  int pass_mask = single ? (1 << mdl) : 0xF;
  if (filter) {
    if (single) {
      double M[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) M[q] = S.models[12 * mdl + q];
      pass_mask = p3p_filter_model_block(F, M, n, s, A.pt3d, A.xn) ? (1 << mdl) : 0;
      __syncthreads();  // F (in S.key) is dead from here on
    } else {
      const int wv = tid >> 6;
      bool pass_mine = false;
      if (wv < nm) {
        double M[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) M[q] = S.models[12 * wv + q];
        pass_mine = p3p_filter_model(F, wv, M, n, s, A.pt3d, A.xn);
      }
      if ((tid & 63) == 0) F.pass[wv] = pass_mine ? 1 : 0;
      __syncthreads();
      pass_mask = 0;
      for (int k = 0; k < 4; ++k) pass_mask |= F.pass[k] << k;
      __syncthreads();  // F (in S.key) is dead from here on: the LDS sort may overwrite it
    }
  }
  if (!fast || single) {
    STAMP_P3P(stamp_round, b, 7);   // LDS path: 3 -> 6 = table scan, 6 -> 7 = the filter, 7 -> 4 = the models that passed
  }
  double best = pos_inf();
  int best_k = 0, best_m = -1;
  double best_err = pos_inf();
  if (single) {
    // one model per workgroup: the four waves share the model (p3p_eval_coop4 / _multi; every set the LDS forms hold)
    // (behind the filter once more: has an earlier hypothesis been found to change the index set meanwhile?)
    if (A.skip_overtaken) {  // (uniform)
      __syncthreads();
      if (tid == 0) s_first_hit = __hip_atomic_load(&A.state->first_hit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (s_first_hit < (unsigned)b) pass_mask = 0;
    }
    if ((pass_mask >> mdl) & 1) {  // (uniform over the workgroup)
      double M[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) M[q] = S.models[12 * mdl + q];
      NfaBest r{pos_inf(), 0x7FFFFFFF};
      double r_err = pos_inf();
      const double *pt3d = A.pt3d, *xn = A.xn;
      const float *logc_n = A.logc_n, *logc_k = A.logc_k;
      if (P <= 256) p3p_eval_coop4<1>(M, r, r_err, pt3d, xn, logc_n, logc_k, n, logalpha0, loge0, S.key, S.idx, S.red_nfa, S.red_k);
      else if (P == 512) p3p_eval_coop4<2>(M, r, r_err, pt3d, xn, logc_n, logc_k, n, logalpha0, loge0, S.key, S.idx, S.red_nfa, S.red_k);
      else if (P == 1024) p3p_eval_coop4<4>(M, r, r_err, pt3d, xn, logc_n, logc_k, n, logalpha0, loge0, S.key, S.idx, S.red_nfa, S.red_k);
      else if (P == 2048) p3p_eval_coop4_multi<2>(M, r, r_err, pt3d, xn, logc_n, logc_k, n, logalpha0, loge0, S.key, S.idx, S.red_nfa, S.red_k);
      else p3p_eval_coop4_multi<4>(M, r, r_err, pt3d, xn, logc_n, logc_k, n, logalpha0, loge0, S.key, S.idx, S.red_nfa, S.red_k);
      if (r.nfa < best) {
        best = r.nfa;
        best_k = r.k;
        best_m = mdl;
        best_err = r_err;
        int32_t *dst = A.hyp_inl + (size_t)slot * inl_stride;
        for (int p = tid; p < best_k; p += kThreads) store_through(dst + p, (int32_t)S.idx[p]);
      }
    }
    STAMP_P3P(stamp_round, b, 4);
  } else if (fast) {
    // register path: a model is evaluated by ONE wave, residuals sorted in its registers; the (up to 4) models of the
    // hypothesis side by side, one wave each -- or (one model per workgroup) wave 0 takes the workgroup's model
    const int wv = tid >> 6, lane = tid & 63;
    // (never in single mode: a one-model-per-workgroup launch took the branch above)
    const int my_model = wv;
    const bool mine = wv < nm;
    uint32_t *iw = S.idx + (size_t)wv * wave_seg;
    NfaBest r{pos_inf(), 0x7FFFFFFF};
    double r_err = pos_inf();
    if (mine && ((pass_mask >> my_model) & 1)) {
      double M[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) M[q] = S.models[12 * my_model + q];
      // (the lambda must not capture the kernel-argument struct: that would put all of it on the stack)
      const double *pt3d = A.pt3d, *xn = A.xn;
      const float *logc_n = A.logc_n, *logc_k = A.logc_k;
      if constexpr (kSmall) {
        const unsigned *first_hit = A.skip_overtaken ? &A.state->first_hit : nullptr;
        // (inlined: 152 VGPRs for the whole kernel, against 212 with the evaluation as a called function)
#ifndef SFMLOC_STAMPS
        const int stamp_round = 0;
#endif
        switch (P >> 6) {
          case 1: p3p_eval_regs<1>(M, r, r_err, pt3d, xn, logc_n, logc_k, iw, lane, n, logalpha0, loge0, stamp_round, b, first_hit); break;
          case 2: p3p_eval_regs<2>(M, r, r_err, pt3d, xn, logc_n, logc_k, iw, lane, n, logalpha0, loge0, stamp_round, b, first_hit); break;
          case 4: p3p_eval_regs<4>(M, r, r_err, pt3d, xn, logc_n, logc_k, iw, lane, n, logalpha0, loge0, stamp_round, b, first_hit); break;
          default: p3p_eval_regs<8>(M, r, r_err, pt3d, xn, logc_n, logc_k, iw, lane, n, logalpha0, loge0, stamp_round, b, first_hit); break;
        }
      } else {
        // (a lambda the compiler keeps out of line -- the text of p3p_eval_regs once more: with a call to that function in
        // it the compiler inlines all five instantiations and the kernel needs 256 VGPRs and a SIMD to itself)
#ifdef SFMLOC_STAMPS
        auto run = [&M, &r, &r_err, pt3d, xn, logc_n, logc_k, iw, lane, n, logalpha0, loge0, stamp_round, b](auto e_tag) {
#else
        auto run = [&M, &r, &r_err, pt3d, xn, logc_n, logc_k, iw, lane, n, logalpha0, loge0](auto e_tag) {
#endif
          constexpr int E = decltype(e_tag)::value;
          uint64_t key[E];
          uint32_t idx[E];
          float cn[E], ck[E];
          nfa_tables_fetch<E>(cn, ck, n, logc_n, logc_k);  // in flight during the residuals and the sort
#pragma unroll
          for (int rr = 0; rr < E; ++rr) {
            const int p = (rr << 6) + lane;
            const int pc = p < n ? p : n - 1;
            const double e = err_resection(M, pt3d[3 * pc], pt3d[3 * pc + 1], pt3d[3 * pc + 2], xn[2 * pc], xn[2 * pc + 1]);
            key[rr] = p < n ? d2u(e) : ~0ull;
            idx[rr] = (uint32_t)p;
          }
          STAMP_P3P(stamp_round, b, 6);
          wave_sort_fast<E>(key, idx, iw);
          STAMP_P3P(stamp_round, b, 7);
#pragma unroll
          for (int rr = 0; rr < E; ++rr) iw[(rr << 6) + lane] = idx[rr];
          r = best_nfa_regs_ilp<E>(key, n, 3, pos_inf(), logalpha0, 1.0, loge0, cn, ck);
          if (r.k != 0x7FFFFFFF) r_err = u2d(sorted_key_at<E>(key, r.k - 1));
        };
        switch (P >> 6) {
          case 1: run(std::integral_constant<int, 1>{}); break;
          case 2: run(std::integral_constant<int, 2>{}); break;
          default: run(std::integral_constant<int, 4>{}); break;
        }
      }
    }
    if (lane == 0) {
      S.red_nfa[wv] = r.nfa;
      S.red_k[wv] = r.k;
      S.red_err[wv] = r_err;
    }
    __syncthreads();
    STAMP_P3P(stamp_round, b, 4);
    for (int k = 0; k < nm; ++k)
      if (S.red_nfa[k] < best) {  // strict: the first model of the hypothesis wins ties, as the sequential loop does
        best = S.red_nfa[k];
        best_k = S.red_k[k];
        best_m = k;
      }
    if (best_m >= 0) {
      const int w_best = best_m;
      best_err = S.red_err[w_best];
      int32_t *dst = A.hyp_inl + (size_t)slot * inl_stride;
      const uint32_t *src = S.idx + (size_t)w_best * wave_seg;
      for (int p = tid; p < best_k; p += kThreads) store_through(dst + p, (int32_t)src[p]);
    }
  } else
  for (int k = 0; k < nm; ++k) {
    if (!((pass_mask >> k) & 1)) continue;  // (uniform over the workgroup)
    double M[12];
    for (int q = 0; q < 12; ++q) M[q] = S.models[12 * k + q];
    __syncthreads();
    for (int p = tid; p < P; p += kThreads) {
      uint64_t kv = ~0ull;
      if (p < n)
        kv = d2u(err_resection(M, A.pt3d[3 * p], A.pt3d[3 * p + 1], A.pt3d[3 * p + 2], A.xn[2 * p], A.xn[2 * p + 1]));
      skey[p] = kv;
      sidx[p] = (uint32_t)p;
    }
    __syncthreads();
    bitonic_sort(skey, sidx, P);
    const NfaBest r = best_nfa_block(skey, n, s, pos_inf(), logalpha0, 1.0, loge0, A.logc_n, A.logc_k, S.red_nfa,
                                     S.red_k);
    if (r.nfa < best) {  // strict: the first model of the hypothesis wins ties, as the sequential loop does
      best = r.nfa;
      best_k = r.k;
      best_m = k;
      best_err = u2d(skey[r.k - 1]);
      int32_t *dst = A.hyp_inl + (size_t)slot * inl_stride;
      for (int p = tid; p < best_k; p += kThreads) store_through(dst + p, (int32_t)sidx[p]);
    }
  }
  if (!fast) {
    STAMP_P3P(stamp_round, b, 4);
  }
  if (tid == 0) {
    // (min_nfa < 0 from the switch on, so every improvement changes the index set; before it only a meaningful model does)
    if (best < nfa_to_beat && best < 0.0) atomicMin(&A.state->first_hit, (unsigned)b);
    store_through(A.hyp_nfa + slot, best);
    store_through(A.hyp_k + slot, best_k);
    store_through(A.hyp_err + slot, best_err);
    if (best_m >= 0)
      for (int q = 0; q < 12; ++q) store_through(A.hyp_model + 12 * slot + q, S.models[12 * best_m + q]);
  }
  STAMP_P3P(stamp_round, b, 5);
}

// ---------------------------------------------------------------------------------------------------
// K6 (north-star extension A13; absent from the reference, which never calls RefinePose): non-linear refinement
// of [R|t] on the inliers, intrinsics fixed -- Levenberg-Marquardt on the squared reprojection error with a left
// rotation increment.  The normal equations [J r]^T [J r] (7x7: J^T J, J^T r, r^T r) are accumulated on the
// matrix cores: v_mfma_f64_16x16x4_f64 takes four rows of M = [J | r] (padded to 16 columns) per instruction,
// and because A[i][k] = M[4s+k][i] and B[k][j] = M[4s+k][j] sit in the same lane (i = j = lane & 15, k = lane >> 4)
// one f64 per lane feeds both operands.  Executed by the whole workgroup (barriers), arithmetic by wave 0.
// ---------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct RefineShared {
  double H[7][8];
  double delta[6];
};

__device__ void normal_equations_wave0(const double *pt2d, const double *pt3d, const int32_t *inl, int n, double f,
                                       double ppx, double ppy, const double *Rm, const double *tv,
                                       RefineShared &S) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int k = tid >> 4, col = tid & 15;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int rows = 2 * n;
    for (int s4 = 0; s4 < rows; s4 += 4) {
      const int row = s4 + k;
      double m = 0.0;
      if (row < rows && col < 7) {
        const int p = inl[row >> 1];
        const double X = pt3d[3 * p], Y = pt3d[3 * p + 1], Z = pt3d[3 * p + 2];
        const double rx = (Rm[0] * X + Rm[1] * Y) + Rm[2] * Z;  // R X
        const double ry = (Rm[3] * X + Rm[4] * Y) + Rm[5] * Z;
        const double rz = (Rm[6] * X + Rm[7] * Y) + Rm[8] * Z;
        const double xc = rx + tv[0], yc = ry + tv[1], zc = rz + tv[2];
        const double iz = 1.0 / zc;
        // d(u or v)/d(Xc)
        double g0, g1, g2, res;
        if ((row & 1) == 0) {
          g0 = f * iz;
          g1 = 0.0;
          g2 = -f * xc * iz * iz;
          res = (f * xc * iz + ppx) - pt2d[2 * p];
        } else {
          g0 = 0.0;
          g1 = f * iz;
          g2 = -f * yc * iz * iz;
          res = (f * yc * iz + ppy) - pt2d[2 * p + 1];
        }
        // d(Exp(w) R X)/dw = -[R X]_x  ->  g^T (-[RX]_x) = ( RX x g )
        switch (col) {
          case 0: m = ry * g2 - rz * g1; break;
          case 1: m = rz * g0 - rx * g2; break;
          case 2: m = rx * g1 - ry * g0; break;
          case 3: m = g0; break;
          case 4: m = g1; break;
          case 5: m = g2; break;
          default: m = res; break;
        }
      }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);
    }
    // D[row = (lane>>4) + 4*reg][col = lane&15]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int drow = k + 4 * r;
      if (drow < 7 && col < 7) S.H[drow][col] = acc[r];
    }
  }
  __syncthreads();
}

// Exp(w) * R by Rodrigues' formula
__device__ void rotate_left(const double *w, const double *Rm, double *out) {
  const double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
  const double th = sqrt(th2);
  double a, b;  // Exp(w) = I + a [w]x + b [w]x^2
  if (th < 1e-8) {
    a = 1.0 - th2 / 6.0;
    b = 0.5 - th2 / 24.0;
  } else {
    a = sin(th) / th;
    b = (1.0 - cos(th)) / th2;
  }
  const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  double W2[9], E[9];
  matmul3(W, W, W2);
  for (int i = 0; i < 9; ++i) E[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * W[i] + b * W2[i];
  matmul3(E, Rm, out);
}

// returns the final cost; Rm/tv are updated in place (identically in every thread)
__device__ double refine_pose_block(const double *pt2d, const double *pt3d, const int32_t *inl, int n, double f,
                                    double ppx, double ppy, double *Rm, double *tv, int max_iter, int *iters_out) {
  __shared__ RefineShared S;
  normal_equations_wave0(pt2d, pt3d, inl, n, f, ppx, ppy, Rm, tv, S);
  double H[6][6], g[6], cost = S.H[6][6];
  for (int i = 0; i < 6; ++i) {
    g[i] = S.H[i][6];
    for (int j = 0; j < 6; ++j) H[i][j] = S.H[i][j];
  }
  double lambda = 1e-4;
  int it = 0;
  for (; it < max_iter; ++it) {
    // (H + lambda diag(H)) d = -g   -- every thread solves the same 6x6 system (Gaussian elimination, partial pivoting)
    double A[6][7];
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 6; ++j) A[i][j] = H[i][j];
      A[i][i] = H[i][i] * (1.0 + lambda);
      A[i][6] = -g[i];
    }
    bool singular = false;
    for (int c = 0; c < 6; ++c) {
      int piv = c;
      for (int r = c + 1; r < 6; ++r)
        if (dabs(A[r][c]) > dabs(A[piv][c])) piv = r;
      if (!(dabs(A[piv][c]) > 0.0)) {
        singular = true;
        break;
      }
      if (piv != c)
        for (int j = c; j < 7; ++j) {
          const double t = A[c][j];
          A[c][j] = A[piv][j];
          A[piv][j] = t;
        }
      for (int r = c + 1; r < 6; ++r) {
        const double fct = A[r][c] / A[c][c];
        for (int j = c; j < 7; ++j) A[r][j] = A[r][j] - fct * A[c][j];
      }
    }
    if (singular) break;
    double d[6];
    for (int i = 5; i >= 0; --i) {
      double acc = A[i][6];
      for (int j = i + 1; j < 6; ++j) acc = acc - A[i][j] * d[j];
      d[i] = acc / A[i][i];
    }
    double Rn[9], tn[3];
    rotate_left(d, Rm, Rn);
    tn[0] = tv[0] + d[3];
    tn[1] = tv[1] + d[4];
    tn[2] = tv[2] + d[5];
    __syncthreads();  // S.H of the previous pass has been read by every thread
    normal_equations_wave0(pt2d, pt3d, inl, n, f, ppx, ppy, Rn, tn, S);
    const double c_new = S.H[6][6];
    if (c_new < cost) {
      const double rel = (cost - c_new) / cost;
      for (int i = 0; i < 9; ++i) Rm[i] = Rn[i];
      for (int i = 0; i < 3; ++i) tv[i] = tn[i];
      cost = c_new;
      for (int i = 0; i < 6; ++i) {
        g[i] = S.H[i][6];
        for (int j = 0; j < 6; ++j) H[i][j] = S.H[i][j];
      }
      lambda = lambda * 0.1;
      if (lambda < 1e-12) lambda = 1e-12;
      if (rel < 1e-10) {
        ++it;
        break;
      }
    } else {
      lambda = lambda * 10.0;
      if (lambda > 1e10) break;
    }
  }
  __syncthreads();
  *iters_out = it;
  return cost;
}

// The sequential acceptance rule of ACRANSAC replayed over one round's results; executed by ONE workgroup, the last of
// the round to deliver its hypothesis (k_p3p_round), so a round is a single launch.
struct P3pReplayShared {
  double nfa[kP3pBatchMax];
  double err[kP3pBatchMax];
  int k[kP3pBatchMax];
  unsigned char m[kP3pBatchMax];  // wide launches: which of a hypothesis's four slots holds its best model
  int first[kThreads / 64];
};
// wide4: the launch has four result slots per hypothesis (4 b + m) and all four were written (one model per workgroup);
// a wide launch whose query turned out small wrote slot 4 b only, a plain launch slot b
__device__ __forceinline__ void p3p_replay_impl(const P3pArgs &A, int batch, int wide4, int slot_mul, P3pReplayShared &RS);
__device__ __forceinline__ void p3p_replay(const P3pArgs &A, int batch, int single_mode, P3pReplayShared &RS) {
  // (four result slots per hypothesis only in single mode; any other launch, wide or not, wrote slot b)
  p3p_replay_impl(A, batch, single_mode, single_mode ? 4 : 1, RS);
}
__device__ __forceinline__ void p3p_replay_impl(const P3pArgs &A, int batch, int wide, int slot_mul, P3pReplayShared &RS) {
  P3pState &st = *A.state;
  const int tid = threadIdx.x;
#ifdef SFMLOC_STAMPS
  const int stamp_round = st.rounds;
#endif
  STAMP_SEL(stamp_round, 0);
  batch = p3p_round_batch(st.n, batch);  // what the round evaluated
  if (batch > st.batch_limit) batch = st.batch_limit;
  const size_t inl_stride = p3p_inl_stride(st.n, A.max_n);
  // every thread replays the same scalar state machine; only the copies are cooperative
  long iter0 = st.iter, n_iter = st.n_iter, n_reserve = st.n_reserve;
  const long switch_iter0 = st.switch_iter;
  const long n_iter_evaluated = n_iter;  // k_p3p_eval ran hypotheses iter0 <= it < min(iter0+batch, n_iter)
  double min_nfa = st.min_nfa, errmax = st.errmax;
  int n_in = st.n_in, n_index = st.n_index, identity = st.identity;
  const int identity0 = identity;  // still sampling uniformly when the round began
  int best_b = -1;  // hypothesis of this batch that currently holds the best model
  long processed = 0;
  bool index_changed = false;
  // the replay is a dependent chain over up to 512 hypotheses: read their results from LDS, not from L2
  double *const s_nfa = RS.nfa;
  double *const s_err = RS.err;
  int *const s_k = RS.k;
  for (int b = tid; b < batch && b < kP3pBatchMax; b += kThreads) {
    const bool live = iter0 + b < n_iter;
    int slot = slot_mul * b;
    if (wide && live) {  // the hypothesis's best model: strictly smaller NFA, the first model on ties
      slot = 4 * b;
      double v = A.hyp_nfa[slot];
      for (int mm = 1; mm < 4; ++mm) {
        const double o = A.hyp_nfa[4 * b + mm];
        if (o < v) {
          v = o;
          slot = 4 * b + mm;
        }
      }
    }
    RS.m[b] = (unsigned char)(slot & 3);
    s_nfa[b] = live ? A.hyp_nfa[slot] : pos_inf();
    // (a slot that reports +inf wrote nothing else)
    s_err[b] = live && s_nfa[b] < pos_inf() ? A.hyp_err[slot] : 0.0;
    s_k[b] = live && s_nfa[b] < pos_inf() ? A.hyp_k[slot] : 0;
  }
  __syncthreads();
  // The sequential rule only acts at two kinds of hypotheses: one whose NFA improves on the running minimum, and the
  // one that ends the uniform phase.  Everything in between is skipped by a block-wide "first index below the
  // minimum" search instead of a dependent 512-step loop.
  int *const s_first = RS.first;
  auto first_below = [&](int from, int limit, double thr) -> int {
    int mine = 0x7FFFFFFF;
    for (int b = from + tid; b < limit; b += kThreads)
      if (s_nfa[b] < thr) {
        mine = b;
        break;
      }
    for (int off = 32; off > 0; off >>= 1) mine = min(mine, __shfl_xor(mine, off, 64));
    __syncthreads();  // s_first of the previous search is no longer read
    if ((tid & 63) == 0) s_first[tid >> 6] = mine;
    __syncthreads();
    int r = s_first[0];
    for (int w = 1; w < kThreads / 64; ++w) r = min(r, s_first[w]);
    return r;
  };
  int b = 0;
  for (;;) {
    // hypotheses b .. limit-1 were evaluated and are within the budget
    long lim = n_iter - iter0;
    if (lim > n_iter_evaluated - iter0) lim = n_iter_evaluated - iter0;
    if (lim > batch) lim = batch;
    const int limit = (int)lim;
    if (b >= limit) break;
    int e = first_below(b, limit, min_nfa);
    if (n_reserve) {  // "it + 1 == n_iter" with budget in reserve
      const long eb = n_iter - 1 - iter0;
      if (eb >= b && eb < limit && eb < e) e = (int)eb;
    }
    if (e >= limit) {
      processed = limit;
      break;
    }
    const long it = iter0 + e;
    const double nfa = s_nfa[e];
    bool better = false;
    if (nfa < min_nfa) {
      better = true;
      min_nfa = nfa;
      n_in = s_k[e];
      errmax = s_err[e];
      best_b = e;
    }
    processed = e + 1;
    b = e + 1;
    if ((better && min_nfa < 0.0) || (it + 1 == n_iter && n_reserve)) {
      if (n_in == 0) {
        n_iter++;
        n_reserve--;
      } else {
        index_changed = true;
        n_index = n_in;
        identity = 0;
        if (n_reserve) {
          n_iter = it + 1 + n_reserve;
          n_reserve = 0;
        }
        break;  // later hypotheses of this batch sampled from the old index set
      }
    }
  }
  STAMP_SEL(stamp_round, 1);
  const int best_slot = best_b < 0 ? 0 : (wide ? 4 * best_b + (int)RS.m[best_b] : slot_mul * best_b);
  if (best_b >= 0) {
    const int32_t *src = A.hyp_inl + (size_t)best_slot * inl_stride;
    for (int p = tid; p < n_in; p += kThreads) A.best_inl[p] = src[p];
    if (index_changed)
      for (int p = tid; p < n_in; p += kThreads) A.vec_index[p] = src[p];
  } else if (index_changed) {
    // the index set switches to the inliers of a model found in an EARLIER round (end of the main phase)
    for (int p = tid; p < n_in; p += kThreads) A.vec_index[p] = A.best_inl[p];
  }
  // (the index set is read again below -- "Prepared ahead" -- by other waves of this workgroup: the stores have left)
  if (A.prep_ahead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const bool done = (iter0 + processed >= n_iter);
  const long iter_next = iter0 + processed;
  const long switch_iter = (index_changed && identity0) ? iter_next : switch_iter0;
  const int next_limit = p3p_next_batch_limit(A, identity, iter_next, switch_iter, st.n);
  // how many of the coming round's hypotheses are solved here (none: every workgroup solves its own)
  int n_prep = 0;
  if (A.prep_ahead && !done) {
    n_prep = next_limit < kP3pBatchMax ? next_limit : kP3pBatchMax;
    n_prep = p3p_round_batch(st.n, n_prep);
    if ((long)n_prep > n_iter - iter_next) n_prep = (int)(n_iter - iter_next);
  }
  if (tid == 0) {
    st.prep_iter = (int)iter_next;
    st.prep_n = n_prep;
    st.iter = (int)(iter0 + processed);
    st.n_iter = (int)n_iter;
    st.n_reserve = (int)n_reserve;
    st.min_nfa = min_nfa;
    st.errmax = errmax;
    st.n_in = n_in;
    st.n_index = n_index;
    st.identity = identity;
    st.rounds += 1;
    st.switch_iter = (int)switch_iter;
    st.batch_limit = next_limit;
    st.first_hit = ~0u;
    if (best_b >= 0)
      for (int q = 0; q < 12; ++q) st.model[q] = A.hyp_model[12 * best_slot + q];
    if (done) st.done = 1;
    st.arrive = 0u;  // the next round counts from zero
  }
  STAMP_SEL(stamp_round, 2);
  // Prepared ahead.  In a round every workgroup spends its first 8-11 us with ONE lane solving its hypothesis's P3P --
  // a chain of ~2 500 dependent f64 instructions -- while the other 255 lanes wait, and a lone lane costs its SIMD as
  // many issue cycles as a full wave.  When the GPU is shared (prep_ahead) the workgroup that has just replayed the round
  // solves the COMING round's hypotheses instead, one per lane (the same scalar code on the same inputs: the same bits):
  // it knows the state they sample from, its 256 lanes take as long as one did, and the coming round's workgroups find
  // their four models in memory (p3p_eval_hypothesis: `have`).  Whatever the coming launch does not cover by these
  // (a larger batch, a state that moved on) is solved by its workgroup as before.  A query alone on the GPU keeps the
  // old form: the four roots' models side by side are 5 us shorter than one lane doing them in turn.
  if (n_prep > 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the index set as this workgroup has just written it, not an L1 line
    for (int h = tid; h < n_prep; h += kThreads) {
      int32_t smp[3];
      ac_sample<3>(identity ? nullptr : A.vec_index, n_index, A.seed, STAGE_P3P, A.stream, (uint32_t)(iter_next + h), smp);
      double x[6], X[9];
      for (int i = 0; i < 3; ++i) {
        x[2 * i] = A.xn[2 * smp[i]];
        x[2 * i + 1] = A.xn[2 * smp[i] + 1];
        X[3 * i] = A.pt3d[3 * smp[i]];
        X[3 * i + 1] = A.pt3d[3 * smp[i] + 1];
        X[3 * i + 2] = A.pt3d[3 * smp[i] + 2];
      }
      P3pPrep prep;
      const int nm = p3p_kneip_prepare(x, X, prep);
      double *M = A.prep_models + 48 * (size_t)h;
      for (int r = 0; r < nm; ++r) {
        double Mr[12];
        p3p_kneip_model(prep, r, Mr);
        for (int q = 0; q < 12; ++q) M[12 * r + q] = Mr[q];
      }
      A.prep_nm[h] = nm;
    }
  }
}

// One round of P3P AC-RANSAC in one launch: every workgroup evaluates one hypothesis and delivers it (release fence +
// counter); the workgroup that arrives last -- it sees every result (acquire fence) -- replays the sequential rule and
// writes the state of the next round.  Nobody waits for anybody, so there is no co-residency requirement; workgroups
// without a hypothesis (past the batch or the budget) only count themselves in.  A launch on a finished state returns at
// once (st.done is written by the previous launch's last workgroup, i.e. before this launch starts).
struct P3pRoundBody {
  static constexpr int kGangThreads = kThreads;
  static constexpr int kGangMinWaves = 4;  // (128 VGPRs, as k_p3p_round)
  static __device__ __forceinline__ void run(P3pArgs A, int batch, int wide) {
    constexpr int kForm = 0;
#include "p3p_round.body.inc"
  }
};
__global__ __launch_bounds__(kThreads, 4) void k_p3p_round(P3pArgs A, int batch, int wide) {
  constexpr int kForm = 0;
#include "p3p_round.body.inc"
}
// the small form (above, at P3pShared): at most kP3pSmallN correspondences, a fraction of the registers
struct P3pRoundSmallBody {
  static constexpr int kGangThreads = kThreads;
  static constexpr int kGangMinWaves = 4;
  static __device__ __forceinline__ void run(P3pArgs A, int batch, int wide) {
    constexpr int kForm = 1;
#include "p3p_round.body.inc"
  }
};
// (min. 4 waves per SIMD = at most 128 VGPRs: Kneip's solver spills -- one lane's chain, once per workgroup -- and the
// round fits beside four workgroups of the lean shortlist scan, 4 x 96 VGPRs per SIMD, without evicting one)
__global__ __launch_bounds__(kThreads, 4) void k_p3p_round_small(P3pArgs A, int batch, int wide) {
  constexpr int kForm = 1;
#include "p3p_round.body.inc"
}

// ---------------------------------------------------------------------------------------------------
// K5, the SEQUENTIAL form (k_p3p_seq): the whole AC-RANSAC of a query in ONE workgroup and ONE launch -- the form a
// query takes while the GPU is shared.
//
// The round form above buys latency with speculation: a round evaluates 64-256 hypotheses side by side and everything
// behind the round's first index-changing iteration is thrown away -- 1 040 hypotheses evaluated for the 410 iterations a
// headline query needs (tools/k5_policy_sim.py: no schedule gets below 850), in 9 launches of 64+ workgroups whose waves
// mostly wait on their own dependent f64 chains while holding a quarter of a SIMD's registers each.  When other queries
// fill the chip that speculation is pure cost.  Here NW waves (8 or 16) walk the iterations in order, `slots` models per
// step (a model is evaluated by W = next_pow2(n) / 256 waves: one wave's register sort up to 256 correspondences, W sorted
// runs of 256 merged through LDS beyond), the sequential rule of OpenMVG's ACRANSAC (restated in
// oracle/sfm_oracle_geom.c acransac()) is applied model by model in LDS, and a change of the index set costs a barrier, not a
// launch: at most slots - 1 models are evaluated for nothing per change.  Hypotheses are solved (Kneip) kSeqPool at a
// time, four lanes per hypothesis -- all four run the common part, each then its root's model.  Nobody waits for another
// workgroup; the loop advances by at least one iteration per step and n_iter is bounded, so every wave reaches the end.
// Same arithmetic per model as the round form (err_resection, the (key, index) order, det_log10, the first minimum),
// same rule: the result is the round form's bit for bit (tests/test_gpu_geom.py, fuzz_parity with SFMLOC_P3P_SEQ=2,
// bench.py's identical_to_single_flight: timed queries take this form, the single-flight reference the round form).
// ---------------------------------------------------------------------------------------------------
constexpr int kSeqPool = 64;          // hypotheses solved per refill (4 lanes each: the first four waves)
constexpr int kSeqRun = 256;          // correspondences one wave sorts (E = 4 registers per lane)
constexpr int kSeqPtsLds = 1024;      // up to this many correspondences the points live in LDS for the whole launch
struct P3pSeqLds {
  uint32_t pool, nm, res, key, low, vec, best, pts, ftab, hist, total;
};
__host__ __device__ inline P3pSeqLds p3p_seq_lds(int nw, int cap) {
  P3pSeqLds L;
  uint32_t o = 0;
  const uint32_t cpts = cap < kSeqPtsLds ? (uint32_t)cap : (uint32_t)kSeqPtsLds;
  L.pool = o; o += kSeqPool * 48 * sizeof(double);
  L.nm = o;   o += kSeqPool * sizeof(int);
  L.res = o;  o += 1024;                                          // per-slot results, reductions, the best model
  L.key = o;  o += cap > kSeqRun ? (uint32_t)nw * kSeqRun * sizeof(uint64_t) : 0u;  // sorted runs / merged keys (W > 1 only)
  L.low = o;  o += (uint32_t)nw * kSeqRun * sizeof(uint32_t);    // wave_sort_fast's parking space, then sorted indices
  L.vec = o;  o += (uint32_t)cap * sizeof(int32_t);              // the index set sampling draws from
  L.best = o; o += (uint32_t)cap * sizeof(int32_t);              // inliers of the best model so far
  L.pts = o;  o += cpts * 5 * sizeof(double);                    // X | Y | Z | x | y of the correspondences (n <= kSeqPtsLds)
  L.ftab = o; o += (kP3pFilterBins + 8) * sizeof(double);        // the NFA filter's thresholds T'
  // its counts, one table per model slot: in the merged keys' space when there is one (the filter runs before the merge)
  if (cap > kSeqRun) {
    L.hist = L.key;
  } else {
    L.hist = o; o += ((uint32_t)nw * (kSeqRun + 1) + 8) * sizeof(uint32_t);
  }
  L.total = o;
  return L;
}
struct P3pSeqRes {
  double nfa[16], err[16], red_nfa[16], best_model[12];
  int k[16], red_k[16];
};
static_assert(sizeof(P3pSeqRes) <= 1024, "P3pSeqLds::res");

// entries of a sorted run of kSeqRun keys that come before x: those < x, or (le) those <= x
__device__ __forceinline__ int seq_run_count(const uint64_t *run, uint64_t x, bool le) {
  int lo = 0, hi = kSeqRun;
#pragma unroll
  for (int it = 0; it < 9; ++it) {  // kSeqRun + 1 = 257 possible answers: nine halvings
    const int mid = (lo + hi) >> 1;
    const uint64_t v = run[mid < kSeqRun ? mid : kSeqRun - 1];
    const bool open = lo < hi;
    const bool before = le ? (v <= x) : (v < x);
    lo = (open && before) ? mid + 1 : lo;
    hi = (open && !before) ? mid : hi;
  }
  return lo;
}

// one hypothesis of the pool by four lanes: all four sample and run Kneip's common part (the same bits in each), lane
// `root` then its root's model.  Out of line: the solver's registers (~150) are not the loop's.
__device__ __noinline__ void seq_solve(const int32_t *vec_index, int n_index, uint64_t seed, uint32_t stream, uint32_t it,
                                       const double *__restrict__ xn, const double *__restrict__ pt3d, int root,
                                       double *models_h, int *nm_h) {
  int32_t smp[3];
  ac_sample<3>(vec_index, n_index, seed, STAGE_P3P, stream, it, smp);
  double x[6], X[9];
  for (int i = 0; i < 3; ++i) {
    x[2 * i] = xn[2 * smp[i]];
    x[2 * i + 1] = xn[2 * smp[i] + 1];
    X[3 * i] = pt3d[3 * smp[i]];
    X[3 * i + 1] = pt3d[3 * smp[i] + 1];
    X[3 * i + 2] = pt3d[3 * smp[i] + 2];
  }
  P3pPrep prep;
  const int nm = p3p_kneip_prepare(x, X, prep);
  if (root < nm) {
    double Mr[12];
    p3p_kneip_model(prep, root, Mr);
#pragma unroll
    for (int q = 0; q < 12; ++q) models_h[12 * root + q] = Mr[q];
  }
  if (root == 0) *nm_h = nm;
}

// The NFA filter's thresholds for the sequential form (the table of "The NFA filter" above, same formula and margin):
// T'[j] for the bins j = 0 .. nb - 1, recomputed whenever the best NFA B has changed.  All NT threads; barriers inside.
template <int NT>
__device__ __forceinline__ void seq_filter_table(double *T, int n, int s, double B, double logalpha0, double loge0,
                                                 const float *__restrict__ logc_n, const float *__restrict__ logc_k) {
  const int sh = p3p_filter_shift(n), w = 1 << sh;
  const int nb = (n >> sh) + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int jb = tid; jb < nb; jb += NT) {
    double m = 0.0;
    for (int k = jb << sh; k < (jb << sh) + w; ++k)
      if (k > s && k <= n) {
        const double x = (B - loge0 - (double)logc_n[k] - (double)logc_k[k]) / (double)(k - s) - logalpha0 + 1e-7;
        const double t = x > 300.0 ? pos_inf() : exp10(x);
        m = t > m ? t : m;
      }
    T[jb] = m;
  }
  __syncthreads();
  if (tid < 64) {  // running maximum, one wave: a contiguous run of bins per lane, then across the lanes
    const int per = (nb + 63) >> 6;
    double run = 0.0;
    for (int i = 0; i < per; ++i) {
      const int jb = lane * per + i;
      const double x = jb < nb ? T[jb] : 0.0;
      run = x > run ? x : run;
    }
    double inc = run;
    for (int off = 1; off < 64; off <<= 1) {
      const double o = __shfl_up(inc, off, 64);
      if (lane >= off) inc = o > inc ? o : inc;
    }
    double before = __shfl_up(inc, 1, 64);
    if (lane == 0) before = 0.0;
    run = before;
    for (int i = 0; i < per; ++i) {
      const int jb = lane * per + i;
      if (jb < nb) {
        const double x = T[jb];
        run = x > run ? x : run;
        T[jb] = run;
      }
    }
  }
  __syncthreads();
}

template <int NW>
__device__ __forceinline__ void p3p_seq_run(const P3pArgs &A, int cap) {
  P3pState &st = *A.state;
  if (st.done) return;
  const int n = st.n;
  const int P = next_pow2(n);
  const int W = P <= kSeqRun ? 1 : P / kSeqRun;  // waves per model
  // a set this launch is not built for: not its business (nothing is touched; the host queues the round form)
  if (n > cap || W > NW) return;
  extern __shared__ unsigned char smem_raw[];
  const P3pSeqLds L = p3p_seq_lds(NW, cap);
  double *const s_pool = reinterpret_cast<double *>(smem_raw + L.pool);
  int *const s_nm = reinterpret_cast<int *>(smem_raw + L.nm);
  P3pSeqRes &R = *reinterpret_cast<P3pSeqRes *>(smem_raw + L.res);
  uint64_t *const s_key = reinterpret_cast<uint64_t *>(smem_raw + L.key);
  uint32_t *const s_low = reinterpret_cast<uint32_t *>(smem_raw + L.low);
  int32_t *const s_vec = reinterpret_cast<int32_t *>(smem_raw + L.vec);
  int32_t *const s_best = reinterpret_cast<int32_t *>(smem_raw + L.best);
  double *const s_pts = reinterpret_cast<double *>(smem_raw + L.pts);
  double *const s_T = reinterpret_cast<double *>(smem_raw + L.ftab);
  uint32_t *const s_hist = reinterpret_cast<uint32_t *>(smem_raw + L.hist);
  constexpr int NT = NW * 64;
  constexpr int E = kSeqRun / 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int slots = NW / W;          // models evaluated per step
  const int grp = wv / W, j = wv - grp * W;  // this wave's model slot and its rank among the slot's waves
  const int base = j * kSeqRun;      // first element of this wave's run
  constexpr int s = 3;
  const double logalpha0 = det_log10(3.14159265358979323846);
  const double loge0 = det_log10(4.0 * (double)(n - s));
  const double *const pt3d = A.pt3d, *const xn = A.xn;
  const float *const logc_n = A.logc_n, *const logc_k = A.logc_k;
  const bool pts_lds = n <= kSeqPtsLds && n <= cap;
  const int np = pts_lds ? n : 0;
  // the state machine (every thread carries it; uniform)
  long iter = st.iter, n_iter = st.n_iter, n_reserve = st.n_reserve;
  double min_nfa = st.min_nfa, errmax = st.errmax;
  int n_in = st.n_in, n_index = st.n_index, identity = st.identity;
  int better = 0, steps = 0;
  // (a state some rounds have already advanced: its index set and best model)
  if (!identity)
    for (int p = tid; p < n_index; p += NT) s_vec[p] = A.vec_index[p];
  for (int p = tid; p < n_in; p += NT) s_best[p] = A.best_inl[p];
  if (tid < 12) R.best_model[tid] = st.model[tid];
  for (int p = tid; p < np; p += NT) {
    s_pts[p] = pt3d[3 * p];
    s_pts[np + p] = pt3d[3 * p + 1];
    s_pts[2 * np + p] = pt3d[3 * p + 2];
    s_pts[3 * np + p] = xn[2 * p];
    s_pts[4 * np + p] = xn[2 * p + 1];
  }
  // the two table entries of the sorted positions this lane reads in the NFA scan (k = position + 1)
  float cn[E], ck[E];
#pragma unroll
  for (int rr = 0; rr < E; ++rr) {
    const int kk = base + (rr << 6) + lane + 1;
    cn[rr] = kk <= n ? logc_n[kk] : 0.0f;
    ck[rr] = kk <= n ? logc_k[kk] : 0.0f;
  }
  // the NFA filter (section above): from the first finite best NFA on, a model is sorted only if it can beat it
  const int f_sh = p3p_filter_shift(n);
  const int f_nb = (n >> f_sh) + 1;
  int f_steps = 0;
  while ((1 << f_steps) < f_nb + 1) ++f_steps;
  uint32_t *const hist = s_hist + (size_t)grp * (W * kSeqRun + 1);
  double table_for = pos_inf();  // the best NFA the table in s_T was built for (+inf: none)
  long pool_it0 = 0;
  int pool_n = 0, g0 = 0;
#ifdef SFMLOC_SEQ_TIMING  // diagnostic build (make EXTRA=-DSFMLOC_SEQ_TIMING ...): where a launch's time goes, by thread 0's clock
  unsigned long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq0 = wall_clock64(), tq1;
  int n_refill = 0, n_pass = 0;
#define SEQ_T(k) do { tq1 = wall_clock64(); tq[k] += tq1 - tq0; tq0 = tq1; } while (0)
#else
#define SEQ_T(k) do {} while (0)
#endif
  __syncthreads();
  SEQ_T(0);
  while (iter < n_iter) {
    if (g0 >= 4 * pool_n) {
      // ---- solve the next hypotheses: iterations iter .. iter + pool_n - 1 from the current index set
      pool_it0 = iter;
      pool_n = (int)((n_iter - iter) < (long)kSeqPool ? (n_iter - iter) : (long)kSeqPool);
      g0 = 0;
      if (tid < 4 * pool_n)
        seq_solve(identity ? nullptr : s_vec, n_index, A.seed, A.stream, (uint32_t)(pool_it0 + (tid >> 2)), xn, pt3d, tid & 3,
                  s_pool + 48 * (tid >> 2), s_nm + (tid >> 2));
      __syncthreads();
      SEQ_T(1);
#ifdef SFMLOC_SEQ_TIMING
      ++n_refill;
#endif
    }
    const bool filter = A.nfa_filter && min_nfa < pos_inf();
    if (filter && table_for != min_nfa) {
      seq_filter_table<NT>(s_T, n, s, min_nfa, logalpha0, loge0, logc_n, logc_k);
      table_for = min_nfa;
    }
    // ---- evaluate models g0 .. g0 + slots - 1 (model m of pool hypothesis h is g = 4 h + m), one per slot
    const int g_mine = g0 + grp;
    const int h_mine = g_mine >> 2, m_mine = g_mine & 3;
    const bool active = h_mine < pool_n && pool_it0 + h_mine < n_iter && m_mine < s_nm[h_mine < pool_n ? h_mine : 0];
    uint64_t key[E];
    uint32_t idx[E];
    {
      double M[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) M[q] = s_pool[48 * (active ? h_mine : 0) + 12 * m_mine + q];
#pragma unroll
      for (int rr = 0; rr < E; ++rr) {
        const int p = base + (rr << 6) + lane;
        const int pc = p < n ? p : n - 1;
        double X, Y, Z, x, y;
        if (pts_lds) {
          X = s_pts[pc], Y = s_pts[np + pc], Z = s_pts[2 * np + pc], x = s_pts[3 * np + pc], y = s_pts[4 * np + pc];
        } else {
          X = pt3d[3 * pc], Y = pt3d[3 * pc + 1], Z = pt3d[3 * pc + 2], x = xn[2 * pc], y = xn[2 * pc + 1];
        }
        const double e = err_resection(M, X, Y, Z, x, y);
        key[rr] = (active && p < n) ? d2u(e) : ~0ull;
        idx[rr] = 0u;
      }
    }
    bool pass = active;
    SEQ_T(2);
    if (filter) {  // (uniform)
      for (int jb = j * 64 + lane; jb < f_nb; jb += W * 64) hist[jb] = 0u;
      __syncthreads();
      if (active) {
        int lo[E], hi[E];
        double rv[E];
#pragma unroll
        for (int rr = 0; rr < E; ++rr) {
          rv[rr] = u2d(key[rr]) + (double)FLT_EPSILON;
          lo[rr] = 0;
          hi[rr] = f_nb;
        }
        // first bin whose threshold exceeds the residual (f_nb: none); T' is non-decreasing
        for (int it = 0; it < f_steps; ++it) {
#pragma unroll
          for (int rr = 0; rr < E; ++rr) {
            const int mid = (lo[rr] + hi[rr]) >> 1;
            const bool below = mid < f_nb && rv[rr] < s_T[mid < f_nb ? mid : f_nb - 1];
            const bool open = lo[rr] < hi[rr];
            hi[rr] = (open && below) ? mid : hi[rr];
            lo[rr] = (open && !below) ? mid + 1 : lo[rr];
          }
        }
#pragma unroll
        for (int rr = 0; rr < E; ++rr)
          if (base + (rr << 6) + lane < n && lo[rr] < f_nb) atomicAdd(&hist[lo[rr]], 1u);
      }
      __syncthreads();
      // running counts over the slot's table (every wave of the slot walks it and reaches the same verdict): the model
      // can beat the best NFA only if some bin's running count reaches the bin's lowest k
      const int per = (f_nb + 63) >> 6;
      uint32_t sum = 0;
      for (int i = 0; i < per; ++i) {
        const int jb = lane * per + i;
        sum += jb < f_nb ? hist[jb] : 0u;
      }
      uint32_t inc = sum;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
      }
      uint32_t run = inc - sum;
      bool any = false;
      for (int i = 0; i < per; ++i) {
        const int jb = lane * per + i;
        if (jb < f_nb) {
          run += hist[jb];
          const int k_lo = (jb << f_sh) > s + 1 ? (jb << f_sh) : s + 1;  // lowest k of the bin that NFA ranges over
          const int k_hi = (jb << f_sh) + (1 << f_sh) - 1;
          if (k_hi > s && k_lo <= n && run >= (uint32_t)k_lo) any = true;
        }
      }
      pass = active && __ballot(any) != 0ull;
    }
    SEQ_T(3);
#ifdef SFMLOC_SEQ_TIMING
    n_pass += pass ? 1 : 0;
#endif
    if (W == 1) {
      NfaBest r{pos_inf(), 0x7FFFFFFF};
      double r_err = pos_inf();
      if (pass) {
        uint32_t *iw = s_low + (size_t)wv * kSeqRun;
        wave_sort_fast<E>(key, idx, iw);
#pragma unroll
        for (int rr = 0; rr < E; ++rr) iw[(rr << 6) + lane] = idx[rr];
        r = best_nfa_regs_ilp<E>(key, n, s, pos_inf(), logalpha0, 1.0, loge0, cn, ck);
        if (r.k != 0x7FFFFFFF) r_err = u2d(sorted_key_at<E>(key, r.k - 1));
      }
      if (lane == 0) {
        R.nfa[grp] = r.nfa;
        R.k[grp] = r.k;
        R.err[grp] = r_err;
      }
    } else if (__syncthreads_or(pass ? 1 : 0)) {
      // W waves, one model: wave j sorts elements 256 j .. 256 j + 255 in its registers, the W sorted runs are merged
      // by rank -- an element's final position is its position in its own run plus, per other run, the number of that
      // run's elements that come before it in the (key, index) order: runs of lower element indices win ties.  (Every
      // slot walks through the barriers; a slot whose model failed the filter reports +inf at the end.)
      wave_sort_fast<E>(key, idx, s_low + (size_t)wv * kSeqRun);
      uint64_t *const runs = s_key + (size_t)grp * W * kSeqRun;  // the slot's W runs, then its merged keys
      uint32_t *const fidx = s_low + (size_t)grp * W * kSeqRun;  // the slot's merged element indices
#pragma unroll
      for (int rr = 0; rr < E; ++rr) runs[base + (rr << 6) + lane] = key[rr];
      __syncthreads();
      int rank[E];
#pragma unroll
      for (int rr = 0; rr < E; ++rr) rank[rr] = (rr << 6) + lane;
      for (int u = 0; u < W; ++u) {
        if (u == j) continue;  // (uniform over the wave)
#pragma unroll
        for (int rr = 0; rr < E; ++rr) rank[rr] += seq_run_count(runs + u * kSeqRun, key[rr], u < j);
      }
      __syncthreads();  // every rank is known: the runs may be overwritten
#pragma unroll
      for (int rr = 0; rr < E; ++rr) {
        runs[rank[rr]] = key[rr];
        fidx[rank[rr]] = (uint32_t)base + idx[rr];
      }
      __syncthreads();
      // bestNFA over the merged order: position pos holds e_(pos + 1); the same candidates and the same first minimum
      // as best_nfa_regs_ilp
      double lb = pos_inf();
      int lk = 0x7FFFFFFF;
      if (pass) {
#pragma unroll
        for (int rr = 0; rr < E; ++rr) {
          const int kk = base + (rr << 6) + lane + 1;
          if (kk > s && kk <= n) {
            const double ek = u2d(runs[kk - 1]);
            const double logalpha = logalpha0 + 1.0 * det_log10_inline(ek + (double)FLT_EPSILON);
            const double nfa = loge0 + logalpha * (double)(kk - s) + (double)cn[rr] + (double)ck[rr];
            if (nfa < lb) {
              lb = nfa;
              lk = kk;
            }
          }
        }
      }
      const NfaBest wr = wave_reduce_nfa(lb, lk);
      if (lane == 0) {
        R.red_nfa[wv] = wr.nfa;
        R.red_k[wv] = wr.k;
      }
      __syncthreads();
      if (j == 0 && lane == 0) {
        NfaBest r{pos_inf(), 0x7FFFFFFF};
        for (int u = 0; u < W; ++u) {
          const double o = R.red_nfa[wv + u];
          const int ok = R.red_k[wv + u];
          if (o < r.nfa || (o == r.nfa && ok < r.k)) {
            r.nfa = o;
            r.k = ok;
          }
        }
        R.nfa[grp] = r.nfa;
        R.k[grp] = r.k;
        R.err[grp] = r.k != 0x7FFFFFFF ? u2d(runs[r.k - 1]) : pos_inf();
      }
    } else if (j == 0 && lane == 0) {
      R.nfa[grp] = pos_inf();  // no slot's model can beat the best NFA
    }
    __syncthreads();
    SEQ_T(4);
    // ---- the sequential rule over this step's models, in order (acransac(): per model "is it better", per hypothesis
    // -- behind its last model -- "does the index set change / does the budget end")
    int best_slot = -1;
    bool index_changed = false;
    for (int sl = 0; sl < slots; ++sl) {
      const int g = g0 + sl;
      const int h = g >> 2, m = g & 3;
      const long it = pool_it0 + h;
      if (h >= pool_n || it >= n_iter) break;
      if (m < s_nm[h]) {
        const double nfa = R.nfa[sl];
        if (nfa < min_nfa) {
          better = 1;
          min_nfa = nfa;
          n_in = R.k[sl];
          errmax = R.err[sl];
          best_slot = sl;
        }
      }
      if (m == 3) {
        if ((better && min_nfa < 0.0) || (it + 1 == n_iter && n_reserve)) {
          if (n_in == 0) {
            n_iter++;
            n_reserve--;
          } else {
            index_changed = true;
            n_index = n_in;
            identity = 0;
            if (n_reserve) {
              n_iter = it + 1 + n_reserve;
              n_reserve = 0;
            }
          }
        }
        better = 0;
        iter = it + 1;
        if (index_changed) break;
      }
    }
    if (best_slot >= 0) {
      const uint32_t *src = s_low + (size_t)best_slot * W * kSeqRun;
      for (int p = tid; p < n_in; p += NT) s_best[p] = (int32_t)src[p];
      const int g = g0 + best_slot;
      if (tid < 12) R.best_model[tid] = s_pool[48 * (g >> 2) + 12 * (g & 3) + tid];
    }
    __syncthreads();
    if (index_changed) {
      for (int p = tid; p < n_in; p += NT) s_vec[p] = s_best[p];
      pool_n = 0;  // what was solved ahead sampled from the old index set
      g0 = 0;
      __syncthreads();
    } else {
      g0 += slots;
    }
    ++steps;
    SEQ_T(5);
  }
#ifdef SFMLOC_SEQ_TIMING
  if (tid == 0)
    printf("seq NW=%d n=%d W=%d: %d steps, %d refills, wave0 passed %d | us: setup %.1f refill %.1f resid %.1f filter %.1f sort+nfa %.1f replay %.1f\n",
           NW, n, W, steps, n_refill, n_pass, tq[0] * 0.01, tq[1] * 0.01, tq[2] * 0.01, tq[3] * 0.01, tq[4] * 0.01, tq[5] * 0.01);
#endif
  // ---- the state k_p3p_finish reads
  for (int p = tid; p < n_in; p += NT) A.best_inl[p] = s_best[p];
  if (tid == 0) {
    st.iter = (int)iter;
    st.n_iter = (int)n_iter;
    st.n_reserve = (int)n_reserve;
    st.min_nfa = min_nfa;
    st.errmax = errmax;
    st.n_in = n_in;
    st.n_index = n_index;
    st.identity = identity;
    st.rounds += steps;
    st.prep_n = 0;
    for (int q = 0; q < 12; ++q) st.model[q] = R.best_model[q];
    st.done = 1;
  }
}
template <int NW>
struct P3pSeqBody {
  static constexpr int kGangThreads = NW * 64;
  static __device__ __forceinline__ void run(P3pArgs A, int cap) { p3p_seq_run<NW>(A, cap); }
};
// (min. 4 waves per SIMD = at most 128 VGPRs, whatever NW: the workgroup takes NW / 4 wave slots and NW / 4 x 128 registers of
// each SIMD and leaves the rest of the compute unit to the scans)
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void k_p3p_seq(P3pArgs A, int cap) {
  p3p_seq_run<NW>(A, cap);
}

// ACRANSAC's epilogue + SfM_Localizer::Localize + localization.cpp:511-547, once per query after the last round
// what the host reads after the stream has drained: the record's head (state -- "done?" --, pose, status, view counts)
// and, of a finished query, the inlier pairs.  Written by the kernel itself into the pinned host record: a copy launch
// per query (and per refill of rounds) less.
__device__ void p3p_publish(const P3pArgs &A) {
  if (!A.host) return;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(A.record);
  uint32_t *dst = reinterpret_cast<uint32_t *>(A.host);
  constexpr int head = (int)(offsetof(HostResult, pair_qfeat) / sizeof(uint32_t));
  for (int i = threadIdx.x; i < head; i += kThreads) dst[i] = src[i];
  const P3pState &st = *A.state;
  if (st.done && A.max_n <= kP3pMaxN) {  // (a regrown workspace keeps its pair lists outside the record)
    const int k = A.result->ok ? A.result->n_inliers : 0;
    for (int i = threadIdx.x; i < k; i += kThreads) {
      A.host->pair_qfeat[i] = A.record->pair_qfeat[i];
      A.host->pair_landmark[i] = A.record->pair_landmark[i];
    }
  }
}

struct P3pFinishBody {
  static constexpr int kGangThreads = kThreads;
  static __device__ __forceinline__ void run(P3pArgs A) {
    P3pState &st = *A.state;
    if (!st.done || st.finished) {  // rounds still to come, or nothing to estimate (k_p3p_init's verdict): just report
      p3p_publish(A);
      return;
    }
    const int tid = threadIdx.x;
    const double min_nfa = st.min_nfa, errmax = st.errmax;
    const int n_in = st.n_in;
#ifdef SFMLOC_STAMPS
    const int stamp_round = st.rounds;
#endif

    __shared__ double Msh[12];
    if (tid < 12) Msh[tid] = st.model[tid];
    __syncthreads();
    int n_final = n_in;
    if (min_nfa >= 0.0) n_final = 0;
    const bool resection = (double)n_final > 2.5 * 3;
    const bool ok = resection && n_final > A.min_inliers;
    Pose &R = *A.result;
    if (ok)
      for (int p = tid; p < n_final; p += kThreads) {
        const int32_t c = A.best_inl[p];
        A.pair_qfeat[p] = A.ms_qfeat[c];
        A.pair_landmark[p] = A.ms_landmark[c];
        A.inlier_idx[p] = c;
      }
    // every thread computes the (tiny) pose epilogue redundantly so that the refinement can use block barriers
    const double inv_f = 1.0 / A.focal;
    double Pm[12];
    for (int j = 0; j < 4; ++j) {  // P = K * [R|t]
      Pm[j] = A.focal * Msh[j] + A.ppx * Msh[8 + j];
      Pm[4 + j] = A.focal * Msh[4 + j] + A.ppy * Msh[8 + j];
      Pm[8 + j] = Msh[8 + j];
    }
    if (n_final == 0)
      for (int j = 0; j < 12; ++j) Pm[j] = 0.0;
    double Kq[9], Rq[9], tq[3], cq[3];
    double refine_cost = 0.0;
    int refine_iters = 0;
    if (ok) {
      krt_from_p(Pm, Kq, Rq, tq);
      if (A.refine_pose) {
        __syncthreads();  // inlier_idx written above by all threads
        refine_cost = refine_pose_block(A.pt2d, A.pt3d, A.best_inl, n_final, A.focal, A.ppx, A.ppy, Rq, tq, 20,
                                        &refine_iters);
        for (int r = 0; r < 3; ++r) {  // P = K [R|t] of the refined pose
          for (int j = 0; j < 3; ++j) {
            Pm[4 * r + j] = (r == 0) ? A.focal * Rq[j] + A.ppx * Rq[6 + j]
                          : (r == 1) ? A.focal * Rq[3 + j] + A.ppy * Rq[6 + j]
                                     : Rq[6 + j];
          }
          Pm[4 * r + 3] = (r == 0) ? A.focal * tq[0] + A.ppx * tq[2] : (r == 1) ? A.focal * tq[1] + A.ppy * tq[2] : tq[2];
        }
        Kq[0] = A.focal; Kq[1] = 0.0; Kq[2] = A.ppx; Kq[3] = 0.0; Kq[4] = A.focal; Kq[5] = A.ppy;
        Kq[6] = 0.0; Kq[7] = 0.0; Kq[8] = 1.0;
      }
      center_from_rt(Rq, tq, cq);
    }
    if (tid == 0) {
      R.ok = ok ? 1 : 0;
      R.n_inliers = n_final;
      R.n_matches_2d3d = st.n;
      R.iterations = st.iter;
      R.nfa = min_nfa;
      R.status = st.status;
      R.reserved = refine_iters;
      R.error_max = (n_final > 0) ? sqrt(errmax) / inv_f : errmax;
      for (int j = 0; j < 12; ++j) R.P[j] = Pm[j];
      if (ok) {
        for (int j = 0; j < 9; ++j) {
          R.K[j] = Kq[j];
          R.R[j] = Rq[j];
        }
        for (int j = 0; j < 3; ++j) {
          R.t[j] = tq[j];
          R.center[j] = cq[j];
        }
        if (A.refine_pose) R.stage_seconds[0] = refine_cost;  // overwritten by the host; kept for sfmloc_pose_read
      }
      st.finished = 1;
    }
    __syncthreads();  // the record is complete
    p3p_publish(A);
    STAMP_SEL(stamp_round, 3);
  }
};
__global__ __launch_bounds__(kThreads) void k_p3p_finish(P3pArgs A) {
  P3pFinishBody::run(A);
}

// L10[i] = log10(i) for the logcombi tables, computed once per map on the device
__global__ void k_fill_log10(double *L10, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) L10[i] = (i == 0) ? 0.0 : det_log10((double)i);
}

// parity probes for the f64 building blocks (sfmloc_debug_math)
__global__ void k_debug_math(int op, const double *in, int n, int in_stride, double *out, int out_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *x = in + (size_t)i * in_stride;
  double *o = out + (size_t)i * out_stride;
  switch (op) {
    case 0: o[0] = det_log10(x[0]); break;
    case 1: o[0] = sqrt(x[0]); o[1] = x[0] / x[1]; break;
    case 2: {
      double r[3] = {0, 0, 0};
      o[0] = (double)solve_cubic(x[0], x[1], x[2], x[3], r);
      o[1] = r[0]; o[2] = r[1]; o[3] = r[2];
    } break;
    case 3: solve_quartic_real(x, o); break;
    case 4: {
      double F[27];
      for (int k = 0; k < 27; ++k) F[k] = 0.0;
      o[0] = (double)seven_point(x, x + 14, F);
      for (int k = 0; k < 27; ++k) o[1 + k] = F[k];
    } break;
    case 5: {
      double M[48];
      for (int k = 0; k < 48; ++k) M[k] = 0.0;
      o[0] = (double)p3p_kneip(x, x + 6, M);
      for (int k = 0; k < 48; ++k) o[1 + k] = M[k];
    } break;
    case 6: {
      double c[3];
      krt_from_p(x, o, o + 9, o + 18);
      center_from_rt(o + 9, o + 18, c);
      o[21] = c[0]; o[22] = c[1]; o[23] = c[2];
    } break;
    case 7: {
      int32_t smp[7];
      const uint64_t seed = (uint64_t)x[0] | ((uint64_t)x[1] << 32);
      ac_sample<7>(nullptr, (int)x[2], seed, (uint32_t)x[3], (uint32_t)x[4], (uint32_t)x[5], smp);
      for (int k = 0; k < 7; ++k) o[k] = (double)smp[k];
    } break;
    case 10: o[0] = (double)round6_dev((float)x[0]); break;  // the .feat round trip of a coordinate
    default: break;
  }
}

// op 8: wave_seven_point, one wave per row (same input/output layout as op 4)
__global__ __launch_bounds__(64) void k_debug_wave7(const double *in, int n, int in_stride, double *out, int out_stride) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const double *x = in + (size_t)i * in_stride;
  double *o = out + (size_t)i * out_stride;
  const int lane = threadIdx.x & 63;
  int r = lane / 9;
  if (r > 6) r = 6;
  double f = 0.0;
  const int nm = wave_seven_point(x[2 * r], x[2 * r + 1], x[14 + 2 * r], x[14 + 2 * r + 1], &f);
  if (lane == 0) o[0] = (double)nm;
  if (lane < 27) o[1 + lane] = (lane < 9 * nm) ? f : 0.0;
}

// op 9: wave_sort_fast, one wave per row of P + 1 doubles (P = 64, 128, ... 1024 keys, then n: elements >= n are
// padding); out[p] = index of the element at sorted position p, out[P + p] = its key as a double
template <int E>
__device__ void debug_sort_row(const double *x, double *o, int n, uint32_t *low) {
  const int lane = threadIdx.x & 63;
  uint64_t key[E];
  uint32_t idx[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int p = (r << 6) + lane;
    key[r] = p < n ? d2u(x[p]) : ~0ull;
    idx[r] = (uint32_t)p;
  }
  wave_sort_fast<E>(key, idx, low);
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int p = (r << 6) + lane;
    o[p] = (double)idx[r];
    o[64 * E + p] = u2d(key[r]);
  }
}
__global__ __launch_bounds__(64) void k_debug_sort(const double *in, int n_rows, int in_stride, double *out, int out_stride) {
  __shared__ uint32_t low[1024];
  const int i = blockIdx.x;
  if (i >= n_rows) return;
  const double *x = in + (size_t)i * in_stride;
  double *o = out + (size_t)i * out_stride;
  const int P = in_stride - 1;
  const int n = (int)x[P];
  switch (P >> 6) {
    case 1: debug_sort_row<1>(x, o, n, low); break;
    case 2: debug_sort_row<2>(x, o, n, low); break;
    case 4: debug_sort_row<4>(x, o, n, low); break;
    case 8: debug_sort_row<8>(x, o, n, low); break;
    case 16: debug_sort_row<16>(x, o, n, low); break;
    default: break;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
int launch_fill_log10(double *d_L10, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_fill_log10, dim3((n + 255) / 256), dim3(256), 0, s, d_L10, n);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

int launch_debug_math(int op, const double *d_in, int n, int in_stride, double *d_out, int out_stride,
                      hipStream_t s) {
  if (op == 8)
    hipLaunchKernelGGL(k_debug_wave7, dim3(n), dim3(64), 0, s, d_in, n, in_stride, d_out, out_stride);
  else if (op == 9)
    hipLaunchKernelGGL(k_debug_sort, dim3(n), dim3(64), 0, s, d_in, n, in_stride, d_out, out_stride);
  else
    hipLaunchKernelGGL(k_debug_math, dim3((n + 63) / 64), dim3(64), 0, s, op, d_in, n, in_stride, d_out, out_stride);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// global-memory slots of k_fmatrix_large, allocated the first time a map whose longest view exceeds the LDS form is used
static int ensure_fmatrix_large(Ctx *c) {
  if (c->fl_key) return SFMLOC_OK;
  Map *m = c->map;
  int slot_m = 64;
  while (slot_m < (int)m->max_view_rows && slot_m < kFLargeMaxM) slot_m <<= 1;
  c->fl_slot_m = slot_m;
  const size_t n = (size_t)kFLargeSlots * slot_m, n1 = (size_t)kFLargeSlots * (slot_m + 1);
  SFM_HIP(hipMalloc((void **)&c->fl_key, n * sizeof(uint64_t)));
  SFM_HIP(hipMalloc((void **)&c->fl_idx, n * sizeof(uint32_t)));
  SFM_HIP(hipMalloc((void **)&c->fl_vec_index, n * sizeof(int32_t)));
  SFM_HIP(hipMalloc((void **)&c->fl_best_inl, n * sizeof(int32_t)));
  SFM_HIP(hipMalloc((void **)&c->fl_logc_n, n1 * sizeof(float)));
  SFM_HIP(hipMalloc((void **)&c->fl_logc_k, n1 * sizeof(float)));
  SFM_HIP(hipMalloc((void **)&c->fl_count, sizeof(uint32_t)));
  SFM_HIP(hipMemset(c->fl_count, 0, sizeof(uint32_t)));  // (from the next query on the query's reset kernel clears it)
  SFM_HIP(hipStreamSynchronize(nullptr));                // (null stream: not ordered with the context's stream otherwise)
  SFM_HIP(hipMalloc((void **)&c->fl_list, ((size_t)m->n_views + 1) * sizeof(uint32_t)));
  c->hbm_bytes += n * 20 + n1 * 8 + ((size_t)m->n_views + 2) * 4;
  return SFMLOC_OK;
}

// the argument list as the kernels take it (FFilterArgsPacked): the context's block on the device is rewritten when its
// content changes -- the first query, and the query after a lazily made buffer appeared -- and the host waits for that
// write (the block is read by launches queued behind it; a second change must not overtake them)
static_assert(sizeof(FFilterStatic) <= sizeof(Ctx::k3_static_host), "Ctx::k3_static_host holds an FFilterStatic");
static int k3_pack(Ctx *c, const FFilterArgs &A, FFilterArgsPacked *P) {
  FFilterStatic T;
  memset(&T, 0, sizeof(T));
  T.view_off = A.view_off;
  T.view_id = A.view_id;
  T.view_wh = A.view_wh;
  T.put_count = A.put_count;
  T.match_i = A.match_i;
  T.match_key = A.match_key;
  T.map_kpt = A.map_kpt;
  T.precision = A.precision;
  T.n_iter = A.n_iter;
  T.seed = A.seed;
  T.L10 = A.L10;
  T.geo_count = A.geo_count;
  T.geo_idx = A.geo_idx;
  T.geo_model = A.geo_model;
  T.large_count = A.large_count;
  T.large_list = A.large_list;
  T.status = A.status;
  // (a launch that does not merge leaves the block's merge pointers as they are: queries that alternate between the two
  // do not rewrite it)
  if (A.merge.enabled) T.merge = A.merge;
  else if (c->k3_static_valid) T.merge = reinterpret_cast<const FFilterStatic *>(c->k3_static_host)->merge;
  T.merge.enabled = 0;
  T.merge.view_sel = nullptr;
  T.merge.view_widx0 = nullptr;
  if (!c->d_k3_static) SFM_HIP(hipMalloc(&c->d_k3_static, sizeof(c->k3_static_host)));
  if (!c->k3_static_valid || memcmp(&T, c->k3_static_host, sizeof(T)) != 0) {
    memcpy(c->k3_static_host, &T, sizeof(T));
    // (ahead of whatever a recording member has recorded -- nothing recorded reads the block before this call's launches
    // -- and without issuing it: a plain use of the stream would end the session's sharing for this member)
    hipStream_t s = c->stream.unordered();
    SFM_HIP(hipMemcpyAsync(c->d_k3_static, c->k3_static_host, sizeof(T), hipMemcpyHostToDevice, s));
    SFM_HIP(hipStreamSynchronize(s));
    c->k3_static_valid = true;
  }
  P->st = reinterpret_cast<const FFilterStatic *>(c->d_k3_static);
  P->view_sel = A.view_sel;
  P->q_kpt6 = A.q_kpt6;
  P->merge_view_sel = A.merge.view_sel;
  P->merge_view_widx0 = A.merge.view_widx0;
  P->spec = A.spec;
  P->spec_arrive = A.spec_arrive;
  P->n_sel = A.n_sel;
  P->qw = A.qw;
  P->qh = A.qh;
  P->min_putative = A.min_putative;
  P->skip_le = A.skip_le;
  P->fast_min = A.fast_min;
  P->merge_enabled = A.merge.enabled;
  return SFMLOC_OK;
}

int launch_fmatrix_filter(Ctx *c, const Query *q, uint32_t n_sel, bool all_views, int min_putative) {
  Map *m = c->map;
  if (n_sel == 0) return SFMLOC_OK;
  FFilterArgs A;
  A.view_sel = all_views ? nullptr : c->d_view_sel;
  A.n_sel = n_sel;
  A.view_off = m->d_view_off;
  A.view_id = m->d_view_id;
  A.view_wh = m->d_view_wh;
  A.put_count = c->d_view_count;
  A.match_i = c->d_match_i;
  A.match_key = c->d_match_key;
  A.map_kpt = m->d_kpt;
  A.q_kpt6 = q->d_kpt6;
  A.qw = q->width;
  A.qh = q->height;
  A.precision = m->params.geom_precision;
  A.n_iter = m->params.ransac_round;
  A.seed = m->params.seed;
  A.min_putative = min_putative >= 0 ? min_putative : m->params.min_putative;
  A.L10 = m->d_L10;
  A.geo_count = c->d_geo_count;
  A.geo_idx = c->d_geo_idx;
  A.geo_model = m->params.guided_matching ? c->d_geo_model : nullptr;
  A.status = c->d_status;
  c->geo_is_pairs = false;
  A.large_count = nullptr;
  A.large_list = nullptr;
  if (m->max_view_rows > (uint32_t)kFMaxM) {
    const bool fresh = c->fl_key == nullptr;
    int rc = ensure_fmatrix_large(c);
    if (rc) return rc;
    // (a memset on the context's stream would end a gang session's recording: every real map has views above 2 048 rows,
    // and with this line a session of image-in frames issued 93 launches, none of them shared -- the query's reset
    // kernel clears the counter; only the staged API, which has no reset kernel, clears it here)
    if (!c->cleared && !fresh) SFM_HIP(hipMemsetAsync(c->fl_count, 0, sizeof(uint32_t), c->stream));
    A.large_count = c->fl_count;
    A.large_list = c->fl_list;
  }
  // views with <= kF2MaxM putative matches take the wave-parallel kernel, the rest (if any: the second launch
  // returns at once for the others) the block-wide one; SFMLOC_K3_FAST=0 sends every view to the latter
  static const bool fast = [] {
    const char *e = getenv("SFMLOC_K3_FAST");
    return !(e && atoi(e) == 0);
  }();
  A.skip_le = fast ? kF2MaxM : -1;
  A.fast_min = -1;
  A.merge = MergeMaskedArgs{};
  // The wide form (one workgroup per iteration of a view's first batch, fmatrix_fast.body.inc) for a query alone on the GPU
  // with a short view list: its 1 024-match instance takes every view the register forms hold, in one launch.
  // SFMLOC_K3_WIDE = 0 never, 2 always (tests, campaigns).
  static const int env_wide = [] { const char *e = getenv("SFMLOC_K3_WIDE"); return e ? atoi(e) : 1; }();
  const int n_uniform = m->params.ransac_round - m->params.ransac_round / 10;
  const int wide_b0 = n_uniform < kF2Batch ? n_uniform : kF2Batch;
  const bool wide = fast && wide_b0 >= 2 && n_sel <= (uint32_t)kK3WideViews &&
                    (env_wide == 2 || (env_wide == 1 && c->k1_may_slice && c->stream.gang == nullptr));
  A.spec = nullptr;
  A.spec_arrive = nullptr;
  if (c->merge_is_deferred) {  // K2 was left to this stage (launch_merge_ratio_compact)
    c->merge_is_deferred = false;
    if (fast && !wide) {
      A.merge = c->deferred_merge;  // k_fmatrix_fast runs on every selected view, whatever its size
    } else {  // (the wide form's workgroups of a view all need the view's lists: K2 as a launch of its own)
      int rc = launch_merge_masked_now(c, n_sel);
      if (rc) return rc;
    }
  }
  if (wide) {
    if (!c->d_k3_spec) {
      SFM_HIP(hipMalloc((void **)&c->d_k3_spec, (size_t)kK3WideViews * kF2Batch * sizeof(K3Spec)));
      SFM_HIP(hipMalloc((void **)&c->d_k3_arrive, (size_t)kK3WideViews * sizeof(unsigned int)));
      SFM_HIP(hipMemset(c->d_k3_arrive, 0, (size_t)kK3WideViews * sizeof(unsigned int)));  // (the last arrival clears its slot)
      SFM_HIP(hipStreamSynchronize(nullptr));  // (null stream: not ordered with the context's stream otherwise)
      c->hbm_bytes += (size_t)kK3WideViews * (kF2Batch * sizeof(K3Spec) + sizeof(unsigned int));
    }
    A.spec = reinterpret_cast<K3Spec *>(c->d_k3_spec);
    A.spec_arrive = c->d_k3_arrive;
    // (views of 1 025 .. 2 048 matches -- 32 residuals per lane in the register sort, one workgroup per compute unit --
    // only while the map's queries have had such views lately, Map::k3_huge_credit; SFMLOC_K3_WIDE_2048 = 0 never, 2 always:
    // k_fmatrix_filter's block-wide sort took 0.6 of a lone 1080p frame's 0.95 ms in this stage)
    static const int env_2048 = [] { const char *e = getenv("SFMLOC_K3_WIDE_2048"); return e ? atoi(e) : 1; }();
    const bool huge = env_2048 == 2 || (env_2048 == 1 && m->k3_huge_credit.load(std::memory_order_relaxed) > 0);
    auto go_wide = [&](auto m_tag) {
      constexpr int MaxM = decltype(m_tag)::value;
      using Sh = F2SharedT<4, MaxM>;
      static const hipError_t attrw = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fmatrix_fast<4, MaxM, true>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Sh));
      if (attrw != hipSuccess) return attrw;
      FFilterArgsPacked P;
      if (k3_pack(c, A, &P) != SFMLOC_OK) return hipErrorOutOfMemory;
      sfm_launch<FmatrixFastBody<4, MaxM, true>>(c, k_fmatrix_fast<4, MaxM, true>, dim3(n_sel, (unsigned)wide_b0), dim3(256),
                                                 (uint32_t)sizeof(Sh), P);
      return hipSuccess;
    };
    const hipError_t ew = huge ? go_wide(std::integral_constant<int, 2048>{}) : go_wide(std::integral_constant<int, 1024>{});
    SFM_HIP(ew);
    SFM_HIP(hipGetLastError());
    A.skip_le = huge ? 2048 : 1024;
  } else if (fast) {
    // waves per view (F2SharedT): 16 for a query alone on the GPU, 4 when other contexts have work queued
    static const int env_waves = [] { const char *e = getenv("SFMLOC_K3_WAVES_SHARED"); return e ? atoi(e) : 4; }();
    static const int env_waves_alone = [] { const char *e = getenv("SFMLOC_K3_WAVES_ALONE"); return e ? atoi(e) : 16; }();
    const int waves = c->k1_may_slice ? env_waves_alone : env_waves;
    auto go = [&](auto w_tag) {
      constexpr int W = decltype(w_tag)::value;
      static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fmatrix_fast<W>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(F2SharedT<W>));
      if (attr2 != hipSuccess) return attr2;
      FFilterArgsPacked P;
      if (k3_pack(c, A, &P) != SFMLOC_OK) return hipErrorOutOfMemory;
      sfm_launch<FmatrixFastBody<W>>(c, k_fmatrix_fast<W>, dim3(n_sel), dim3(W * 64), (uint32_t)sizeof(F2SharedT<W>), P);
      return hipSuccess;
    };
    const hipError_t e2 = waves == 16 ? go(std::integral_constant<int, 16>{})
                          : waves == 4 ? go(std::integral_constant<int, 4>{}) : go(std::integral_constant<int, 8>{});
    SFM_HIP(e2);
    SFM_HIP(hipGetLastError());
    // views with 513 .. 1 024 putative matches (a query that nearly duplicates a map frame): the same kernel with 16
    // residuals per lane -- while the map's queries have had such views lately (Map::k3_big_credit, set from the largest
    // view of every finished query: a launch whose 100 workgroups all leave at once still costs a launch)
    static const int env_big = [] { const char *e = getenv("SFMLOC_K3_BIG"); return e ? atoi(e) : 1; }();
    if (env_big == 2 || (env_big == 1 && m->k3_big_credit.load(std::memory_order_relaxed) > 0)) {
      FFilterArgs B = A;
      B.merge.enabled = 0;  // (the lists exist: the launch above built them)
      B.fast_min = kF2MaxM;
      auto go_big = [&](auto w_tag) {
        constexpr int W = decltype(w_tag)::value;
        using Sh = F2SharedT<W, 1024>;
        static const hipError_t attr3 = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fmatrix_fast<W, 1024>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Sh));
        if (attr3 != hipSuccess) return attr3;
        FFilterArgsPacked P;
        if (k3_pack(c, B, &P) != SFMLOC_OK) return hipErrorOutOfMemory;
        sfm_launch<FmatrixFastBody<W, 1024>>(c, k_fmatrix_fast<W, 1024>, dim3(n_sel), dim3(W * 64), (uint32_t)sizeof(Sh), P);
        return hipSuccess;
      };
      const hipError_t e3 = c->k1_may_slice ? go_big(std::integral_constant<int, 8>{}) : go_big(std::integral_constant<int, 4>{});
      SFM_HIP(e3);
      SFM_HIP(hipGetLastError());
      A.skip_le = 1024;
    }
  }
  const size_t lds = sizeof(FShared);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fmatrix_filter),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FShared));
  SFM_HIP(attr1);
  A.merge.enabled = 0;  // (the lists exist by now)
  FFilterArgsPacked PF;
  {
    int rc = k3_pack(c, A, &PF);
    if (rc) return rc;
  }
  sfm_launch<FmatrixFilterBody>(c, k_fmatrix_filter, dim3(n_sel), dim3(kThreads), (uint32_t)lds, PF);
  SFM_HIP(hipGetLastError());
  if (A.large_list) {  // some view of this map can have more than kFMaxM matches: the queue's consumer
    FLargeArgs W;
    W.key = c->fl_key;
    W.idx = c->fl_idx;
    W.vec_index = c->fl_vec_index;
    W.best_inl = c->fl_best_inl;
    W.logc_n = c->fl_logc_n;
    W.logc_k = c->fl_logc_k;
    W.slot_m = c->fl_slot_m;
    sfm_launch<FmatrixLargeBody>(c, k_fmatrix_large, dim3(kFLargeSlots), dim3(kThreads), 0, PF, W);
    SFM_HIP(hipGetLastError());
  }
  return SFMLOC_OK;
}

int launch_emit_candidates(Ctx *c, const Query *q, uint32_t n_sel, bool all_views) {
  Map *m = c->map;
  if (!c->cleared) {
    SFM_HIP(hipMemsetAsync(c->d_cand_part, 0, kPartHeaderBytes, c->stream));
    SFM_HIP(hipMemsetAsync(c->d_view_stats, 0, 3 * sizeof(uint32_t), c->stream));
    SFM_HIP(hipMemsetAsync(c->d_best64, 0xFF, (size_t)(q->n ? q->n : 1) * sizeof(unsigned long long), c->stream));
  }
  if (n_sel == 0 || q->n == 0) return SFMLOC_OK;
  const uint32_t *sel = all_views ? nullptr : c->d_view_sel;
  const uint32_t *gj = c->geo_is_pairs ? c->d_geo_j : nullptr;
  sfm_launch<EmitMinBody>(c, k_emit_min, dim3(n_sel), dim3(256), 0, sel, n_sel, m->d_view_off, m->d_view_id,
                          c->d_view_count, c->d_match_i, c->d_match_key, c->d_geo_count, c->d_geo_idx, gj,
                          m->d_row_landmark, c->d_best64, c->d_geo_dist, (uint32_t)m->params.min_putative, c->d_view_stats);
  SFM_HIP(hipGetLastError());
  sfm_launch<EmitWinBody>(c, k_emit_win, dim3(n_sel), dim3(256), 0, sel, n_sel, m->d_view_off, m->d_view_id, c->d_match_i,
                          c->d_match_key, c->d_geo_count, c->d_geo_idx, gj, m->d_row_landmark, m->d_landmark_id,
                          m->d_landmark_X, c->d_best64, c->d_geo_dist,
                          reinterpret_cast<Candidate *>(c->d_cand_part + kPartHeaderBytes), c->cand_cap,
                          reinterpret_cast<uint32_t *>(c->d_cand_part), c->d_status);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// a context's candidate part -> a caller's buffer: the 16-byte header (true count) and the candidates that exist, at
// most cap of them (a fixed-size copy would move cap * 40 bytes for a few hundred candidates)
struct ExportPartBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst,
                                             uint32_t cap) {
    const uint32_t n = min(*reinterpret_cast<const uint32_t *>(src), cap);
    const uint64_t words = (kPartHeaderBytes + (uint64_t)n * sizeof(Candidate)) / 8;  // both multiples of 8
    const uint2 *s8 = reinterpret_cast<const uint2 *>(src);
    uint2 *d8 = reinterpret_cast<uint2 *>(dst);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
      d8[i] = s8[i];
  }
};
__global__ __launch_bounds__(256) void k_export_part(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst,
                                             uint32_t cap) {
  ExportPartBody::run(src, dst, cap);
}

int launch_export_part(Ctx *c, void *dst_dev, uint32_t cap) {
  sfm_launch<ExportPartBody>(c, k_export_part, dim3(8), dim3(256), 0, c->d_cand_part,
                     reinterpret_cast<unsigned char *>(dst_dev), cap);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// a context's candidates appended to a batch's packed part (layout: PartLayout above).  One workgroup: lane 0 claims
// [off, off + n) of the candidate area with one atomic on the header's running total, then everybody copies.
struct ExportPackedBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const unsigned char *__restrict__ src, uint32_t src_cap,
                                               unsigned char *__restrict__ dst, uint32_t n_queries,
                                               uint32_t budget, uint32_t qi) {
    __shared__ uint32_t s_off, s_n;
    uint32_t *h = reinterpret_cast<uint32_t *>(dst);
    if (threadIdx.x == 0) {
      const uint32_t n_true = *reinterpret_cast<const uint32_t *>(src);
      const uint32_t n = min(n_true, src_cap);
      if (n_true > src_cap) atomicOr(&h[3], 2u);      // the context itself overflowed: candidates are lost
      const uint32_t off = atomicAdd(&h[0], n);       // the total keeps counting past the budget: every rank sees by how much
      const bool fits = (uint64_t)off + n <= budget;
      if (!fits) atomicOr(&h[3], 1u);
      h[1] = n_queries;
      h[2] = budget;
      h[4 + qi] = fits ? n : 0u;
      h[4 + n_queries + qi] = fits ? off : 0u;
      s_off = off;
      s_n = fits ? n : 0u;
    }
    __syncthreads();
    const uint2 *s8 = reinterpret_cast<const uint2 *>(src + kPartHeaderBytes);
    uint2 *d8 = reinterpret_cast<uint2 *>(dst + packed_cands_offset(n_queries) + (uint64_t)s_off * sizeof(Candidate));
    const uint64_t words = (uint64_t)s_n * (sizeof(Candidate) / 8);
    for (uint64_t i = threadIdx.x; i < words; i += blockDim.x) d8[i] = s8[i];
  }
};
__global__ __launch_bounds__(256) void k_export_packed(const unsigned char *__restrict__ src, uint32_t src_cap,
                                               unsigned char *__restrict__ dst, uint32_t n_queries,
                                               uint32_t budget, uint32_t qi) {
  ExportPackedBody::run(src, src_cap, dst, n_queries, budget, qi);
}

uint64_t packed_part_bytes(uint32_t n_queries, uint32_t budget) {
  return packed_cands_offset(n_queries) + (uint64_t)budget * sizeof(Candidate);
}

int launch_export_packed(Ctx *c, void *dst_dev, uint32_t n_queries, uint32_t budget, uint32_t qi) {
  sfm_launch<ExportPackedBody>(c, k_export_packed, dim3(1), dim3(256), 0, c->d_cand_part, c->cand_cap,
                               reinterpret_cast<unsigned char *>(dst_dev), n_queries, budget, qi);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

static P3pArgs make_p3p_args(Ctx *c);

// what a query's 2D-3D selection starts from when no k_query_reset ran for it (sfmloc_merge_begin, the staged API)
struct SelectResetBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(unsigned long long *__restrict__ best64, uint32_t nq, uint32_t *__restrict__ ms_n,
                                             int *__restrict__ status) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < nq) best64[i] = ~0ull;
    if (i == 0) {
      *ms_n = 0;
      if (status) *status = 0;
    }
  }
};
__global__ __launch_bounds__(256) void k_select_reset(unsigned long long *__restrict__ best64, uint32_t nq,
                                                      uint32_t *__restrict__ ms_n, int *__restrict__ status) {
  SelectResetBody::run(best64, nq, ms_n, status);
}

int launch_select_candidates(Ctx *c, const Query *q, const unsigned char *parts, uint32_t n_parts,
                             uint64_t part_bytes, uint32_t cap, uint32_t packed_b, uint32_t packed_qi, bool reset_status) {
  PartLayout L;
  L.packed_b = packed_b;
  L.qi = packed_qi;
  if (!c->cleared) {  // (one launch, and one that a gang session can carry, instead of three memsets)
    const uint32_t nq1 = q->n ? q->n : 1;
    sfm_launch<SelectResetBody>(c, k_select_reset, dim3((nq1 + 255) / 256), dim3(256), 0, c->d_best64, nq1, c->d_ms_n,
                                reset_status ? c->d_status : (int *)nullptr);
    SFM_HIP(hipGetLastError());
  }
  if (q->n == 0 || n_parts == 0) return SFMLOC_OK;
  const dim3 grid(16, n_parts < 64 ? n_parts : 64);
  // the context's own part (single GPU: the emission just ran on this stream) already holds exactly the winners and
  // d_best64 their keys, so the minimum pass would change nothing
  const bool own_part = (parts == c->d_cand_part && n_parts == 1 && packed_b == 0 && c->cleared);
  if (!own_part) {
    sfm_launch<CandidatesMinBody>(c, k_candidates_min, grid, dim3(256), 0, parts, n_parts, part_bytes, cap, q->n,
                                  c->d_best64, c->d_status, L);
    SFM_HIP(hipGetLastError());
  }
  sfm_launch<MatchSetFinishBody>(c, k_match_set_finish, dim3(1), dim3(1024), 0, parts, n_parts, part_bytes, cap,
                                 c->d_best64, c->d_winner, q->n, q->d_kpt, c->d_ms_n, c->d_ms_qfeat, c->d_ms_landmark,
                                 c->d_pt2d, c->d_pt3d, c->map->intrinsic_type == 3 ? 1 : 0, c->map->focal, c->map->ppx,
                                 c->map->ppy, c->map->k1, c->map->k2, c->map->k3, L, make_p3p_args(c));
  SFM_HIP(hipGetLastError());
  c->p3p_init_fused = true;  // launch_p3p_init is then a no-op for this query
  return SFMLOC_OK;
}

int launch_match_set(Ctx *c, const Query *q, uint32_t n_sel, bool all_views) {
  int rc = launch_emit_candidates(c, q, n_sel, all_views);
  if (rc) return rc;
  return launch_select_candidates(c, q, c->d_cand_part, 1, kPartHeaderBytes + (uint64_t)c->cand_cap * sizeof(Candidate),
                                  c->cand_cap);
}

static P3pArgs make_p3p_args(Ctx *c) {
  Map *m = c->map;
  P3pArgs A;
  A.state = c->d_p3p_state;
  A.result = c->d_pose;
  A.record = reinterpret_cast<const HostResult *>(c->d_result);
  A.host = reinterpret_cast<HostResult *>(c->h_result);
  A.ms_n = c->d_ms_n;
  A.ms_qfeat = c->d_ms_qfeat;
  A.ms_landmark = c->d_ms_landmark;
  A.pt2d = c->d_pt2d;
  A.pt3d = c->d_pt3d;
  A.xn = c->d_xn;
  A.L10 = m->d_L10;
  A.logc_n = c->d_logc_n;
  A.logc_k = c->d_logc_k;
  A.vec_index = c->d_vec_index;
  A.best_inl = c->d_best_inl;
  A.hyp_nfa = c->d_hyp_nfa;
  A.hyp_k = c->d_hyp_k;
  A.hyp_err = c->d_hyp_err;
  A.hyp_model = c->d_hyp_model;
  A.prep_models = c->d_prep_models;
  A.prep_nm = c->d_prep_nm;
  A.hyp_inl = c->d_hyp_inl;
  A.pair_qfeat = c->d_pair_qfeat;
  A.pair_landmark = c->d_pair_landmark;
  A.inlier_idx = c->d_inlier_idx;
  A.focal = m->focal;
  A.ppx = m->ppx;
  A.ppy = m->ppy;
  A.max_iteration = m->params.p3p_max_iteration;
  A.min_resection_points = m->params.min_resection_points;
  A.min_inliers = m->params.min_inliers;
  A.max_n = (int)c->p3p_cap;
  A.ws_key = c->d_p3p_ws_key;
  A.ws_idx = c->d_p3p_ws_idx;
  A.ws_terms = c->d_p3p_terms;
  A.refine_pose = m->params.refine_pose;
  // a context that is not alone on the map's GPU (other contexts have work queued, ctx_mark_busy) spends fewer
  // speculative hypotheses per round; SFMLOC_P3P_ADAPTIVE=0/1 overrides (measurements)
  static const int env_adaptive = [] {
    const char *e = getenv("SFMLOC_P3P_ADAPTIVE");
    return e ? atoi(e) : -1;
  }();
  static const int env_filter = [] { const char *e = getenv("SFMLOC_P3P_FILTER"); return e ? atoi(e) : 1; }();
  A.nfa_filter = env_filter;  // (0: every model is sorted, as before round 3 -- comparison runs and tests)
  static const int env_filter_p = [] { const char *e = getenv("SFMLOC_P3P_FILTER_MIN_P"); return e ? atoi(e) : 512; }();
  A.nfa_filter_min_p = env_filter_p;
  A.adaptive_batch = env_adaptive >= 0 ? env_adaptive : (c->k1_may_slice ? 0 : 1);
  // (SFMLOC_P3P_PREP_AHEAD: 0 never, 1 when the GPU is shared -- the default --, 2 always: comparison runs and tests)
  static const int env_prep = [] { const char *e = getenv("SFMLOC_P3P_PREP_AHEAD"); return e ? atoi(e) : 1; }();
  A.prep_ahead = env_prep == 2 || (env_prep == 1 && A.adaptive_batch);
  static const int env_quarters = [] { const char *e = getenv("SFMLOC_P3P_ADAPT_QUARTERS"); return e ? atoi(e) : 12; }();
  static const int env_floor = [] { const char *e = getenv("SFMLOC_P3P_ADAPT_FLOOR"); return e ? atoi(e) : 64; }();
  static const int env_skip = [] { const char *e = getenv("SFMLOC_P3P_SKIP_OVERTAKEN"); return e ? atoi(e) : 1; }();
  A.skip_overtaken = env_skip;  // (0: every hypothesis of a round is evaluated, as before round 4 -- comparison runs)
  A.adapt_quarters = env_quarters;
  A.adapt_floor = env_floor;
  A.seed = m->params.seed;
  A.stream = 0;
  return A;
}

int launch_p3p_init(Ctx *c) {
  if (c->p3p_init_fused) {  // k_match_set_finish has done it
    c->p3p_init_fused = false;
    return SFMLOC_OK;
  }
  P3pArgs A = make_p3p_args(c);
  sfm_launch<P3pInitBody>(c, k_p3p_init, dim3(1), dim3(kThreads), 0, A);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

int launch_p3p_round(Ctx *c, int batch) {
  P3pArgs A = make_p3p_args(c);
  const size_t lds = sizeof(P3pShared);
  static_assert(sizeof(P3pReplayShared) <= sizeof(P3pShared), "the replay reuses the round's LDS");
  // (a function-local static with an initialiser: set once, safely, whichever host thread gets here first)
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p3p_round),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(P3pShared));
  SFM_HIP(attr);
  // from 513 features on a query may have that many correspondences: four workgroups per hypothesis, one model each
  // (p3p_eval_hypothesis); the result slots then are 4 b + m, of which there are kP3pSlots
  // -- and only while queries of this map HAVE had that many lately (Map::p3p_wide_credit, refreshed by every finished
  // query that did): the idle workgroups of a wide launch cost a small query ~15 us and 3 % of the throughput, and the
  // host cannot know the match set's size when it queues the rounds.  Either launch shape gives the same bits.
  // (SFMLOC_P3P_WIDE_ALONE=1: also for any query alone on the GPU -- measured and not the default: with 4 x 256 workgroups
  // the chip is NOT idle, four waves per SIMD share its f64 pipe and a headline round takes 47 us instead of 29,
  // profiles/r04_k5_forms.txt)
  static const int env_wide_alone = [] { const char *e = getenv("SFMLOC_P3P_WIDE_ALONE"); return e ? atoi(e) : 0; }();
  const int wide = ((c->p3p_query_n > 512 && c->map->p3p_wide_credit.load(std::memory_order_relaxed) > 0) ||
                    (env_wide_alone == 2 || (env_wide_alone == 1 && c->k1_may_slice && !c->stream.gang))) ? 1 : 0;
  if (wide && batch > kP3pSlots / 4) batch = kP3pSlots / 4;
  // (wide rounds are nominally 128 hypotheses, not 256: on large sets most rounds evaluate 16 .. 64 -- p3p_next_batch_limit --
  // and every workgroup of the 4 x batch launched has to be dispatched with its LDS and registers even to find that it has
  // nothing to do: image-in frames 0.86 -> 0.71 ms of PnP alone, 1 229 -> 1 302 images/s, profiles/r04_k5_forms.txt)
  static const int env_wide_batch = [] { const char *e = getenv("SFMLOC_P3P_WIDE_BATCH"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= 256 ? v : 128; }();
  if (wide && batch > env_wide_batch) batch = env_wide_batch;
  // (a wide launch sized for fewer hypotheses than the nominal batch, each workgroup taking several in turn, was tried:
  // the empty workgroups of a 4 x 256 launch cost ~20-50 us per round on large sets, but the loop cost the small form's
  // text 10 % of a headline round; not kept, profiles/r04_k5_forms.txt)
  const int wide_groups = batch;
  // the small form (above, at P3pShared) when this query's rounds were queued on that prediction and nothing has refuted it
  if (!wide && c->p3p_small) {
    constexpr size_t lds_small = std::max(offsetof(P3pShared, idx) + 4 * kP3pSmallN * sizeof(uint32_t), sizeof(P3pReplayShared));
    sfm_launch<P3pRoundSmallBody>(c, k_p3p_round_small, dim3(batch), dim3(kThreads), (uint32_t)lds_small, A, batch, 0);
  } else {
    sfm_launch<P3pRoundBody>(c, k_p3p_round, dim3(wide ? 4 * wide_groups : batch), dim3(kThreads), (uint32_t)lds, A, batch, wide);
  }
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// the sequential form (k_p3p_seq): one workgroup of `nw` waves takes the query's whole AC-RANSAC.  cap bounds the LDS
// index lists: the query's feature count rounded up (a query has at most one correspondence per feature).
int p3p_seq_waves() {
  static const int v = [] {
    const char *e = getenv("SFMLOC_P3P_SEQ_WAVES");
    const int w = e ? atoi(e) : 8;
    return w == 16 ? 16 : (w == 4 ? 4 : 8);
  }();
  return v;
}
template <int NW>
static int launch_p3p_seq_nw(Ctx *c, const P3pArgs &A, int cap) {
  const uint32_t lds = p3p_seq_lds(NW, cap).total;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(k_p3p_seq<NW>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)p3p_seq_lds(NW, kP3pMaxN).total);
  SFM_HIP(attr);
  sfm_launch<P3pSeqBody<NW>>(c, k_p3p_seq<NW>, dim3(1), dim3(NW * 64), lds, A, cap);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}
int launch_p3p_seq(Ctx *c) {
  P3pArgs A = make_p3p_args(c);
  int cap = 256;
  while (cap < (int)c->p3p_query_n && cap < kP3pMaxN) cap <<= 1;
  switch (p3p_seq_waves()) {
    case 16: return launch_p3p_seq_nw<16>(c, A, cap);
    case 4: return launch_p3p_seq_nw<4>(c, A, cap);
    default: return launch_p3p_seq_nw<8>(c, A, cap);
  }
}

int launch_p3p_finish(Ctx *c) {
  P3pArgs A = make_p3p_args(c);
  sfm_launch<P3pFinishBody>(c, k_p3p_finish, dim3(1), dim3(kThreads), 0, A);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}


#ifdef SFMLOC_STAMPS
int debug_stamps_read(int which, unsigned long long *out, size_t n) {
  hipDeviceSynchronize();
  hipError_t e = which == 0   ? hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_p3p), n * 8)
                 : which == 1 ? hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_sel), n * 8)
                              : hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_f), n * 8);
  return e == hipSuccess ? 0 : -1;
}
int debug_stamps_clear() {
  static unsigned long long z[16 * 256 * 8];
  hipDeviceSynchronize();
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_p3p), z, sizeof(z));
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_sel), z, 16 * 8 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_f), z, 256 * 64 * 8);
  return 0;
}
#endif
}  // namespace sfmloc

#ifdef SFMLOC_STAMPS
extern "C" int sfmloc_debug_stamps_read(int which, unsigned long long *out, unsigned long long n) {
  return sfmloc::debug_stamps_read(which, out, (size_t)n);
}
extern "C" int sfmloc_debug_stamps_clear(void) { return sfmloc::debug_stamps_clear(); }
#endif
