// K7 / K8: the bag-of-words view shortlist for gfx950.
//
//   K8  k_bow_dist + k_bow_topk   selectViewByBoF (BoFUtils.cpp:27-68) made exact: squared L2 between the query's
//                                 BoW vector and every candidate view's (V x dim f32, resident in HBM), k smallest,
//                                 ties to the lower view index.  HBM-bound: V*dim*4 bytes per query (20 MB at
//                                 V = 10 k), one wave per view, coalesced 256-B reads.
//   K7  k_bof_assign + k_bof_finish  PcaWrapper::calcPcaProject (PcaWrapper.cpp:67-89) + BoFSpatialPyramids::calcBoF
//                                 (BoFSpatialPyramids.cpp:108-302) on the query's dense descriptors.
// float32 sums run in the fixed order the oracle documents (64 strided partials + butterfly for K8, sequential
// for K7) so that ranks and bins can be compared bit for bit.
#include <math.h>

#include "chain_device.h"
#include "sfmloc_internal.h"

namespace sfmloc {
namespace {

struct BowDistBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ bow, uint32_t dim,
                                          const uint32_t *__restrict__ cand, uint32_t n_cand,
                                          const float *__restrict__ query, uint32_t *__restrict__ dist_bits) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_cand) return;
    const uint32_t v = cand ? cand[p] : p;
    const float *row = bow + (size_t)v * dim;
    float s = 0.0f;
    for (uint32_t i = lane; i < dim; i += 64) {
      const float d = row[i] - query[i];
      const float d2 = d * d;
      s = s + d2;
    }
    for (int stride = 32; stride >= 1; stride >>= 1) s = s + __shfl_xor(s, stride, 64);
    if (lane == 0) dist_bits[p] = __float_as_uint(s);
  }
};
__global__ __launch_bounds__(256) void k_bow_dist(const float *__restrict__ bow, uint32_t dim,
                                          const uint32_t *__restrict__ cand, uint32_t n_cand,
                                          const float *__restrict__ query, uint32_t *__restrict__ dist_bits) {
  BowDistBody::run(bow, dim, cand, n_cand, query, dist_bits);
}

// exclusive prefix sum of one value per thread over a 1024-thread workgroup (wave shuffles + one LDS hop); every
// thread also gets the grand total
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *wave_tot /*[16] LDS*/, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc += o;
  }
  __syncthreads();  // wave_tot of an earlier call is no longer read
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t before = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; ++w) {
    const uint32_t t = wave_tot[w];
    if (w < wave) before += t;
    tot += t;
  }
  *total = tot;
  return before + inc - v;
}

// One 1024-thread workgroup: exact k-th smallest distance by a 3-pass radix select on the float bits
// (11+11+10), then an order-preserving compaction of {dist < T} plus the first (k - #less) of {dist == T}.
// Every search over the histogram and every scan over the per-thread counts is a block-wide scan: the first version
// left them to thread 0 and spent 130 us on 10 000 views, most of it in those loops.
struct BowTopkBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(const uint32_t *__restrict__ dist_bits, uint32_t n,
                                           const uint32_t *__restrict__ cand, uint32_t k,
                                           uint32_t *__restrict__ out_sel, ChainArgs chain) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t sh_prefix, sh_rank;
    const uint32_t tid = threadIdx.x;
    if (k > n) k = n;
    if (k == 0) return;  // (the launcher never chains an empty shortlist)
    const uint32_t k_sel = k;
    uint32_t prefix = 0, rank = k;  // the k-th smallest (1-based) within the elements matching `prefix`
    const int shifts[3] = {21, 10, 0};
    const uint32_t widths[3] = {11, 11, 10};
    // (up to kRegVals values per thread -- 16 384 views -- are fetched ONCE, side by side, and the three passes count from
    // registers: as a loop over global memory each pass was a chain of n / 1024 load -> LDS-atomic steps, ~6 us of a lone
    // query's 27 us in this kernel per pass)
    constexpr uint32_t kRegVals = 16;
    const bool in_regs = n <= kRegVals * 1024u;
    uint32_t xs[kRegVals];
    if (in_regs) {
#pragma unroll
      for (uint32_t u = 0; u < kRegVals; ++u) xs[u] = tid + 1024u * u < n ? dist_bits[tid + 1024u * u] : 0u;
    }
    for (int pass = 0; pass < 3; ++pass) {
      const uint32_t bins = 1u << widths[pass];
      for (uint32_t b = tid; b < bins; b += 1024) hist[b] = 0;
      __syncthreads();
      const uint32_t hi_shift = shifts[pass] + widths[pass];
      if (in_regs) {
#pragma unroll
        for (uint32_t u = 0; u < kRegVals; ++u) {
          const uint32_t x = xs[u];
          const bool match = (pass == 0) || ((x >> hi_shift) == (prefix >> hi_shift));
          if (tid + 1024u * u < n && match) atomicAdd(&hist[(x >> shifts[pass]) & (bins - 1)], 1u);
        }
      } else {
        for (uint32_t i = tid; i < n; i += 1024) {
          const uint32_t x = dist_bits[i];
          const bool match = (pass == 0) || ((x >> hi_shift) == (prefix >> hi_shift));
          if (match) atomicAdd(&hist[(x >> shifts[pass]) & (bins - 1)], 1u);
        }
      }
      __syncthreads();
      // the bin holding the rank-th element: thread t owns bins 2t, 2t+1 (1024 bins in the last pass: bin t, and 0)
      const uint32_t per = bins / 1024;  // 2 or 1
      const uint32_t h0 = hist[tid * per], h1 = per == 2 ? hist[tid * per + 1] : 0u;
      uint32_t total;
      const uint32_t before = block_exclusive_scan_1024(h0 + h1, wave_tot, &total);
      if (before < rank && rank <= before + h0 + h1) {  // exactly one thread (total >= rank by construction)
        const bool second = rank > before + h0;
        sh_prefix = prefix | ((tid * per + (second ? 1u : 0u)) << shifts[pass]);
        sh_rank = rank - before - (second ? h0 : 0u);
      }
      __syncthreads();
      prefix = sh_prefix;
      rank = sh_rank;
      __syncthreads();
    }
    const uint32_t T = prefix;  // bit pattern of the k-th smallest distance; `rank` of the equal ones are taken
    // contiguous chunk per thread keeps index order
    const uint32_t chunk = (n + 1023) / 1024;
    const uint32_t lo = min(n, tid * chunk), hi = min(n, lo + chunk);
    uint32_t c_eq = 0;
    for (uint32_t i = lo; i < hi; ++i) c_eq += (dist_bits[i] == T);
    uint32_t total;
    const uint32_t eq_before = block_exclusive_scan_1024(c_eq, wave_tot, &total);
    uint32_t c_sel = 0;
    {
      uint32_t e = eq_before;
      for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t x = dist_bits[i];
        if (x < T) ++c_sel;
        else if (x == T) {
          if (e < rank) ++c_sel;
          ++e;
        }
      }
    }
    uint32_t pos = block_exclusive_scan_1024(c_sel, wave_tot, &total);
    uint32_t e = eq_before;
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t x = dist_bits[i];
      bool take = x < T;
      if (x == T) {
        take = e < rank;
        ++e;
      }
      if (take) out_sel[pos++] = cand ? cand[i] : i;
    }
    if (chain.keys_out) {  // sharded shortlist: the selection as (distance bits << 32 | global view id), padded
      __syncthreads();
      for (uint32_t i = tid; i < chain.k_out; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < k_sel) {
          const uint32_t v = out_sel[i];
          key = ((unsigned long long)dist_bits[v] << 32) | (unsigned long long)chain.key_view_id[v];
        }
        chain.keys_out[i] = key;
      }
    }
    chain_after_shortlist(chain);  // the query's counters and the block list of the k views (chain_device.h)
  }
};
__global__ __launch_bounds__(1024) void k_bow_topk(const uint32_t *__restrict__ dist_bits, uint32_t n,
                                           const uint32_t *__restrict__ cand, uint32_t k,
                                           uint32_t *__restrict__ out_sel, ChainArgs chain) {
  BowTopkBody::run(dist_bits, n, cand, k, out_sel, chain);
}

// ----- sharded shortlist -----------------------------------------------------------------------------
// Every rank ranks its own views; what travels is each rank's k best as ONE sortable 64-bit key per view,
//   key = (float32 distance bits) << 32 | global view id
// (non-negative floats order like their bit patterns; equal distances fall to the lower view id, which is how the
// unsharded k_bow_topk breaks ties because view ids ascend with the view index).  Lists shorter than k are padded
// with ~0.
struct BowKeysBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const uint32_t *__restrict__ dist_bits, const uint32_t *__restrict__ sel,
                                          uint32_t n_sel, const uint32_t *__restrict__ view_id, uint32_t k,
                                          unsigned long long *__restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    unsigned long long key = ~0ull;
    if (i < n_sel) {
      const uint32_t v = sel[i];
      key = ((unsigned long long)dist_bits[v] << 32) | (unsigned long long)view_id[v];
    }
    keys[i] = key;
  }
};
__global__ __launch_bounds__(256) void k_bow_keys(const uint32_t *__restrict__ dist_bits, const uint32_t *__restrict__ sel,
                                          uint32_t n_sel, const uint32_t *__restrict__ view_id, uint32_t k,
                                          unsigned long long *__restrict__ keys) {
  BowKeysBody::run(dist_bits, sel, n_sel, view_id, k, keys);
}

// One 1024-thread workgroup per query: the k smallest of the n_parts x k gathered keys (rank by counting: the keys
// are distinct), then those whose view id belongs to this shard (binary search in the ascending local id table) as
// ASCENDING local view indices, padded with the phantom view (index n_views) up to n_pad entries.
constexpr uint32_t kBowMergeMaxKeys = 8192;
struct BowMergeSelectBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(const unsigned long long *__restrict__ keys, uint32_t n_parts,
                                                   uint64_t part_stride, uint32_t k,
                                                   const uint32_t *__restrict__ view_id, uint32_t n_views,
                                                   uint32_t n_pad, uint32_t *__restrict__ sel_out,
                                                   ChainArgs chain) {
    // all scratch in the dynamic region (16-byte aligned base): keys [n_parts * k] u64, then this shard's winners'
    // local indices [1024] (unordered), then their count
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t tid = threadIdx.x;
    const uint32_t n = n_parts * k;
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(s_raw);
    uint32_t *s_mine = reinterpret_cast<uint32_t *>(s_raw + (size_t)((n + 1) & ~1u) * 8);
    uint32_t &s_n_mine = s_mine[1024];
    if (tid == 0) s_n_mine = 0;
    for (uint32_t i = tid; i < n; i += 1024) s_keys[i] = keys[(uint64_t)(i / k) * part_stride + (i % k)];
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) {
      const unsigned long long mine = s_keys[i];
      if (mine == ~0ull) continue;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; ++j) rank += (s_keys[j] < mine);
      if (rank >= k) continue;
      const uint32_t id = (uint32_t)mine;
      uint32_t lo = 0, hi = n_views;  // first local view with id >= the key's
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (view_id[mid] < id) lo = mid + 1;
        else hi = mid;
      }
      if (lo < n_views && view_id[lo] == id) {
        const uint32_t slot = atomicAdd(&s_n_mine, 1u);
        if (slot < 1024) s_mine[slot] = lo;
      }
    }
    __syncthreads();
    const uint32_t n_mine = min(min(s_n_mine, 1024u), n_pad);
    for (uint32_t i = tid; i < n_pad; i += 1024) sel_out[i] = n_views;  // phantom padding
    __syncthreads();
    if (tid < n_mine) {
      const uint32_t v = s_mine[tid];
      uint32_t pos = 0;
      for (uint32_t j = 0; j < n_mine; ++j) pos += (s_mine[j] < v);
      sel_out[pos] = v;
    }
    chain_after_shortlist(chain);
  }
};
__global__ __launch_bounds__(1024) void k_bow_merge_select(const unsigned long long *__restrict__ keys, uint32_t n_parts,
                                                   uint64_t part_stride, uint32_t k,
                                                   const uint32_t *__restrict__ view_id, uint32_t n_views,
                                                   uint32_t n_pad, uint32_t *__restrict__ sel_out,
                                                   ChainArgs chain) {
  BowMergeSelectBody::run(keys, n_parts, part_stride, k, view_id, n_views, n_pad, sel_out, chain);
}

__global__ __launch_bounds__(256) void k_bof_assign(const float *__restrict__ desc, const float *__restrict__ kxy,
                                                    int n, int in_dim, const float *__restrict__ pca_mean,
                                                    const float *__restrict__ pca_evec,
                                                    const float *__restrict__ pca_eval, int n_pca,
                                                    const float *__restrict__ centers, int K, int resized, int levels,
                                                    uint32_t *__restrict__ counts,
                                                    const uint8_t *__restrict__ desc8 /*or null: rows of 64 bytes*/) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  // the descriptor as floats: given so, or the bytes of a .desc row converted (convertTo(CV_32FC1),
  // DenseLocalFeatureWrapper.cpp:176-181: byte values 0..255 as floats)
  float xin[128];
  if (desc8) {
    for (int i = 0; i < in_dim; ++i) xin[i] = (float)desc8[(size_t)r * 64 + i];
  } else {
    for (int i = 0; i < in_dim; ++i) xin[i] = desc[(size_t)r * in_dim + i];
  }
  const float *x = xin;
  const int cdim = n_pca > 0 ? n_pca : in_dim;
  float y[128];
  if (n_pca > 0) {
    for (int d = 0; d < n_pca; ++d) {
      float acc = 0.0f;
      for (int i = 0; i < in_dim; ++i) {
        const float c = x[i] - pca_mean[i];
        const float pr = c * pca_evec[(size_t)d * in_dim + i];
        acc = acc + pr;
      }
      y[d] = acc / pca_eval[d];
    }
  } else {
    for (int i = 0; i < in_dim; ++i) y[i] = x[i];
  }
  int best = 0;
  float bestd = INFINITY;
  for (int c = 0; c < K; ++c) {
    float s = 0.0f;
    for (int i = 0; i < cdim; ++i) {
      const float d = y[i] - centers[(size_t)c * cdim + i];
      const float d2 = d * d;
      s = s + d2;
    }
    if (s < bestd) {
      bestd = s;
      best = c;
    }
  }
  const float px = kxy[2 * r], py = kxy[2 * r + 1];
  int cell0 = 0;
  for (int level = 0; level < levels; ++level) {
    const int len = level + 1;
    const int edge = resized / len;
    if (level == 2) {
      for (int cy = 0; cy < len; ++cy)
        if (px >= 0 && px < (float)resized && py >= (float)(edge * cy) && py < (float)(edge * (cy + 1)))
          atomicAdd(&counts[(size_t)K * (cell0 + cy) + best], 1u);
      cell0 += 3;
    } else {
      for (int cx = 0; cx < len; ++cx)
        for (int cy = 0; cy < len; ++cy)
          if (px >= (float)(edge * cx) && px < (float)(edge * (cx + 1)) && py >= (float)(edge * cy) &&
              py < (float)(edge * (cy + 1)))
            atomicAdd(&counts[(size_t)K * (cell0 + cy * len + cx) + best], 1u);
      cell0 += len * len;
    }
  }
}

// The same assignment with the work laid out for the machine (round 3: the one-thread-per-descriptor form above walks
// 61 x 32 + 100 x 32 dependent operations per thread out of scratch arrays -- 616 us for the 10 000 descriptors of a
// frame's dense grid, 13 % of the GPU time of the image-in path).  A workgroup takes kBofTile descriptors; every
// (descriptor, PCA component) pair and every (descriptor, centre) pair is its own thread's chain -- the SAME chain of
// float operations in the same order as above (and as the oracle), so the same bits -- with the descriptors, the PCA
// basis, the projected vectors and the centres in LDS; the nearest centre of a descriptor is then the first minimum
// over its row of distances.  Used when the model fits the LDS layout (the reference's: 61 -> 32, K = 100); any other
// model takes the form above.
constexpr int kBofTile = 32;
// (rows of the centre and distance tables are padded to an odd number of floats: with 32 floats per centre every lane of a
// wave -- one centre each -- read the same LDS bank, 64 different addresses of it: the kernel took 107 us instead of 30)
struct BofTileLds {   // sized by the launcher: floats
  static __host__ __device__ int cen_stride(int cdim) { return cdim | 1; }
  static __host__ __device__ int dist_stride(int K) { return K | 1; }
  static __host__ __device__ size_t floats(int in_dim, int n_pca, int cdim, int K) {
    return (size_t)kBofTile * in_dim + (n_pca > 0 ? (size_t)in_dim + (size_t)n_pca * in_dim + n_pca : 0) +
           (size_t)kBofTile * cdim + (size_t)K * cen_stride(cdim) + (size_t)kBofTile * dist_stride(K);
  }
};
struct BofAssignTiledBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ desc, const float *__restrict__ kxy,
                                                          int n, int in_dim, const float *__restrict__ pca_mean,
                                                          const float *__restrict__ pca_evec,
                                                          const float *__restrict__ pca_eval, int n_pca,
                                                          const float *__restrict__ centers, int K, int resized,
                                                          int levels, uint32_t *__restrict__ counts,
                                                          const uint8_t *__restrict__ desc8) {
  extern __shared__ float bof_lds[];
  const int cdim = n_pca > 0 ? n_pca : in_dim;
  float *sx = bof_lds;                                     // [tile][in_dim]
  float *smean = sx + (size_t)kBofTile * in_dim;           // [in_dim]        } only with PCA
  float *sevec = smean + (n_pca > 0 ? in_dim : 0);         // [n_pca][in_dim] }
  float *seval = sevec + (n_pca > 0 ? (size_t)n_pca * in_dim : 0);  // [n_pca]
  float *sy = seval + (n_pca > 0 ? n_pca : 0);             // [tile][cdim]
  const int cs = BofTileLds::cen_stride(cdim), ds = BofTileLds::dist_stride(K);
  float *scen = sy + (size_t)kBofTile * cdim;              // [K][cs]
  float *sdist = scen + (size_t)K * cs;                    // [tile][ds]
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * kBofTile;
  const int rows = min(kBofTile, n - r0);
  for (int e = tid; e < rows * in_dim; e += 256) {
    const int r = e / in_dim, i = e - r * in_dim;
    sx[e] = desc8 ? (float)desc8[(size_t)(r0 + r) * 64 + i] : desc[(size_t)(r0 + r) * in_dim + i];
  }
  if (n_pca > 0) {
    for (int e = tid; e < in_dim; e += 256) smean[e] = pca_mean[e];
    for (int e = tid; e < n_pca * in_dim; e += 256) sevec[e] = pca_evec[e];
    for (int e = tid; e < n_pca; e += 256) seval[e] = pca_eval[e];
  }
  for (int e = tid; e < K * cdim; e += 256) {
    const int c = e / cdim;
    scen[c * cs + (e - c * cdim)] = centers[e];
  }
  __syncthreads();
  if (n_pca > 0) {
    for (int e = tid; e < rows * n_pca; e += 256) {
      const int r = e / n_pca, d = e - r * n_pca;
      const float *x = sx + (size_t)r * in_dim, *ev = sevec + (size_t)d * in_dim;
      float acc = 0.0f;
      for (int i = 0; i < in_dim; ++i) {
        const float c = x[i] - smean[i];
        const float pr = c * ev[i];
        acc = acc + pr;
      }
      sy[(size_t)r * cdim + d] = acc / seval[d];
    }
  } else {
    for (int e = tid; e < rows * in_dim; e += 256) sy[e] = sx[e];
  }
  __syncthreads();
  for (int e = tid; e < rows * K; e += 256) {
    const int r = e / K, c = e - r * K;
    const float *y = sy + (size_t)r * cdim, *cen = scen + (size_t)c * cs;
    float sacc = 0.0f;
    for (int i = 0; i < cdim; ++i) {
      const float d = y[i] - cen[i];
      const float d2 = d * d;
      sacc = sacc + d2;
    }
    sdist[r * ds + c] = sacc;
  }
  __syncthreads();
  if (tid < rows) {
    const int r = r0 + tid;
    int best = 0;
    float bestd = INFINITY;
    for (int c = 0; c < K; ++c) {
      const float sacc = sdist[(size_t)tid * ds + c];
      if (sacc < bestd) {
        bestd = sacc;
        best = c;
      }
    }
    const float px = kxy[2 * r], py = kxy[2 * r + 1];
    int cell0 = 0;
    for (int level = 0; level < levels; ++level) {
      const int len = level + 1;
      const int edge = resized / len;
      if (level == 2) {
        for (int cy = 0; cy < len; ++cy)
          if (px >= 0 && px < (float)resized && py >= (float)(edge * cy) && py < (float)(edge * (cy + 1)))
            atomicAdd(&counts[(size_t)K * (cell0 + cy) + best], 1u);
        cell0 += 3;
      } else {
        for (int cx = 0; cx < len; ++cx)
          for (int cy = 0; cy < len; ++cy)
            if (px >= (float)(edge * cx) && px < (float)(edge * (cx + 1)) && py >= (float)(edge * cy) &&
                py < (float)(edge * (cy + 1)))
              atomicAdd(&counts[(size_t)K * (cell0 + cy * len + cx) + best], 1u);
        cell0 += len * len;
      }
    }
  }
}
};
__global__ __launch_bounds__(256) void k_bof_assign_tiled(const float *__restrict__ desc, const float *__restrict__ kxy,
                                                          int n, int in_dim, const float *__restrict__ pca_mean,
                                                          const float *__restrict__ pca_evec,
                                                          const float *__restrict__ pca_eval, int n_pca,
                                                          const float *__restrict__ centers, int K, int resized,
                                                          int levels, uint32_t *__restrict__ counts,
                                                          const uint8_t *__restrict__ desc8) {
  BofAssignTiledBody::run(desc, kxy, n, in_dim, pca_mean, pca_evec, pca_eval, n_pca, centers, K, resized, levels, counts, desc8);
}

// one thread per pyramid cell: counts -> /n -> per-cell normalisation, sequential in the reference's order
// one thread per cell (any K; the form below needs K doubles of LDS)
__global__ void k_bof_finish_serial(const uint32_t *__restrict__ counts, int n, int K, int cells, int norm_type,
                                    double *__restrict__ out, float *__restrict__ out_f32) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cells) return;
  double *h = out + (size_t)K * c;
  for (int i = 0; i < K; ++i) h[i] = (double)counts[(size_t)K * c + i] / (double)n;
  if (norm_type == 2) {
    double norm = 0.0;
    for (int i = 0; i < K; ++i) norm += h[i];
    if (norm > 0.0)
      for (int i = 0; i < K; ++i) h[i] = sqrt(h[i] / norm);
  } else if (norm_type == 1) {
    double norm = 0.0;
    for (int i = 0; i < K; ++i) norm += h[i] * h[i];
    norm = sqrt(norm);
    if (norm > 0.0)
      for (int i = 0; i < K; ++i) h[i] = h[i] / norm;
  }
  if (out_f32)
    for (int i = 0; i < K; ++i) out_f32[(size_t)K * c + i] = (float)h[i];  // BoFUtils.cpp:51-54 converts to CV_32F
}

// (round 3: one WAVE per cell.  The element-wise steps run side by side; the two sums stay what they were -- one lane
// adding the K values in index order -- so every bit is the one-thread-per-cell result's, which took 62 us: four dependent
// walks over global memory.)
struct BofFinishBody {
  static constexpr int kGangThreads = 64;
  static __device__ __forceinline__ void run(const uint32_t *__restrict__ counts, int n, int K, int cells, int norm_type,
                                                   double *__restrict__ out, float *__restrict__ out_f32) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c >= cells) return;
  extern __shared__ double bof_h[];  // [K]
  double *h = out + (size_t)K * c;
  for (int i = lane; i < K; i += 64) bof_h[i] = (double)counts[(size_t)K * c + i] / (double)n;
  __syncthreads();
  __shared__ double s_norm;
  if (norm_type == 2) {
    if (lane == 0) {
      double norm = 0.0;
      for (int i = 0; i < K; ++i) norm += bof_h[i];
      s_norm = norm;
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
      for (int i = lane; i < K; i += 64) bof_h[i] = sqrt(bof_h[i] / norm);
  } else if (norm_type == 1) {
    if (lane == 0) {
      double norm = 0.0;
      for (int i = 0; i < K; ++i) norm += bof_h[i] * bof_h[i];
      s_norm = sqrt(norm);
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
      for (int i = lane; i < K; i += 64) bof_h[i] = bof_h[i] / norm;
  }
  for (int i = lane; i < K; i += 64) {  // (a lane reads back only what it wrote itself)
    h[i] = bof_h[i];
    if (out_f32) out_f32[(size_t)K * c + i] = (float)bof_h[i];  // BoFUtils.cpp:51-54 converts to CV_32F
  }
}
};
__global__ __launch_bounds__(64) void k_bof_finish(const uint32_t *__restrict__ counts, int n, int K, int cells, int norm_type,
                                                   double *__restrict__ out, float *__restrict__ out_f32) {
  BofFinishBody::run(counts, n, K, cells, norm_type, out, out_f32);
}

}  // namespace

int launch_bow_select(Ctx *c, const float *d_query, const uint32_t *d_cand, uint32_t n_cand,
                      uint32_t k, uint32_t *d_dist_bits, uint32_t *d_out_sel, const ChainArgs *chain) {
  SFM_CHECK(!chain || (n_cand > 0 && k > 0 && k <= n_cand), SFMLOC_EINVAL, "shortlist chain on an empty shortlist");
  if (n_cand == 0 || k == 0) return SFMLOC_OK;
  Map *m = c->map;
  ChainArgs C{};
  if (chain) C = *chain;
  sfm_launch<BowDistBody>(c, k_bow_dist, dim3((n_cand + 3) / 4), dim3(256), 0, m->d_bow, m->bow_dim, d_cand, n_cand, d_query,
                          d_dist_bits);
  SFM_HIP(hipGetLastError());
  sfm_launch<BowTopkBody>(c, k_bow_topk, dim3(1), dim3(1024), 0, d_dist_bits, n_cand, d_cand, k, d_out_sel, C);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

int launch_bow_keys(Ctx *c, const float *d_query, uint32_t k, uint32_t *d_dist_bits, uint32_t *d_sel_tmp,
                    unsigned long long *d_keys_out) {
  if (k == 0) return SFMLOC_OK;
  Map *m = c->map;
  const uint32_t kk = k < m->n_views ? k : m->n_views;
  if (kk) {  // the keys come out of the selection's own workgroup (no third launch)
    ChainArgs C{};
    C.keys_out = d_keys_out;
    C.key_view_id = m->d_view_id;
    C.k_out = k;
    sfm_launch<BowDistBody>(c, k_bow_dist, dim3((m->n_views + 3) / 4), dim3(256), 0, m->d_bow, m->bow_dim,
                            (const uint32_t *)nullptr, m->n_views, d_query, d_dist_bits);
    SFM_HIP(hipGetLastError());
    sfm_launch<BowTopkBody>(c, k_bow_topk, dim3(1), dim3(1024), 0, d_dist_bits, m->n_views, (const uint32_t *)nullptr, kk,
                            d_sel_tmp, C);
    SFM_HIP(hipGetLastError());
    return SFMLOC_OK;
  }
  sfm_launch<BowKeysBody>(c, k_bow_keys, dim3((k + 255) / 256), dim3(256), 0, d_dist_bits, d_sel_tmp, kk, m->d_view_id,
                     k, d_keys_out);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

int launch_bow_merge_select(Ctx *c, const unsigned long long *d_keys, uint32_t n_parts,
                            uint64_t part_stride_keys, uint32_t k, uint32_t n_pad, uint32_t *d_sel_out,
                            const ChainArgs *chain) {
  Map *m = c->map;
  ChainArgs C{};
  if (chain) C = *chain;
  const uint64_t n = (uint64_t)n_parts * k;
  SFM_CHECK(n > 0 && n <= kBowMergeMaxKeys && k <= 1024, SFMLOC_EINVAL,
            "sharded shortlist: %u parts x %u keys (at most %u keys in all, k <= 1024)", n_parts, k, kBowMergeMaxKeys);
  const size_t lds = (size_t)((n + 1) & ~1ull) * 8 + 1024 * 4 + 16;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bow_merge_select),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(kBowMergeMaxKeys * 8 + 1024 * 4 + 16));
  SFM_HIP(attr);
  sfm_launch<BowMergeSelectBody>(c, k_bow_merge_select, dim3(1), dim3(1024), (uint32_t)lds, d_keys, n_parts, part_stride_keys,
                                 k, m->d_view_id, m->n_views, n_pad, d_sel_out, C);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

struct BofZeroBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(uint32_t *__restrict__ counts, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) counts[t] = 0u;
  }
};
__global__ __launch_bounds__(256) void k_bof_zero(uint32_t *__restrict__ counts, int n) { BofZeroBody::run(counts, n); }

// launch_bof on a gang member's stream -- or, while the member records for a session, into its record: every step a
// kernel (the counters' memset included), so that a batch of frames shares each launch (sfmloc_imgbow_compute_batch).
// Only the forms the reference's model takes (the tiled assignment, the one-wave-per-cell finish); -> SFMLOC_EINVAL for a
// model that needs the others (the caller then takes the frames one at a time through launch_bof).
int launch_bof_member(const BofModel *b, GangMember *m, const float *d_kxy, int n, uint32_t *d_counts, double *d_out,
                      float *d_out_f32, const uint8_t *d_desc8) {
  const int cells = b->cells;
  const size_t lds = BofTileLds::floats(b->in_dim, b->n_pca, b->cdim, b->K) * sizeof(float);
  if (n <= 0 || lds > 48 * 1024 || (size_t)b->K * sizeof(double) > 48 * 1024) return SFMLOC_EINVAL;
  const int nc = b->K * cells;
  sfm_launch<BofZeroBody>(m, k_bof_zero, dim3((nc + 255) / 256), dim3(256), 0, d_counts, nc);
  sfm_launch<BofAssignTiledBody>(m, k_bof_assign_tiled, dim3((n + kBofTile - 1) / kBofTile), dim3(256), (uint32_t)lds,
                                 (const float *)nullptr, d_kxy, n, b->in_dim, (const float *)b->d_pca_mean,
                                 (const float *)b->d_pca_evec, (const float *)b->d_pca_eval, b->n_pca,
                                 (const float *)b->d_centers, b->K, b->resized, b->levels, d_counts, d_desc8);
  sfm_launch<BofFinishBody>(m, k_bof_finish, dim3(cells), dim3(64), (uint32_t)((size_t)b->K * sizeof(double)),
                            (const uint32_t *)d_counts, n, b->K, cells, b->norm_type, d_out, d_out_f32);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

int launch_bof(const BofModel *b, hipStream_t s, const float *d_desc, const float *d_kxy, int n, uint32_t *d_counts,
               double *d_out, float *d_out_f32, const uint8_t *d_desc8) {
  const int cells = b->cells;
  SFM_HIP(hipMemsetAsync(d_counts, 0, (size_t)b->K * cells * sizeof(uint32_t), s));
  if (n > 0) {
    const size_t lds = BofTileLds::floats(b->in_dim, b->n_pca, b->cdim, b->K) * sizeof(float);
    static const bool tiled_ok = [] { const char *e = getenv("SFMLOC_BOF_TILED"); return !(e && atoi(e) == 0); }();
    if (tiled_ok && lds <= 60 * 1024) {  // the model fits the LDS layout (the reference's does: 45 KB)
      hipLaunchKernelGGL(k_bof_assign_tiled, dim3((n + kBofTile - 1) / kBofTile), dim3(256), lds, s, d_desc, d_kxy, n,
                         b->in_dim, b->d_pca_mean, b->d_pca_evec, b->d_pca_eval, b->n_pca, b->d_centers, b->K, b->resized,
                         b->levels, d_counts, d_desc8);
    } else {
      hipLaunchKernelGGL(k_bof_assign, dim3((n + 255) / 256), dim3(256), 0, s, d_desc, d_kxy, n, b->in_dim, b->d_pca_mean,
                         b->d_pca_evec, b->d_pca_eval, b->n_pca, b->d_centers, b->K, b->resized, b->levels, d_counts, d_desc8);
    }
    SFM_HIP(hipGetLastError());
  }
  if ((size_t)b->K * sizeof(double) <= 48 * 1024)
    hipLaunchKernelGGL(k_bof_finish, dim3(cells), dim3(64), (size_t)b->K * sizeof(double), s, d_counts, n > 0 ? n : 1, b->K,
                       cells, b->norm_type, d_out, d_out_f32);
  else
    hipLaunchKernelGGL(k_bof_finish_serial, dim3(1), dim3(64), 0, s, d_counts, n > 0 ? n : 1, b->K, cells, b->norm_type, d_out,
                       d_out_f32);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

}  // namespace sfmloc
