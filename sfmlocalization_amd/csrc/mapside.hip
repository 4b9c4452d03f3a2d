// Map-side matching: the reference's matchAKAZE / trackAKAZE
// (VisionLocalizeCommon/src/MatchUtils.cpp:73-152, :156-277) on the kernels of the query path.
//
// Both reference functions are "2-NN + ratio of every descriptor of image `first` among the descriptors of image
// `second`" -- exactly K1/K2 with `first` as a bank view and `second` as the query -- followed by
//   * the one-to-one filter: a train index hit by two or more rows loses ALL its hits (MatchUtils.cpp:125-143),
//   * the emit loop that stops one row early: the last row of `first` is never emitted (`i < size - 1`, :146),
//     although its hit still counts in the one-to-one test;
// trackAKAZE then chains consecutive-frame matches into longer-range pairs on the host (:223-276).
// The images of a map already sit in the tiled bank, so the "query" side is rebuilt from the bank on the device
// (k_untile_view); nothing is re-read from disk (the reference re-reads both .desc files per pair, :85-96).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <new>
#include <utility>
#include <vector>

#include "sfmloc_internal.h"

namespace sfmloc {
namespace {

// bank (tiled64) rows [row0, row0 + n) -> row-major 64-byte rows, zero padded up to n_pad rows
__global__ __launch_bounds__(256) void k_untile_view(const uint4 *__restrict__ bank, uint32_t row0, uint32_t n,
                                                     uint32_t n_pad, uint4 *__restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;  // one uint4 (quarter row) per thread
  if (t >= n_pad * 4) return;
  const uint32_t p = t >> 2, c = t & 3u;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (p < n) {
    const uint32_t r = row0 + p;
    v = bank[((uint64_t)(r >> 6) * 4 + c) * 64 + (r & 63u)];
  }
  out[t] = v;
}

// One workgroup per selected view: the one-to-one filter and the early-stopping emit loop of matchAKAZE, applied
// in place to the view's putative list (ascending i; key = d0 << 16 | j).
__global__ __launch_bounds__(256) void k_one_to_one(const uint32_t *__restrict__ view_sel, uint32_t n_sel,
                                                    const uint32_t *__restrict__ view_off, uint32_t nq,
                                                    uint32_t *__restrict__ view_count, uint32_t *__restrict__ match_i,
                                                    uint32_t *__restrict__ match_key) {
  extern __shared__ uint32_t bits[];  // [2 * words]: hit at least once | hit at least twice
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t base_s;
  const uint32_t v = view_sel ? view_sel[blockIdx.x] : blockIdx.x;
  const uint32_t off = view_off[v];
  const uint32_t n_rows = view_off[v + 1] - off;
  const uint32_t n = view_count[v];
  const uint32_t words = (nq + 31) / 32;
  uint32_t *once = bits, *twice = bits + words;
  for (uint32_t w = threadIdx.x; w < 2 * words; w += 256) bits[w] = 0;
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < n; p += 256) {
    const uint32_t j = match_key[off + p] & 0xFFFFu;
    const uint32_t bit = 1u << (j & 31u);
    const uint32_t old = atomicOr(&once[j >> 5], bit);
    if (old & bit) atomicOr(&twice[j >> 5], bit);
  }
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t p0 = 0; p0 < n; p0 += 256) {
    const uint32_t p = p0 + threadIdx.x;
    uint32_t mi = 0, mk = 0;
    bool keep = false;
    if (p < n) {
      mi = match_i[off + p];
      mk = match_key[off + p];
      const uint32_t j = mk & 0xFFFFu;
      keep = ((twice[j >> 5] >> (j & 31u)) & 1u) == 0 && mi + 1 < n_rows;
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();  // also: every read of this chunk happened before any write below
    uint32_t pre = base_s;
    for (uint32_t w = 0; w < wave; ++w) pre += wave_cnt[w];
    if (keep) {
      const uint32_t dst = pre + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      match_i[off + dst] = mi;  // dst <= p: compaction moves entries towards the front only
      match_key[off + dst] = mk;
    }
    __syncthreads();
    if (threadIdx.x == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) view_count[v] = base_s;
}

struct Matches {
  std::vector<uint32_t> I, J;       // view indices of each pair, ascending (I, J) like std::map<Pair, ...>
  std::vector<uint64_t> off;        // [pairs + 1]
  std::vector<uint32_t> mi, mj;     // IndMatch(i_, j_)
};

int query_from_view(Map *m, uint32_t v, Query **out) {
  *out = nullptr;
  SFM_CHECK(v < m->n_views, SFMLOC_EINVAL, "view index %u out of range (%u views)", v, m->n_views);
  const uint32_t r0 = m->h_view_off[v], n = m->h_view_off[v + 1] - r0;
  SFM_CHECK(n <= SFMLOC_MAX_QUERY_ROWS, SFMLOC_EINVAL, "view %u has %u descriptors > %u", v, n,
            SFMLOC_MAX_QUERY_ROWS);
  Query *q = new (std::nothrow) Query();
  SFM_CHECK(q, SFMLOC_ENOMEM, "out of host memory");
  q->map = m;
  q->n = n;
  q->width = m->h_view_wh.size() > 2 * (size_t)v + 1 ? m->h_view_wh[2 * v] : 0;
  q->height = m->h_view_wh.size() > 2 * (size_t)v + 1 ? m->h_view_wh[2 * v + 1] : 0;
  const uint32_t n_pad = (n + 63) / 64 * 64;
  hipError_t e = hipSuccess;
  hipStream_t s = m->ctx0->stream;
  if (n_pad) {
    e = hipMalloc((void **)&q->d_desc, (size_t)n_pad * 64);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_untile_view, dim3((n_pad * 4 + 255) / 256), dim3(256), 0, s, m->d_bank, r0, n, n_pad,
                         q->d_desc);
      e = hipGetLastError();
    }
    if (e == hipSuccess && m->d_kpt) {
      // the map's keypoints are the .feat values already (6 significant digits), for both uses
      e = hipMalloc((void **)&q->d_kpt, (size_t)n * sizeof(float2));
      if (e == hipSuccess) e = hipMalloc((void **)&q->d_kpt6, (size_t)n * sizeof(float2));
      if (e == hipSuccess)
        e = hipMemcpyAsync(q->d_kpt, m->d_kpt + r0, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess)
        e = hipMemcpyAsync(q->d_kpt6, m->d_kpt + r0, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) {
    set_error("sfmloc_query_from_view: %s", hipGetErrorString(e));
    if (q->d_desc) hipFree(q->d_desc);
    if (q->d_kpt) hipFree(q->d_kpt);
    if (q->d_kpt6) hipFree(q->d_kpt6);
    delete q;
    return e == hipErrorOutOfMemory ? SFMLOC_ENOMEM : SFMLOC_EHIP;
  }
  *out = q;
  return SFMLOC_OK;
}

void free_query(Query *q) {
  if (!q) return;
  if (q->d_desc) hipFree(q->d_desc);
  if (q->d_kpt) hipFree(q->d_kpt);
  if (q->d_kpt6) hipFree(q->d_kpt6);
  delete q;
}

// K1 + K2 + the one-to-one filter of every selected view (ascending indices) against q, on the map's own context
int match_one_to_one(Map *m, Query *q, const uint32_t *view_sel, uint32_t n_sel) {
  Ctx *c = m->ctx0;
  int rc = match_putative_on(c, q, view_sel, n_sel);
  if (rc) return rc;
  const bool all = (view_sel == nullptr);
  if (all) n_sel = m->n_views;
  if (n_sel == 0 || q->n == 0) return SFMLOC_OK;
  const uint32_t words = (q->n + 31) / 32;
  hipLaunchKernelGGL(k_one_to_one, dim3(n_sel), dim3(256), 2 * words * sizeof(uint32_t), c->stream,
                     all ? nullptr : c->d_view_sel, n_sel, m->d_view_off, q->n, c->d_view_count, c->d_match_i,
                     c->d_match_key);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// the (filtered) list of one view from the map's context -> host vectors
int read_view_list(Map *m, uint32_t v, std::vector<uint32_t> *mi, std::vector<uint32_t> *mj) {
  Ctx *c = m->ctx0;
  uint32_t n = 0;
  SFM_HIP(hipMemcpyAsync(&n, c->d_view_count + v, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  SFM_HIP(hipStreamSynchronize(c->stream));
  mi->resize(n);
  mj->resize(n);
  if (n == 0) return SFMLOC_OK;
  const uint32_t off = m->h_view_off[v];
  SFM_HIP(hipMemcpyAsync(mi->data(), c->d_match_i + off, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  SFM_HIP(hipMemcpyAsync(mj->data(), c->d_match_key + off, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  SFM_HIP(hipStreamSynchronize(c->stream));
  for (uint32_t k = 0; k < n; ++k) (*mj)[k] &= 0xFFFFu;
  return SFMLOC_OK;
}

typedef std::map<std::pair<uint32_t, uint32_t>, std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> PairMap;

Matches *flatten(const PairMap &pm) {
  Matches *M = new (std::nothrow) Matches();
  if (!M) return nullptr;
  M->off.push_back(0);
  for (const auto &kv : pm) {
    M->I.push_back(kv.first.first);
    M->J.push_back(kv.first.second);
    M->mi.insert(M->mi.end(), kv.second.first.begin(), kv.second.first.end());
    M->mj.insert(M->mj.end(), kv.second.second.begin(), kv.second.second.end());
    M->off.push_back(M->mi.size());
  }
  return M;
}

}  // namespace
}  // namespace sfmloc

using namespace sfmloc;

extern "C" {

int sfmloc_query_from_view(sfmloc_map *map, uint32_t view_index, sfmloc_query **out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_query_from_view: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  Query *q = nullptr;
  int rc = query_from_view(m, view_index, &q);
  *out = reinterpret_cast<sfmloc_query *>(q);
  return rc;
}

int sfmloc_match_one_to_one(sfmloc_map *map, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel) {
  SFM_CHECK(map && query, SFMLOC_EINVAL, "sfmloc_match_one_to_one: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_match_one_to_one: query belongs to another map");
  SFM_HIP(hipSetDevice(m->device));
  return match_one_to_one(m, q, view_sel, n_sel);
}

int sfmloc_match_pairs(sfmloc_map *map, const uint32_t *pairs, uint32_t n_pairs, sfmloc_matches **out) {
  SFM_CHECK(map && out && (pairs || n_pairs == 0), SFMLOC_EINVAL, "sfmloc_match_pairs: null argument");
  *out = nullptr;
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  // group by the second image: it plays the query, all its firsts are scanned in one launch
  std::map<uint32_t, std::vector<uint32_t>> by_second;
  for (uint32_t k = 0; k < n_pairs; ++k) {
    const uint32_t a = pairs[2 * k], b = pairs[2 * k + 1];
    SFM_CHECK(a < m->n_views && b < m->n_views, SFMLOC_EINVAL, "sfmloc_match_pairs: pair %u = (%u, %u) out of range", k,
              a, b);
    by_second[b].push_back(a);
  }
  PairMap pm;
  for (auto &kv : by_second) {
    const uint32_t b = kv.first;
    std::vector<uint32_t> &firsts = kv.second;
    std::sort(firsts.begin(), firsts.end());
    firsts.erase(std::unique(firsts.begin(), firsts.end()), firsts.end());
    const uint32_t nb = m->h_view_off[b + 1] - m->h_view_off[b];
    if (nb < 2) continue;  // MatchUtils.cpp:101-103
    std::vector<uint32_t> sel;
    for (uint32_t a : firsts)
      if (m->h_view_off[a + 1] - m->h_view_off[a] >= 2) sel.push_back(a);
    if (sel.empty()) continue;
    Query *q = nullptr;
    int rc = query_from_view(m, b, &q);
    if (rc) return rc;
    rc = match_one_to_one(m, q, sel.data(), (uint32_t)sel.size());
    for (size_t k = 0; rc == SFMLOC_OK && k < sel.size(); ++k) {
      std::vector<uint32_t> mi, mj;
      rc = read_view_list(m, sel[k], &mi, &mj);
      if (rc == SFMLOC_OK && !mi.empty()) pm[{sel[k], b}] = {std::move(mi), std::move(mj)};  // no entry when empty (:146-149)
    }
    hipStreamSynchronize(m->ctx0->stream);
    if (m->ctx0->last_query == q) m->ctx0->last_query = nullptr;
    free_query(q);
    if (rc) return rc;
  }
  Matches *M = flatten(pm);
  SFM_CHECK(M, SFMLOC_ENOMEM, "out of host memory");
  *out = reinterpret_cast<sfmloc_matches *>(M);
  return SFMLOC_OK;
}

int sfmloc_track(sfmloc_map *map, uint32_t max_frame_dist, sfmloc_matches **out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_track: null argument");
  *out = nullptr;
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  const uint32_t V = m->n_views;
  PairMap pm;
  if (V >= 2) {
    // consecutive frames (MatchUtils.cpp:163-221)
    for (uint32_t f = 0; f + 1 < V; ++f) {
      pm[{f, f + 1}];  // the chaining loop below touches matches[(f, f+1)] and thereby creates it (:228)
      const uint32_t n1 = m->h_view_off[f + 1] - m->h_view_off[f];
      const uint32_t n2 = m->h_view_off[f + 2] - m->h_view_off[f + 1];
      if (n1 < 2 || n2 < 2) continue;
      Query *q = nullptr;
      int rc = query_from_view(m, f + 1, &q);
      if (rc) return rc;
      rc = match_one_to_one(m, q, &f, 1);
      std::vector<uint32_t> mi, mj;
      if (rc == SFMLOC_OK) rc = read_view_list(m, f, &mi, &mj);
      hipStreamSynchronize(m->ctx0->stream);
      if (m->ctx0->last_query == q) m->ctx0->last_query = nullptr;
      free_query(q);
      if (rc) return rc;
      pm[{f, f + 1}] = {std::move(mi), std::move(mj)};
    }
    // tracks (:223-276): trackPointer[frame][i] = feature of frame+1 (later: of frameTo) that i leads to, or -1
    std::vector<std::vector<int32_t>> tp(V - 1);
    for (uint32_t f = 0; f + 1 < V; ++f) {
      tp[f].assign(m->h_view_off[f + 1] - m->h_view_off[f], -1);
      const auto &pr = pm[{f, f + 1}];
      for (size_t k = 0; k < pr.first.size(); ++k) tp[f][pr.first[k]] = (int32_t)pr.second[k];
    }
    for (uint32_t f = 0; f + 1 < V; ++f) {
      const uint64_t lim = std::min<uint64_t>((uint64_t)f + max_frame_dist, V);
      for (uint64_t to = (uint64_t)f + 2; to < lim; ++to) {
        for (size_t i = 0; i < tp[f].size(); ++i) {
          const int32_t t = tp[f][i];
          if (t != -1) {
            const int32_t nxt = tp[to - 1][(size_t)t];
            tp[f][i] = nxt;
            if (nxt != -1) {
              auto &pr = pm[{f, (uint32_t)to}];
              pr.first.push_back((uint32_t)i);
              pr.second.push_back((uint32_t)nxt);
            }
          }
        }
      }
    }
  }
  Matches *M = flatten(pm);
  SFM_CHECK(M, SFMLOC_ENOMEM, "out of host memory");
  *out = reinterpret_cast<sfmloc_matches *>(M);
  return SFMLOC_OK;
}

int sfmloc_geometric_pairs(sfmloc_map *map, const uint32_t *pairs, uint32_t n_pairs, const uint64_t *offsets,
                           const uint32_t *match_i, const uint32_t *match_j, sfmloc_matches **out) {
  SFM_CHECK(map && out && (n_pairs == 0 || (pairs && offsets)), SFMLOC_EINVAL, "sfmloc_geometric_pairs: null argument");
  *out = nullptr;
  Map *m = reinterpret_cast<Map *>(map);
  SFM_CHECK(m->d_kpt && !m->h_view_wh.empty(), SFMLOC_EINVAL,
            "sfmloc_geometric_pairs: the map was created without keypoints / view sizes");
  SFM_HIP(hipSetDevice(m->device));
  Ctx *c = m->ctx0;
  // group by the second image (the query side of K3); within a group the firsts are distinct views
  std::map<uint32_t, std::vector<uint32_t>> by_second;  // second -> pair indices
  for (uint32_t k = 0; k < n_pairs; ++k) {
    const uint32_t a = pairs[2 * k], b = pairs[2 * k + 1];
    SFM_CHECK(a < m->n_views && b < m->n_views, SFMLOC_EINVAL, "sfmloc_geometric_pairs: pair %u = (%u, %u) out of range",
              k, a, b);
    const uint64_t n = offsets[k + 1] - offsets[k];
    SFM_CHECK(offsets[k + 1] >= offsets[k] && n <= m->h_view_off[a + 1] - m->h_view_off[a], SFMLOC_EINVAL,
              "sfmloc_geometric_pairs: pair %u has %llu matches for a %u-row image", k, (unsigned long long)n,
              m->h_view_off[a + 1] - m->h_view_off[a]);
    SFM_CHECK(n == 0 || (match_i && match_j), SFMLOC_EINVAL, "sfmloc_geometric_pairs: match arrays missing");
    by_second[b].push_back(k);
  }
  PairMap pm;
  for (auto &kv : by_second) {
    const uint32_t b = kv.first;
    std::map<uint32_t, uint32_t> first_to_pair;  // ascending firsts; a repeated (a, b) keeps its last list
    for (uint32_t k : kv.second) first_to_pair[pairs[2 * k]] = k;
    Query *q = nullptr;
    int rc = query_from_view(m, b, &q);
    if (rc) return rc;
    std::vector<uint32_t> sel;
    std::vector<uint32_t> zero(1, 0);
    rc = SFMLOC_OK;
    for (auto &fp : first_to_pair) {
      const uint32_t a = fp.first, k = fp.second;
      const uint32_t n = (uint32_t)(offsets[k + 1] - offsets[k]);
      const uint32_t off = m->h_view_off[a];
      for (uint32_t t = 0; t < n && rc == SFMLOC_OK; ++t)
        if (match_i[offsets[k] + t] >= m->h_view_off[a + 1] - off || match_j[offsets[k] + t] >= q->n) {
          set_error("sfmloc_geometric_pairs: pair (%u, %u) match %u out of range", a, b, t);
          rc = SFMLOC_EINVAL;
        }
      if (rc) break;
      sel.push_back(a);
      if (n) {
        hipMemcpyAsync(c->d_match_i + off, match_i + offsets[k], n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
        hipMemcpyAsync(c->d_match_key + off, match_j + offsets[k], n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
      }
      hipMemcpyAsync(c->d_view_count + a, &n, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
      hipStreamSynchronize(c->stream);  // `n` and the caller's arrays are pageable host memory
    }
    if (rc == SFMLOC_OK) {
      hipMemcpyAsync(c->d_view_sel, sel.data(), sel.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
      hipStreamSynchronize(c->stream);
      for (uint32_t a : sel) hipMemsetAsync(c->d_geo_count + a, 0, sizeof(uint32_t), c->stream);
      hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream);
      // no >=16 rule here: OpenMVG's Robust_model_estimation takes every pair it is given (the caller has applied
      // ExtFeatAndMatch's minMatch, computeFeaturesAndMatches.cpp:211-221)
      rc = launch_fmatrix_filter(c, q, (uint32_t)sel.size(), false, 0);
    }
    const bool guided = m->params.guided_matching != 0;
    std::vector<uint32_t> passed(sel.size(), 0);
    if (rc == SFMLOC_OK && guided) {
      // which pairs passed the filter must be read before the guided lists replace the inlier counts: a pair whose
      // guided list comes out empty keeps its (empty) entry
      for (size_t t = 0; t < sel.size(); ++t)
        hipMemcpyAsync(&passed[t], c->d_geo_count + sel[t], sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
      hipStreamSynchronize(c->stream);
      rc = launch_guided_matching(c, q, (uint32_t)sel.size(), false);
    }
    for (size_t t = 0; rc == SFMLOC_OK && t < sel.size(); ++t) {
      const uint32_t a = sel[t], k = first_to_pair[a];
      uint32_t ng = 0;
      hipMemcpyAsync(&ng, c->d_geo_count + a, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
      hipStreamSynchronize(c->stream);
      if (guided) {
        if (passed[t] == 0) continue;
        auto &pr = pm[{a, b}];
        if (ng == 0) continue;
        std::vector<uint32_t> gi(ng), gj(ng);
        hipMemcpyAsync(gi.data(), c->d_geo_idx + m->h_view_off[a], ng * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        hipMemcpyAsync(gj.data(), c->d_geo_j + m->h_view_off[a], ng * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        hipStreamSynchronize(c->stream);
        pr.first.assign(gi.begin(), gi.end());
        pr.second.assign(gj.begin(), gj.end());
        continue;
      }
      if (ng == 0) continue;  // pairs that fail the filter get no entry
      std::vector<uint32_t> gi(ng);
      hipMemcpyAsync(gi.data(), c->d_geo_idx + m->h_view_off[a], ng * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
      hipStreamSynchronize(c->stream);
      auto &pr = pm[{a, b}];
      for (uint32_t g : gi) {  // AC-RANSAC's inlier order, indices into the putative list
        pr.first.push_back(match_i[offsets[k] + g]);
        pr.second.push_back(match_j[offsets[k] + g]);
      }
    }
    int status = 0;
    hipMemcpy(&status, c->d_status, sizeof(int), hipMemcpyDeviceToHost);
    if (c->last_query == q) c->last_query = nullptr;
    free_query(q);
    if (rc) return rc;
    SFM_CHECK((status & 1) == 0, SFMLOC_ECAP, "a pair has more than 65536 putative matches");
  }
  Matches *M = flatten(pm);
  SFM_CHECK(M, SFMLOC_ENOMEM, "out of host memory");
  *out = reinterpret_cast<sfmloc_matches *>(M);
  return SFMLOC_OK;
}

uint32_t sfmloc_matches_pairs(const sfmloc_matches *mm) {
  const Matches *M = reinterpret_cast<const Matches *>(mm);
  return M ? (uint32_t)M->I.size() : 0;
}

int sfmloc_matches_pair(const sfmloc_matches *mm, uint32_t k, uint32_t *view_i, uint32_t *view_j, uint32_t *n) {
  const Matches *M = reinterpret_cast<const Matches *>(mm);
  SFM_CHECK(M && k < M->I.size(), SFMLOC_EINVAL, "sfmloc_matches_pair: pair %u out of range", k);
  if (view_i) *view_i = M->I[k];
  if (view_j) *view_j = M->J[k];
  if (n) *n = (uint32_t)(M->off[k + 1] - M->off[k]);
  return SFMLOC_OK;
}

int sfmloc_matches_read(const sfmloc_matches *mm, uint32_t k, uint32_t *i, uint32_t *j, uint32_t cap) {
  const Matches *M = reinterpret_cast<const Matches *>(mm);
  SFM_CHECK(M && k < M->I.size(), SFMLOC_EINVAL, "sfmloc_matches_read: pair %u out of range", k);
  const uint64_t n = M->off[k + 1] - M->off[k];
  SFM_CHECK(cap >= n, SFMLOC_ECAP, "sfmloc_matches_read: cap %u < %llu matches", cap, (unsigned long long)n);
  for (uint64_t t = 0; t < n; ++t) {
    if (i) i[t] = M->mi[M->off[k] + t];
    if (j) j[t] = M->mj[M->off[k] + t];
  }
  return SFMLOC_OK;
}

void sfmloc_matches_destroy(sfmloc_matches *mm) { delete reinterpret_cast<Matches *>(mm); }

}  // extern "C"
