// Guided matching (-gm): GeometricFilter_FMatrix_AC::Geometry_guided_matching of OpenMVG 1.1 as
// ImageCollectionGeometricFilter::Robust_model_estimation(..., b_guided_matching = true) applies it to every pair
// that passed the AC-RANSAC filter (reference call site: MatchUtils.cpp:412-416; -gm is the default of the map-building
// front end, ReconstructParam.py:71 -> reconstructGraph.py:155-163).  For a (view I, query J) pair:
//   m_F = N2^T F N1, threshold = Square(sqrt(errorMax) / N2(0,0)); for every feature i of I, over ALL features j of J in
//   index order, those with EpipolarDistanceError(m_F, x_i, x_j) < threshold compete by the squared Hamming distance;
//   best / second best with strict <; (i, best j) is kept iff a second candidate exists and best < Square(0.6) * second.
// The pair's matches are REPLACED by these, one per i, ascending i.  OpenMVG's source is not in this image: restated
// from its published algorithm, bit-identical to oracle/sfm_oracle_geom.c orc_guided_match (parity unpinned).
//
//   k_guided_rows     one thread per feature of I (its descriptor and epipolar line in registers), query positions
//                     staged through LDS tile by tile; f64 throughout, the division included, so that the band test
//                     rounds as the oracle's.  ~1-2 % of the (i, j) pairs fall in the band and pay a 64-byte
//                     descriptor read + 16 popcounts.
//   k_guided_compact  one wave per view: ordered compaction of the per-row results into (geo_idx, geo_j, geo_count).
#include "sfmloc_internal.h"

namespace sfmloc {
namespace {

constexpr int kGTile = 512;

struct GuidedArgs {
  const uint32_t *view_sel;
  uint32_t n_sel;
  const uint32_t *view_off, *view_wh;
  const uint32_t *geo_count;
  const double *geo_model;
  const uint4 *bank;
  const float2 *map_kpt;
  const uint4 *q_desc;
  const float2 *q_kpt6;
  uint32_t nq, qw, qh;
  double ratio2;
  uint32_t *guided_row;
};

struct GuidedRowsBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(GuidedArgs A) {
    __shared__ double s_q[kGTile][2];
    __shared__ double s_F[9];
    __shared__ double s_th;
    const uint32_t tid = threadIdx.x;
    const uint32_t v = A.view_sel ? A.view_sel[blockIdx.x] : blockIdx.x;
    if (A.geo_count[v] == 0) return;  // the view did not pass the F-matrix filter: nothing to guide
    const uint32_t off = A.view_off[v], n_v = A.view_off[v + 1] - off;
    if (blockIdx.y * 256 >= n_v) return;
    if (tid == 0) {
      // ACKernelAdaptor::Unnormalize / unormalizeError with NormalizePoints(x, w, h): N = [s 0 -w s/2; 0 s -h s/2; 0 0 1]
      const double *gm = A.geo_model + 10 * (size_t)v;
      const int w1 = (int)A.view_wh[2 * v], h1 = (int)A.view_wh[2 * v + 1];
      const int w2 = (int)A.qw, h2 = (int)A.qh;
      const double s1 = 1.0 / sqrt((double)(w1 * h1)), s2 = 1.0 / sqrt((double)(w2 * h2));
      const double N1[9] = {s1, 0.0, -0.5 * (double)w1 * s1, 0.0, s1, -0.5 * (double)h1 * s1, 0.0, 0.0, 1.0};
      const double N2[9] = {s2, 0.0, -0.5 * (double)w2 * s2, 0.0, s2, -0.5 * (double)h2 * s2, 0.0, 0.0, 1.0};
      double T[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[3 * r + c] = (N2[r] * gm[c] + N2[3 + r] * gm[3 + c]) + N2[6 + r] * gm[6 + c];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          s_F[3 * r + c] = (T[3 * r] * N1[c] + T[3 * r + 1] * N1[3 + c]) + T[3 * r + 2] * N1[6 + c];
      const double pr = sqrt(gm[9]) / s2;
      s_th = pr * pr;
    }
    __syncthreads();
    const double th = s_th;
    const uint32_t row = blockIdx.y * 256 + tid;
    const bool live = row < n_v;
    uint32_t b[16];
    double l0 = 0.0, l1 = 0.0, l2 = 0.0, den = 1.0;
    if (live) {
      const uint64_t r = (uint64_t)off + row;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 w = A.bank[((r >> 6) * 4 + c) * 64 + (r & 63u)];
        b[4 * c] = w.x, b[4 * c + 1] = w.y, b[4 * c + 2] = w.z, b[4 * c + 3] = w.w;
      }
      const float2 p = A.map_kpt[r];
      const double x = (double)p.x, y = (double)p.y;
      l0 = (s_F[0] * x + s_F[1] * y) + s_F[2];
      l1 = (s_F[3] * x + s_F[4] * y) + s_F[5];
      l2 = (s_F[6] * x + s_F[7] * y) + s_F[8];
      den = l0 * l0 + l1 * l1;
    }
    uint32_t bd = 0xFFFFFFFFu, sbd = 0xFFFFFFFFu, idx = 0;  // Hamming distances (their squares order the same way)
    for (uint32_t j0 = 0; j0 < A.nq; j0 += kGTile) {
      const uint32_t cnt = min((uint32_t)kGTile, A.nq - j0);
      __syncthreads();
      for (uint32_t t = tid; t < cnt; t += 256) {
        const float2 q = A.q_kpt6[j0 + t];
        s_q[t][0] = (double)q.x;
        s_q[t][1] = (double)q.y;
      }
      __syncthreads();
      if (!live) continue;
      for (uint32_t t = 0; t < cnt; ++t) {
        const double num = (l0 * s_q[t][0] + l1 * s_q[t][1]) + l2;
        const double e = (num * num) / den;
        if (e < th) {
          const uint32_t j = j0 + t;
          uint32_t d = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const uint4 w = A.q_desc[(uint64_t)j * 4 + c];
            d += __builtin_popcount(b[4 * c] ^ w.x) + __builtin_popcount(b[4 * c + 1] ^ w.y) +
                 __builtin_popcount(b[4 * c + 2] ^ w.z) + __builtin_popcount(b[4 * c + 3] ^ w.w);
          }
          if (d < bd) {
            idx = j;
            sbd = bd;
            bd = d;
          } else if (d < sbd) {
            sbd = d;
          }
        }
      }
    }
    if (live) {
      // distanceRatio<double>::isValid on the squared distances
      const bool ok = sbd != 0xFFFFFFFFu && (double)(bd * bd) < A.ratio2 * (double)(sbd * sbd);
      A.guided_row[(uint64_t)off + row] = ok ? idx : SFMLOC_NOMATCH;
    }
  }
};
__global__ __launch_bounds__(256) void k_guided_rows(GuidedArgs A) {
  GuidedRowsBody::run(A);
}

struct GuidedCompactBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const uint32_t *__restrict__ view_sel, uint32_t n_sel,
                                                const uint32_t *__restrict__ view_off,
                                                const uint32_t *__restrict__ guided_row,
                                                uint32_t *__restrict__ geo_count, uint32_t *__restrict__ geo_idx,
                                                uint32_t *__restrict__ geo_j) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= n_sel) return;
    const uint32_t v = view_sel ? view_sel[gw] : gw;
    if (geo_count[v] == 0) return;
    const uint32_t off = view_off[v], end = view_off[v + 1];
    uint32_t base = 0;
    for (uint32_t r0 = off; r0 < end; r0 += 64) {
      const uint32_t r = r0 + lane;
      const uint32_t j = r < end ? guided_row[r] : SFMLOC_NOMATCH;
      const bool has = j != SFMLOC_NOMATCH;
      const unsigned long long mask = __ballot(has);
      if (has) {
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        geo_idx[off + pos] = r - off;
        geo_j[off + pos] = j;
      }
      base += (uint32_t)__popcll(mask);
    }
    if (lane == 0) geo_count[v] = base;
  }
};
__global__ __launch_bounds__(256) void k_guided_compact(const uint32_t *__restrict__ view_sel, uint32_t n_sel,
                                                const uint32_t *__restrict__ view_off,
                                                const uint32_t *__restrict__ guided_row,
                                                uint32_t *__restrict__ geo_count, uint32_t *__restrict__ geo_idx,
                                                uint32_t *__restrict__ geo_j) {
  GuidedCompactBody::run(view_sel, n_sel, view_off, guided_row, geo_count, geo_idx, geo_j);
}

}  // namespace

int ensure_guided_workspace(Ctx *c) {
  if (c->d_geo_j && c->d_guided_row) return SFMLOC_OK;
  const size_t n = c->map->n_rows ? (size_t)c->map->n_rows : 1;
  if (!c->d_geo_j) {
    SFM_HIP(hipMalloc((void **)&c->d_geo_j, n * sizeof(uint32_t)));
    c->hbm_bytes += n * sizeof(uint32_t);
  }
  if (!c->d_guided_row) {
    SFM_HIP(hipMalloc((void **)&c->d_guided_row, n * sizeof(uint32_t)));
    c->hbm_bytes += n * sizeof(uint32_t);
  }
  return SFMLOC_OK;
}

int launch_guided_matching(Ctx *c, const Query *q, uint32_t n_sel, bool all_views) {
  Map *m = c->map;
  int rc = ensure_guided_workspace(c);
  if (rc) return rc;
  c->geo_is_pairs = true;
  if (n_sel == 0 || q->n == 0 || m->max_view_rows == 0) return SFMLOC_OK;
  GuidedArgs A;
  A.view_sel = all_views ? nullptr : c->d_view_sel;
  A.n_sel = n_sel;
  A.view_off = m->d_view_off;
  A.view_wh = m->d_view_wh;
  A.geo_count = c->d_geo_count;
  A.geo_model = c->d_geo_model;
  A.bank = m->d_bank;
  A.map_kpt = m->d_kpt;
  A.q_desc = q->d_desc;
  A.q_kpt6 = q->d_kpt6;
  A.nq = q->n;
  A.qw = q->width;
  A.qh = q->height;
  const double r = 0.6;  // Robust_model_estimation's d_distance_ratio default; hulo::geometricMatch does not pass one
  A.ratio2 = r * r;
  A.guided_row = c->d_guided_row;
  sfm_launch<GuidedRowsBody>(c, k_guided_rows, dim3(n_sel, (m->max_view_rows + 255) / 256), dim3(256), 0, A);
  SFM_HIP(hipGetLastError());
  sfm_launch<GuidedCompactBody>(c, k_guided_compact, dim3((n_sel + 3) / 4), dim3(256), 0, A.view_sel, n_sel, m->d_view_off,
                     c->d_guided_row, c->d_geo_count, c->d_geo_idx, c->d_geo_j);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

}  // namespace sfmloc
