// C ABI of libsfmloc_hip.so (include/sfmloc.h): handles, HBM residency, stream, measurement.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#include <algorithm>
#include <new>

#include "sfmloc_internal.h"

namespace sfmloc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

template <typename T>
int dev_alloc(Map *m, T **p, size_t n) {
  *p = nullptr;
  if (n == 0) return SFMLOC_OK;
  SFM_HIP(hipMalloc((void **)p, n * sizeof(T)));
  m->hbm_bytes += n * sizeof(T);
  return SFMLOC_OK;
}

template <typename T>
int dev_upload(Map *m, T **p, const T *h, size_t n) {
  int rc = dev_alloc(m, p, n);
  if (rc) return rc;
  if (n) SFM_HIP(hipMemcpyAsync(*p, h, n * sizeof(T), hipMemcpyHostToDevice, m->stream));
  return SFMLOC_OK;
}

// number of d0 in [0,512] for which the reference's expression holds (MatchUtils.cpp:347):
//   (0.0f + distMat.at<int>(i,0)) / distMat.at<int>(i,1) < fDistRatio
// evaluated here, on the host, in float32 exactly as written there.
void build_ratio_table(float ratio, uint16_t *cnt) {
  for (int d1 = 0; d1 <= 512; ++d1) {
    int c = 0;
    for (int d0 = 0; d0 <= 512; ++d0) {
      volatile float num = 0.0f + (float)d0;
      volatile float den = (float)d1;
      volatile float r = num / den;
      if (r < ratio) c = d0 + 1;  // monotone in d0, so the accepted set is a prefix
    }
    cnt[d1] = (uint16_t)c;
  }
}

struct EventScope {
  Map *m;
  int which;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  EventScope(Map *m_, int which_) : m(m_), which(which_), on(m_->params.profile != 0) {
    if (!on) return;
    if (!m->event_pool.empty()) {
      a = m->event_pool.back().first;
      b = m->event_pool.back().second;
      m->event_pool.pop_back();
    } else {
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        on = false;
        return;
      }
    }
    hipEventRecord(a, m->stream);
  }
  ~EventScope() {
    if (!on) return;
    hipEventRecord(b, m->stream);
    m->pending_events.push_back({which, {a, b}});
  }
};

int drain_events(Map *m) {
  for (auto &pe : m->pending_events) {
    float ms = 0.f;
    SFM_HIP(hipEventSynchronize(pe.second.second));
    SFM_HIP(hipEventElapsedTime(&ms, pe.second.first, pe.second.second));
    m->stats.total_ms[pe.first] += ms;
    m->stats.launches[pe.first] += 1;
    m->event_pool.push_back(pe.second);
  }
  m->pending_events.clear();
  return SFMLOC_OK;
}

void free_map(Map *m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->stream) hipStreamSynchronize(m->stream);
  for (auto &pe : m->pending_events) {
    hipEventDestroy(pe.second.first);
    hipEventDestroy(pe.second.second);
  }
  for (auto &e : m->event_pool) {
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  void *ptrs[] = {m->d_bank,       m->d_view_off,   m->d_view_id,    m->d_kpt,        m->d_row_landmark,
                  m->d_landmark_id, m->d_landmark_X, m->d_bow,        m->d_part,       m->d_view_sel,
                  m->d_block_list,  m->d_view_count, m->d_match_i,    m->d_match_key,  m->d_ratio_cnt,
                  m->d_view_wh,     m->d_L10,        m->d_geo_count,  m->d_geo_idx,    m->d_status,
                  m->d_cand,        m->d_n_cand,     m->d_best64,     m->d_winner,     m->d_ms_n,
                  m->d_ms_qfeat,    m->d_ms_landmark, m->d_pt2d,      m->d_pt3d,       m->d_xn,
                  m->d_logc_n,      m->d_logc_k,     m->d_vec_index,  m->d_best_inl,   m->d_hyp_nfa,
                  m->d_hyp_err,     m->d_hyp_model,  m->d_hyp_k,      m->d_hyp_inl,    m->d_pair_qfeat,
                  m->d_pair_landmark, m->d_inlier_idx, m->d_p3p_state, m->d_pose};
  for (void *p : ptrs)
    if (p) hipFree(p);
  if (m->h_pinned) hipHostFree(m->h_pinned);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
}

}  // namespace
}  // namespace sfmloc

using namespace sfmloc;

extern "C" {

const char *sfmloc_last_error(void) { return g_err; }
int sfmloc_abi_version(void) { return SFMLOC_ABI_VERSION; }

int sfmloc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void sfmloc_default_params(sfmloc_params *p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->dist_ratio = 0.6f;       // localization.cpp:70
  p->ransac_round = 200;      // localization.cpp:71
  p->geom_precision = 4.0;    // localization.cpp:81
  p->bow_knn = 0;             // localization.cpp:73
  p->min_putative = 16;       // localization.cpp:56
  p->min_resection_points = 8;   // localization.cpp:57
  p->min_inliers = 10;        // localization.cpp:58
  p->p3p_max_iteration = 4096;   // OpenMVG Image_Localizer_Match_Data default
  p->seed = 0x5f3759df12345678ull;
  p->refine_pose = 0;
  p->device = 0;
  p->profile = 0;
}

int sfmloc_map_create(const sfmloc_map_desc *d, const sfmloc_params *params, sfmloc_map **out) {
  SFM_CHECK(d && out, SFMLOC_EINVAL, "sfmloc_map_create: null argument");
  *out = nullptr;
  SFM_CHECK(d->n_views > 0 && d->view_id && d->view_off, SFMLOC_EINVAL, "sfmloc_map_create: no views");
  SFM_CHECK(d->view_off[0] == 0 && d->view_off[d->n_views] == d->n_rows, SFMLOC_EINVAL,
            "sfmloc_map_create: view_off must start at 0 and end at n_rows");
  SFM_CHECK(d->n_rows < (1ull << 32) - 64, SFMLOC_EINVAL, "sfmloc_map_create: more than 2^32 rows per shard");
  SFM_CHECK(d->n_rows == 0 || d->desc, SFMLOC_EINVAL, "sfmloc_map_create: desc is null");
  for (uint32_t v = 0; v < d->n_views; ++v) {
    SFM_CHECK(d->view_off[v] <= d->view_off[v + 1], SFMLOC_EINVAL, "sfmloc_map_create: view_off not monotone at %u", v);
    SFM_CHECK(v == 0 || d->view_id[v - 1] < d->view_id[v], SFMLOC_EINVAL,
              "sfmloc_map_create: view_id must be strictly ascending at %u", v);
  }
  if (d->row_landmark) {
    SFM_CHECK(d->n_landmarks == 0 || (d->landmark_id && d->landmark_X), SFMLOC_EINVAL,
              "sfmloc_map_create: landmark arrays missing");
    for (uint64_t r = 0; r < d->n_rows; ++r)
      SFM_CHECK(d->row_landmark[r] >= -1 && d->row_landmark[r] < (int64_t)d->n_landmarks, SFMLOC_EINVAL,
                "sfmloc_map_create: row_landmark[%llu] out of range", (unsigned long long)r);
  }

  sfmloc_params p;
  if (params)
    p = *params;
  else
    sfmloc_default_params(&p);

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  SFM_CHECK(e == hipSuccess && ndev > 0, SFMLOC_ENODEV,
            "no HIP device visible (%s); this library has no CPU fallback",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  SFM_CHECK(p.device >= 0 && p.device < ndev, SFMLOC_EINVAL, "device %d out of range (0..%d)", p.device, ndev - 1);
  SFM_HIP(hipSetDevice(p.device));

  Map *m = new (std::nothrow) Map();
  SFM_CHECK(m, SFMLOC_ENOMEM, "out of host memory");
  m->device = p.device;
  m->params = p;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, p.device) == hipSuccess) m->n_cu = prop.multiProcessorCount;
  int rc = SFMLOC_OK;
#define SFM_TRY(x)       \
  do {                   \
    rc = (x);            \
    if (rc) {            \
      free_map(m);       \
      return rc;         \
    }                    \
  } while (0)
  {
    hipError_t se = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
      set_error("hipStreamCreate: %s", hipGetErrorString(se));
      free_map(m);
      return SFMLOC_EHIP;
    }
  }
  m->n_rows = d->n_rows;
  m->n_blocks = (uint32_t)((d->n_rows + kBlockRows - 1) / kBlockRows);
  m->n_views = d->n_views;
  m->n_landmarks = d->row_landmark ? d->n_landmarks : 0;
  m->h_view_id.assign(d->view_id, d->view_id + d->n_views);
  m->h_view_off.assign(d->view_off, d->view_off + d->n_views + 1);
  if (d->view_wh) m->h_view_wh.assign(d->view_wh, d->view_wh + 2 * (size_t)d->n_views);
  m->focal = d->focal;
  m->ppx = d->ppx;
  m->ppy = d->ppy;
  m->k1 = d->k1;
  m->k2 = d->k2;
  m->k3 = d->k3;

  // bank: upload row-major chunks and re-tile on the device
  const uint64_t n_pad = (uint64_t)m->n_blocks * kBlockRows;
  SFM_TRY(dev_alloc(m, &m->d_bank, (size_t)n_pad * 4));
  if (n_pad) {
    hipError_t me = hipMemsetAsync(m->d_bank, 0, n_pad * 64, m->stream);
    if (me != hipSuccess) {
      set_error("hipMemsetAsync: %s", hipGetErrorString(me));
      free_map(m);
      return SFMLOC_EHIP;
    }
    const uint64_t chunk = 16ull << 20;  // rows per staging chunk (1 GiB)
    uint4 *d_stage = nullptr;
    const uint64_t stage_rows = std::min<uint64_t>(chunk, d->n_rows);
    if (stage_rows) {
      hipError_t ae = hipMalloc((void **)&d_stage, stage_rows * 64);
      if (ae != hipSuccess) {
        set_error("hipMalloc(stage): %s", hipGetErrorString(ae));
        free_map(m);
        return SFMLOC_ENOMEM;
      }
    }
    for (uint64_t r0 = 0; r0 < d->n_rows; r0 += chunk) {
      const uint64_t n = std::min<uint64_t>(chunk, d->n_rows - r0);
      hipError_t ce = hipMemcpyAsync(d_stage, d->desc + r0 * 64, n * 64, hipMemcpyHostToDevice, m->stream);
      if (ce == hipSuccess) {
        rc = launch_tile_bank(d_stage, r0, n, m->d_bank, m->stream);
        if (!rc) ce = hipStreamSynchronize(m->stream);
      }
      if (ce != hipSuccess || rc) {
        if (ce != hipSuccess) {
          set_error("bank upload: %s", hipGetErrorString(ce));
          rc = SFMLOC_EHIP;
        }
        hipFree(d_stage);
        free_map(m);
        return rc;
      }
    }
    if (d_stage) hipFree(d_stage);
  }

  SFM_TRY(dev_upload(m, &m->d_view_off, d->view_off, (size_t)d->n_views + 1));
  SFM_TRY(dev_upload(m, &m->d_view_id, d->view_id, (size_t)d->n_views));
  if (d->kpt_xy) SFM_TRY(dev_upload(m, (float **)&m->d_kpt, d->kpt_xy, (size_t)d->n_rows * 2));
  if (d->row_landmark) {
    SFM_TRY(dev_upload(m, &m->d_row_landmark, d->row_landmark, (size_t)d->n_rows));
    SFM_TRY(dev_upload(m, &m->d_landmark_id, d->landmark_id, (size_t)d->n_landmarks));
    SFM_TRY(dev_upload(m, &m->d_landmark_X, d->landmark_X, (size_t)d->n_landmarks * 3));
  }
  if (d->bow && d->bow_dim) {
    m->bow_dim = d->bow_dim;
    SFM_TRY(dev_upload(m, &m->d_bow, d->bow, (size_t)d->n_views * d->bow_dim));
  }

  // workspace of the putative stage
  SFM_TRY(dev_alloc(m, &m->d_part, (size_t)m->max_split * n_pad));
  SFM_TRY(dev_alloc(m, &m->d_view_sel, (size_t)m->n_views));
  SFM_TRY(dev_alloc(m, &m->d_block_list, (size_t)m->n_blocks));
  SFM_TRY(dev_alloc(m, &m->d_view_count, (size_t)m->n_views));
  SFM_TRY(dev_alloc(m, &m->d_match_i, (size_t)m->n_rows));
  SFM_TRY(dev_alloc(m, &m->d_match_key, (size_t)m->n_rows));
  SFM_TRY(dev_alloc(m, &m->d_ratio_cnt, (size_t)513));
  // workspace of the geometric stages
  if (d->view_wh) SFM_TRY(dev_upload(m, &m->d_view_wh, d->view_wh, (size_t)d->n_views * 2));
  m->have_geometry = d->view_wh && d->kpt_xy && d->row_landmark && d->focal > 0.0;
  SFM_TRY(dev_alloc(m, &m->d_L10, (size_t)65538));
  SFM_TRY(launch_fill_log10(m->d_L10, 65538, m->stream));
  SFM_TRY(dev_alloc(m, &m->d_geo_count, (size_t)m->n_views));
  SFM_TRY(dev_alloc(m, &m->d_geo_idx, (size_t)m->n_rows));
  SFM_TRY(dev_alloc(m, &m->d_status, (size_t)1));
  SFM_TRY(dev_alloc(m, &m->d_cand, (size_t)m->cand_cap));
  SFM_TRY(dev_alloc(m, &m->d_n_cand, (size_t)1));
  SFM_TRY(dev_alloc(m, &m->d_best64, (size_t)65536));
  SFM_TRY(dev_alloc(m, &m->d_winner, (size_t)65536));
  SFM_TRY(dev_alloc(m, &m->d_ms_n, (size_t)1));
  SFM_TRY(dev_alloc(m, &m->d_ms_qfeat, (size_t)65536));
  SFM_TRY(dev_alloc(m, &m->d_ms_landmark, (size_t)65536));
  SFM_TRY(dev_alloc(m, &m->d_pt2d, (size_t)65536 * 2));
  SFM_TRY(dev_alloc(m, &m->d_pt3d, (size_t)65536 * 3));
  SFM_TRY(dev_alloc(m, &m->d_xn, (size_t)kP3pMaxN * 2));
  SFM_TRY(dev_alloc(m, &m->d_logc_n, (size_t)kP3pMaxN + 1));
  SFM_TRY(dev_alloc(m, &m->d_logc_k, (size_t)kP3pMaxN + 1));
  SFM_TRY(dev_alloc(m, &m->d_vec_index, (size_t)kP3pMaxN));
  SFM_TRY(dev_alloc(m, &m->d_best_inl, (size_t)kP3pMaxN));
  SFM_TRY(dev_alloc(m, &m->d_hyp_nfa, (size_t)kP3pBatchMax));
  SFM_TRY(dev_alloc(m, &m->d_hyp_err, (size_t)kP3pBatchMax));
  SFM_TRY(dev_alloc(m, &m->d_hyp_model, (size_t)kP3pBatchMax * 12));
  SFM_TRY(dev_alloc(m, &m->d_hyp_k, (size_t)kP3pBatchMax));
  SFM_TRY(dev_alloc(m, &m->d_hyp_inl, (size_t)kP3pBatchMax * kP3pMaxN));
  SFM_TRY(dev_alloc(m, &m->d_pair_qfeat, (size_t)kP3pMaxN));
  SFM_TRY(dev_alloc(m, &m->d_pair_landmark, (size_t)kP3pMaxN));
  SFM_TRY(dev_alloc(m, &m->d_inlier_idx, (size_t)kP3pMaxN));
  SFM_TRY(dev_alloc(m, &m->d_p3p_state, (size_t)1));
  SFM_TRY(dev_alloc(m, &m->d_pose, (size_t)1));
  {
    hipError_t ze = hipMemsetAsync(m->d_status, 0, sizeof(int), m->stream);
    if (ze == hipSuccess) ze = hipMemsetAsync(m->d_geo_count, 0, (size_t)m->n_views * sizeof(uint32_t), m->stream);
    if (ze == hipSuccess) ze = hipMemsetAsync(m->d_ms_n, 0, sizeof(uint32_t), m->stream);
    if (ze == hipSuccess) ze = hipMemsetAsync(m->d_pose, 0, sizeof(Pose), m->stream);
    if (ze == hipSuccess) ze = hipMemsetAsync(m->d_p3p_state, 0, sizeof(P3pState), m->stream);
    if (ze != hipSuccess) {
      set_error("workspace init: %s", hipGetErrorString(ze));
      free_map(m);
      return SFMLOC_EHIP;
    }
  }
  {
    hipError_t he = hipHostMalloc((void **)&m->h_pinned, ((size_t)m->n_views + m->n_blocks + 16) * sizeof(uint32_t),
                                  hipHostMallocDefault);
    if (he != hipSuccess) {
      set_error("hipHostMalloc: %s", hipGetErrorString(he));
      free_map(m);
      return SFMLOC_ENOMEM;
    }
  }
  {
    hipError_t he = hipStreamSynchronize(m->stream);
    if (he != hipSuccess) {
      set_error("map upload: %s", hipGetErrorString(he));
      free_map(m);
      return SFMLOC_EHIP;
    }
  }
#undef SFM_TRY
  *out = reinterpret_cast<sfmloc_map *>(m);
  return SFMLOC_OK;
}

void sfmloc_map_destroy(sfmloc_map *map) { free_map(reinterpret_cast<Map *>(map)); }

int sfmloc_map_get_info(const sfmloc_map *map, sfmloc_map_info *info) {
  SFM_CHECK(map && info, SFMLOC_EINVAL, "sfmloc_map_get_info: null argument");
  const Map *m = reinterpret_cast<const Map *>(map);
  info->n_rows = m->n_rows;
  info->n_views = m->n_views;
  info->n_landmarks = m->n_landmarks;
  info->hbm_bytes = m->hbm_bytes;
  info->device = m->device;
  return SFMLOC_OK;
}

int sfmloc_query_create(sfmloc_map *map, const uint8_t *desc, const float *kpt_xy, uint32_t n, uint32_t width,
                        uint32_t height, sfmloc_query **out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_query_create: null argument");
  *out = nullptr;
  SFM_CHECK(n == 0 || desc, SFMLOC_EINVAL, "sfmloc_query_create: desc is null");
  SFM_CHECK(n <= SFMLOC_MAX_QUERY_ROWS, SFMLOC_EINVAL, "sfmloc_query_create: %u query descriptors > %u", n,
            SFMLOC_MAX_QUERY_ROWS);
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  Query *q = new (std::nothrow) Query();
  SFM_CHECK(q, SFMLOC_ENOMEM, "out of host memory");
  q->map = m;
  q->n = n;
  q->width = width;
  q->height = height;
  const size_t n_pad = ((size_t)n + 63) / 64 * 64;
  hipError_t e = hipSuccess;
  if (n_pad) {
    e = hipMalloc((void **)&q->d_desc, n_pad * 64);
    if (e == hipSuccess) e = hipMemsetAsync(q->d_desc, 0, n_pad * 64, m->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(q->d_desc, desc, (size_t)n * 64, hipMemcpyHostToDevice, m->stream);
    std::vector<float> k6;
    if (e == hipSuccess && kpt_xy) {
      q->h_kpt.assign(kpt_xy, kpt_xy + 2 * (size_t)n);
      // The reference writes the query's keypoints to <tmp>/<base>.feat with `ostream << float`
      // (6 significant digits, AKAZEOpenCV.cpp:80-81) and the F-matrix filter reads them back through
      // Regions::Load (:106-111); pt2D keeps the unrounded values (:77-79).  Reproduce the round trip.
      k6.resize(2 * (size_t)n);
      char buf[64];
      for (size_t i = 0; i < 2 * (size_t)n; ++i) {
        snprintf(buf, sizeof(buf), "%.6g", (double)kpt_xy[i]);
        k6[i] = strtof(buf, nullptr);
      }
      e = hipMalloc((void **)&q->d_kpt, (size_t)n * sizeof(float2));
      if (e == hipSuccess) e = hipMalloc((void **)&q->d_kpt6, (size_t)n * sizeof(float2));
      if (e == hipSuccess)
        e = hipMemcpyAsync(q->d_kpt, kpt_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, m->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(q->d_kpt6, k6.data(), (size_t)n * sizeof(float2), hipMemcpyHostToDevice, m->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
  }
  if (e != hipSuccess) {
    set_error("sfmloc_query_create: %s", hipGetErrorString(e));
    if (q->d_desc) hipFree(q->d_desc);
    if (q->d_kpt) hipFree(q->d_kpt);
    if (q->d_kpt6) hipFree(q->d_kpt6);
    delete q;
    return e == hipErrorOutOfMemory ? SFMLOC_ENOMEM : SFMLOC_EHIP;
  }
  *out = reinterpret_cast<sfmloc_query *>(q);
  return SFMLOC_OK;
}

void sfmloc_query_destroy(sfmloc_query *query) {
  Query *q = reinterpret_cast<Query *>(query);
  if (!q) return;
  if (q->map) {
    hipSetDevice(q->map->device);
    hipStreamSynchronize(q->map->stream);
  }
  if (q->d_desc) hipFree(q->d_desc);
  if (q->d_kpt) hipFree(q->d_kpt);
  if (q->d_kpt6) hipFree(q->d_kpt6);
  delete q;
}

int sfmloc_match_putative(sfmloc_map *map, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel) {
  SFM_CHECK(map && query, SFMLOC_EINVAL, "sfmloc_match_putative: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_match_putative: query belongs to another map");
  SFM_HIP(hipSetDevice(m->device));

  const bool all_views = (view_sel == nullptr);
  if (all_views) n_sel = m->n_views;
  SFM_CHECK(n_sel <= m->n_views, SFMLOC_EINVAL, "sfmloc_match_putative: n_sel %u > n_views %u", n_sel, m->n_views);

  // ratio table (only rebuilt when the ratio changes)
  if (m->ratio_cnt_for != m->params.dist_ratio) {
    uint16_t tab[513];
    build_ratio_table(m->params.dist_ratio, tab);
    SFM_HIP(hipMemcpyAsync(m->d_ratio_cnt, tab, sizeof(tab), hipMemcpyHostToDevice, m->stream));
    SFM_HIP(hipStreamSynchronize(m->stream));  // tab is on the stack
    m->ratio_cnt_for = m->params.dist_ratio;
  }

  // selected views -> list of 64-row bank blocks they overlap (ascending, unique)
  uint32_t n_work_blocks = m->n_blocks;
  m->last_blocks.clear();
  if (!all_views) {
    uint32_t *h_sel = m->h_pinned;
    uint32_t *h_blk = m->h_pinned + m->n_views;
    uint32_t nb = 0;
    for (uint32_t k = 0; k < n_sel; ++k) {
      const uint32_t v = view_sel[k];
      SFM_CHECK(v < m->n_views, SFMLOC_EINVAL, "sfmloc_match_putative: view index %u out of range", v);
      SFM_CHECK(k == 0 || view_sel[k - 1] < v, SFMLOC_EINVAL,
                "sfmloc_match_putative: view_sel must be strictly ascending");
      h_sel[k] = v;
      const uint32_t r0 = m->h_view_off[v], r1 = m->h_view_off[v + 1];
      if (r1 == r0) continue;
      uint32_t b0 = r0 / kBlockRows;
      const uint32_t b1 = (r1 - 1) / kBlockRows;
      if (nb && h_blk[nb - 1] >= b0) b0 = h_blk[nb - 1] + 1;
      for (uint32_t b = b0; b <= b1; ++b) h_blk[nb++] = b;
    }
    n_work_blocks = nb;
    m->last_blocks.assign(h_blk, h_blk + nb);
    // the pinned staging area is reused by the next call: the copies must have been consumed
    if (n_sel) SFM_HIP(hipMemcpyAsync(m->d_view_sel, h_sel, n_sel * sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    if (nb) SFM_HIP(hipMemcpyAsync(m->d_block_list, h_blk, nb * sizeof(uint32_t), hipMemcpyHostToDevice, m->stream));
    SFM_HIP(hipStreamSynchronize(m->stream));
  }

  // query split: spread a short block list over the chip (partial top-2 are merged in K2)
  uint32_t split = 1;
  if (n_work_blocks && q->n >= 256) {
    const uint64_t want = (uint64_t)m->n_cu * 8;  // wave-blocks in flight we would like at least
    while (split < m->max_split && (uint64_t)n_work_blocks * split < want && q->n / (split * 2) >= 128) split *= 2;
  }

  SFM_HIP(hipMemsetAsync(m->d_view_count, 0, (size_t)m->n_views * sizeof(uint32_t), m->stream));
  m->last_split = split;
  m->last_nq = q->n;
  m->last_n_sel = n_sel;
  m->last_all_views = all_views;
  m->last_n_work_blocks = n_work_blocks;
  if (q->n == 0 || n_sel == 0) return SFMLOC_OK;

  int rc;
  {
    EventScope ev(m, SFMLOC_K_HAMMING);
    rc = launch_hamming_top2(m, q, n_work_blocks, !all_views, split);
  }
  if (rc) return rc;
  m->stats.hamming_pairs += (uint64_t)n_work_blocks * kBlockRows * q->n;
  m->stats.hamming_alg_bytes += (uint64_t)n_work_blocks * kBlockRows * 64 + (uint64_t)q->n * 64;
  {
    EventScope ev(m, SFMLOC_K_COMPACT);
    rc = launch_merge_ratio_compact(m, q, n_sel, all_views, split);
  }
  return rc;
}

int sfmloc_putative_read(sfmloc_map *map, uint32_t *view_count, uint32_t *match_i, uint32_t *match_j,
                         uint32_t *match_d, uint64_t cap) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_putative_read: null map");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  std::vector<uint32_t> cnt(m->n_views);
  SFM_HIP(hipMemcpy(cnt.data(), m->d_view_count, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (view_count) memcpy(view_count, cnt.data(), cnt.size() * sizeof(uint32_t));
  if (match_i || match_j || match_d) {
    SFM_CHECK(cap >= m->n_rows, SFMLOC_ECAP, "sfmloc_putative_read: cap %llu < n_rows %llu", (unsigned long long)cap,
              (unsigned long long)m->n_rows);
    std::vector<uint32_t> hi(m->n_rows), hk(m->n_rows);
    SFM_HIP(hipMemcpy(hi.data(), m->d_match_i, hi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SFM_HIP(hipMemcpy(hk.data(), m->d_match_key, hk.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t v = 0; v < m->n_views; ++v) {
      const uint32_t off = m->h_view_off[v];
      for (uint32_t k = 0; k < cnt[v]; ++k) {
        if (match_i) match_i[off + k] = hi[off + k];
        if (match_j) match_j[off + k] = hk[off + k] & 0xFFFFu;
        if (match_d) match_d[off + k] = hk[off + k] >> 16;
      }
    }
  }
  return SFMLOC_OK;
}

int sfmloc_putative_read_rows(sfmloc_map *map, uint32_t *best0, uint32_t *best1) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_putative_read_rows: null map");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  const uint64_t n_pad = (uint64_t)m->n_blocks * kBlockRows;
  for (uint64_t r = 0; r < m->n_rows; ++r) {
    if (best0) best0[r] = SFMLOC_NOMATCH;
    if (best1) best1[r] = SFMLOC_NOMATCH;
  }
  if (m->last_nq == 0 || m->last_n_sel == 0 || m->last_split == 0) return SFMLOC_OK;
  std::vector<uint2> part((size_t)m->last_split * n_pad);
  SFM_HIP(hipMemcpy(part.data(), m->d_part, part.size() * sizeof(uint2), hipMemcpyDeviceToHost));
  auto push = [](uint32_t &b0, uint32_t &b1, uint32_t k) {
    if (k < b0) {
      b1 = b0;
      b0 = k;
    } else if (k < b1) {
      b1 = k;
    }
  };
  auto do_block = [&](uint32_t blk) {
    for (uint32_t l = 0; l < kBlockRows; ++l) {
      const uint64_t r = (uint64_t)blk * kBlockRows + l;
      if (r >= m->n_rows) break;
      uint32_t b0 = SFMLOC_NOMATCH, b1 = SFMLOC_NOMATCH;
      for (uint32_t s = 0; s < m->last_split; ++s) {
        const uint2 p = part[(size_t)s * n_pad + r];
        push(b0, b1, p.x);
        push(b0, b1, p.y);
      }
      if (best0) best0[r] = b0;
      if (best1) best1[r] = b1;
    }
  };
  if (m->last_all_views) {
    for (uint32_t b = 0; b < m->n_blocks; ++b) do_block(b);
  } else {
    for (uint32_t b : m->last_blocks) do_block(b);
  }
  return SFMLOC_OK;
}

static int check_stage(Map *m, Query *q, const char *who) {
  SFM_CHECK(m && q, SFMLOC_EINVAL, "%s: null argument", who);
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "%s: query belongs to another map", who);
  SFM_CHECK(m->have_geometry, SFMLOC_EINVAL,
            "%s: the map was created without view sizes / keypoints / landmarks / intrinsics", who);
  SFM_CHECK(q->n == 0 || q->d_kpt, SFMLOC_EINVAL, "%s: the query was created without keypoints", who);
  SFM_CHECK(q->n == 0 || (q->width > 0 && q->height > 0), SFMLOC_EINVAL, "%s: query image size missing", who);
  SFM_CHECK(m->last_nq == q->n, SFMLOC_EINVAL, "%s: run sfmloc_match_putative with this query first", who);
  return SFMLOC_OK;
}

int sfmloc_geometric_filter(sfmloc_map *map, sfmloc_query *query) {
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  int rc = check_stage(m, q, "sfmloc_geometric_filter");
  if (rc) return rc;
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipMemsetAsync(m->d_geo_count, 0, (size_t)m->n_views * sizeof(uint32_t), m->stream));
  SFM_HIP(hipMemsetAsync(m->d_status, 0, sizeof(int), m->stream));
  if (q->n == 0 || m->last_n_sel == 0) return SFMLOC_OK;
  EventScope ev(m, SFMLOC_K_FMATRIX);
  return launch_fmatrix_filter(m, q, m->last_n_sel, m->last_all_views);
}

int sfmloc_geometric_read(sfmloc_map *map, uint32_t *geo_count, uint32_t *geo_idx, uint64_t cap) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_geometric_read: null map");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  std::vector<uint32_t> cnt(m->n_views);
  SFM_HIP(hipMemcpy(cnt.data(), m->d_geo_count, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (geo_count) memcpy(geo_count, cnt.data(), cnt.size() * sizeof(uint32_t));
  if (geo_idx) {
    SFM_CHECK(cap >= m->n_rows, SFMLOC_ECAP, "sfmloc_geometric_read: cap too small");
    std::vector<uint32_t> gi(m->n_rows);
    SFM_HIP(hipMemcpy(gi.data(), m->d_geo_idx, gi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t v = 0; v < m->n_views; ++v)
      for (uint32_t k = 0; k < cnt[v]; ++k) geo_idx[m->h_view_off[v] + k] = gi[m->h_view_off[v] + k];
  }
  int st = 0;
  SFM_HIP(hipMemcpy(&st, m->d_status, sizeof(int), hipMemcpyDeviceToHost));
  SFM_CHECK((st & 1) == 0, SFMLOC_ECAP, "a view has more than 2048 putative matches (F-matrix workspace)");
  return SFMLOC_OK;
}

int sfmloc_match_set(sfmloc_map *map, sfmloc_query *query) {
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  int rc = check_stage(m, q, "sfmloc_match_set");
  if (rc) return rc;
  SFM_HIP(hipSetDevice(m->device));
  EventScope ev(m, SFMLOC_K_MATCHSET);
  return launch_match_set(m, q, m->last_n_sel, m->last_all_views);
}

int sfmloc_match_set_read(sfmloc_map *map, uint32_t *n, uint32_t *qfeat, uint32_t *landmark_id, double *pt2d,
                          double *pt3d, uint32_t cap) {
  SFM_CHECK(map && n, SFMLOC_EINVAL, "sfmloc_match_set_read: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  uint32_t k = 0;
  SFM_HIP(hipMemcpy(&k, m->d_ms_n, sizeof(uint32_t), hipMemcpyDeviceToHost));
  *n = k;
  int st = 0;
  SFM_HIP(hipMemcpy(&st, m->d_status, sizeof(int), hipMemcpyDeviceToHost));
  SFM_CHECK((st & 2) == 0, SFMLOC_ECAP, "more than %u 2D-3D candidates (match-set workspace)", m->cand_cap);
  if (k == 0) return SFMLOC_OK;
  SFM_CHECK(cap >= k, SFMLOC_ECAP, "sfmloc_match_set_read: cap %u < %u", cap, k);
  if (qfeat) SFM_HIP(hipMemcpy(qfeat, m->d_ms_qfeat, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (landmark_id) SFM_HIP(hipMemcpy(landmark_id, m->d_ms_landmark, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (pt2d) SFM_HIP(hipMemcpy(pt2d, m->d_pt2d, (size_t)k * 2 * sizeof(double), hipMemcpyDeviceToHost));
  if (pt3d) SFM_HIP(hipMemcpy(pt3d, m->d_pt3d, (size_t)k * 3 * sizeof(double), hipMemcpyDeviceToHost));
  return SFMLOC_OK;
}

// Enqueues the P3P AC-RANSAC rounds.  The device-side state machine makes surplus rounds no-ops, so a fixed
// number is enqueued without synchronising; the host only looks at the state when that number is spent.
static int run_resection(Map *m) {
  int rc = launch_p3p_init(m);
  if (rc) return rc;
  P3pState st;
  int first = 1;
  for (int guard = 0; guard < 64; ++guard) {
    for (int r = 0; r < 12; ++r) {
      rc = launch_p3p_round(m, first ? 64 : kP3pBatchMax);
      first = 0;
      if (rc) return rc;
    }
    SFM_HIP(hipMemcpyAsync(&st, m->d_p3p_state, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    SFM_HIP(hipStreamSynchronize(m->stream));
    if (st.done) return SFMLOC_OK;
  }
  set_error("P3P AC-RANSAC did not finish in %d rounds (iter %d of %d)", 64 * 12, st.iter, st.n_iter);
  return SFMLOC_EHIP;
}

int sfmloc_resection(sfmloc_map *map, sfmloc_query *query) {
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  int rc = check_stage(m, q, "sfmloc_resection");
  if (rc) return rc;
  SFM_HIP(hipSetDevice(m->device));
  EventScope ev(m, SFMLOC_K_P3P);
  return run_resection(m);
}

int sfmloc_pose_read(sfmloc_map *map, sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark,
                     uint32_t *inlier_idx, uint32_t cap) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_pose_read: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  SFM_HIP(hipMemcpy(out, m->d_pose, sizeof(Pose), hipMemcpyDeviceToHost));
  SFM_CHECK((out->status & 4) == 0, SFMLOC_ECAP, "more than %d 2D-3D correspondences (P3P workspace)", kP3pMaxN);
  if (out->ok && out->n_inliers > 0) {
    const uint32_t k = (uint32_t)out->n_inliers;
    if (pair_qfeat || pair_landmark || inlier_idx)
      SFM_CHECK(cap >= k, SFMLOC_ECAP, "sfmloc_pose_read: cap %u < %u inliers", cap, k);
    if (pair_qfeat) SFM_HIP(hipMemcpy(pair_qfeat, m->d_pair_qfeat, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (pair_landmark)
      SFM_HIP(hipMemcpy(pair_landmark, m->d_pair_landmark, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (inlier_idx) SFM_HIP(hipMemcpy(inlier_idx, m->d_inlier_idx, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  return SFMLOC_OK;
}

int sfmloc_localize(sfmloc_map *map, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel,
                    sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark, uint32_t cap) {
  SFM_CHECK(map && query && out, SFMLOC_EINVAL, "sfmloc_localize: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  int rc = sfmloc_match_putative(map, query, view_sel, n_sel);
  if (rc) return rc;
  rc = check_stage(m, q, "sfmloc_localize");
  if (rc) return rc;
  rc = sfmloc_geometric_filter(map, query);
  if (rc) return rc;
  rc = sfmloc_match_set(map, query);
  if (rc) return rc;
  {
    EventScope ev(m, SFMLOC_K_P3P);
    rc = run_resection(m);
  }
  if (rc) return rc;
  rc = sfmloc_pose_read(map, out, pair_qfeat, pair_landmark, nullptr, cap);
  if (rc) return rc;
  int st = 0;
  SFM_HIP(hipMemcpy(&st, m->d_status, sizeof(int), hipMemcpyDeviceToHost));
  out->status |= st;
  SFM_CHECK((st & 3) == 0, SFMLOC_ECAP, "device workspace exceeded (status %d)", st);
  const auto t1 = clk::now();
  for (int i = 0; i < 7; ++i) out->stage_seconds[i] = 0.0;
  out->stage_seconds[6] = std::chrono::duration<double>(t1 - t0).count();  // split per stage: sfmloc_stats_read
  return SFMLOC_OK;
}

int sfmloc_debug_math(int device, int op, const double *in, int n, int in_stride, double *out, int out_stride) {
  SFM_CHECK(in && out && n > 0 && in_stride > 0 && out_stride > 0, SFMLOC_EINVAL, "sfmloc_debug_math: bad argument");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  SFM_CHECK(e == hipSuccess && ndev > 0, SFMLOC_ENODEV, "no HIP device visible; this library has no CPU fallback");
  SFM_HIP(hipSetDevice(device));
  double *d_in = nullptr, *d_out = nullptr;
  SFM_HIP(hipMalloc((void **)&d_in, (size_t)n * in_stride * sizeof(double)));
  hipError_t e2 = hipMalloc((void **)&d_out, (size_t)n * out_stride * sizeof(double));
  if (e2 != hipSuccess) {
    hipFree(d_in);
    set_error("hipMalloc: %s", hipGetErrorString(e2));
    return SFMLOC_ENOMEM;
  }
  int rc = SFMLOC_OK;
  e = hipMemcpy(d_in, in, (size_t)n * in_stride * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d_out, 0, (size_t)n * out_stride * sizeof(double));
  if (e == hipSuccess) rc = launch_debug_math(op, d_in, n, in_stride, d_out, out_stride, nullptr);
  if (e == hipSuccess && rc == SFMLOC_OK) e = hipDeviceSynchronize();
  if (e == hipSuccess && rc == SFMLOC_OK)
    e = hipMemcpy(out, d_out, (size_t)n * out_stride * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(d_in);
  hipFree(d_out);
  if (e != hipSuccess) {
    set_error("sfmloc_debug_math: %s", hipGetErrorString(e));
    return SFMLOC_EHIP;
  }
  return rc;
}

int sfmloc_sync(sfmloc_map *map) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_sync: null map");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  return SFMLOC_OK;
}

int sfmloc_stats_read(sfmloc_map *map, sfmloc_kernel_stats *out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_stats_read: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  int rc = drain_events(m);
  if (rc) return rc;
  *out = m->stats;
  return SFMLOC_OK;
}

int sfmloc_stats_reset(sfmloc_map *map) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_stats_reset: null map");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(m->stream));
  int rc = drain_events(m);
  if (rc) return rc;
  memset(&m->stats, 0, sizeof(m->stats));
  return SFMLOC_OK;
}

}  // extern "C"
