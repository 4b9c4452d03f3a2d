// C ABI of libsfmloc_hip.so (include/sfmloc.h): handles, HBM residency, streams, measurement.
//
// Two kinds of state:
//   Map  static, read-only after creation: tiled descriptor bank, view table, keypoints, landmarks, tables;
//   Ctx  one in-flight query: a HIP stream plus the workspace of every stage.
// The stage-level entry points (sfmloc_match_putative ... sfmloc_resection) run on the map's own context;
// sfmloc_localize_begin/_end run a whole query asynchronously on any context, so that several queries
// overlap (the latency-bound RANSAC kernels of one under the VALU-bound Hamming kernel of another).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <exception>
#include <new>

#include "chain_device.h"
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

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T>
int dev_alloc(uint64_t *acct, T **p, size_t n) {
  *p = nullptr;
  if (n == 0) return SFMLOC_OK;
  SFM_HIP(hipMalloc((void **)p, n * sizeof(T)));
  *acct += n * sizeof(T);
  return SFMLOC_OK;
}

template <typename T>
int dev_upload(uint64_t *acct, T **p, const T *h, size_t n, hipStream_t s) {
  int rc = dev_alloc(acct, p, n);
  if (rc) return rc;
  if (n) SFM_HIP(hipMemcpyAsync(*p, h, n * sizeof(T), hipMemcpyHostToDevice, s));
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
  Ctx *c;
  int which;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  EventScope(Ctx *c_, int which_)
      : c(c_), which(which_),
        on(c_->map->params.profile == 1 || (c_->map->params.profile == 2 && which_ == SFMLOC_K_HAMMING)) {
    if (!on) return;
    if (!c->event_pool.empty()) {
      a = c->event_pool.back().first;
      b = c->event_pool.back().second;
      c->event_pool.pop_back();
    } else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      on = false;
      return;
    }
    hipEventRecord(a, c->stream);
  }
  ~EventScope() {
    if (!on) return;
    hipEventRecord(b, c->stream);
    c->pending_events.push_back({which, {a, b}});
  }
};

// per_kind_ms (optional, [SFMLOC_K_COUNT]): the drained brackets' time by kind, for the caller's own bookkeeping
int drain_events(Ctx *c, double *per_kind_ms = nullptr) {
  for (auto &pe : c->pending_events) {
    float ms = 0.f;
    SFM_HIP(hipEventSynchronize(pe.second.second));
    SFM_HIP(hipEventElapsedTime(&ms, pe.second.first, pe.second.second));
    c->stats.total_ms[pe.first] += ms;
    if (per_kind_ms) per_kind_ms[pe.first] += ms;
    c->stats.launches[pe.first] += 1;
    c->event_pool.push_back(pe.second);
  }
  c->pending_events.clear();
  return SFMLOC_OK;
}

void free_ctx(Ctx *c) {
  if (!c) return;
  if (c->borrowers > 0) {  // its stream is lent out: the last borrower to go frees it (sfmloc_context_create_sharing)
    c->zombie = true;
    return;
  }
  if (c->stream.own) hipStreamSynchronize(c->stream.own);
  for (auto &pe : c->pending_events) {
    hipEventDestroy(pe.second.first);
    hipEventDestroy(pe.second.second);
  }
  for (auto &e : c->event_pool) {
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  void *ptrs[] = {c->d_part,      c->d_view_sel,  c->d_view_widx0, c->d_block_list, c->d_view_count, c->d_match_i,
                  c->d_match_key, c->d_geo_count, c->d_geo_idx,    c->d_result,     c->d_cand_part,
                  c->d_best64,    c->d_winner,    c->d_ms_n,       c->d_ms_qfeat,   c->d_ms_landmark, c->d_pt2d,
                  c->d_pt3d,      c->d_xn,        c->d_logc_n,     c->d_logc_k,     c->d_vec_index,  c->d_best_inl,
                  c->d_hyp_nfa,   c->d_hyp_err,   c->d_hyp_model,  c->d_hyp_k,      c->d_hyp_inl,
                  c->d_inlier_idx, c->d_prep_models, c->d_prep_nm,
                  c->d_bow_query, c->d_bow_dist,  c->d_bow_cand,   c->d_bow_sel,    c->d_flagged,    c->d_n_flagged,  c->d_k1_counters, c->d_flagmask, c->d_rows_scratch, c->d_rows_arrivals, c->d_flagged_desc,
                  c->d_geo_model, c->d_geo_j,     c->d_guided_row, c->d_geo_dist,
                  c->fl_key, c->fl_idx, c->fl_count, c->fl_list, c->d_k3_spec, c->d_k3_arrive, c->d_k3_static, c->fl_vec_index, c->fl_best_inl, c->fl_logc_n, c->fl_logc_k,
                  c->d_pair_qfeat_big, c->d_pair_landmark_big, c->d_p3p_ws_key, c->d_p3p_ws_idx, c->d_p3p_terms};
  for (void *p : ptrs)
    if (p) hipFree(p);
  if (c->h_pinned) hipHostFree(c->h_pinned);
  if (c->h_result) hipHostFree(c->h_result);
  if (c->pinned_busy) hipEventDestroy(c->pinned_busy);
  if (c->xev_out) hipEventDestroy(c->xev_out);
  if (c->xev_in) hipEventDestroy(c->xev_in);
  gang_member_free(c);
  if (c->stream.own && !c->stream_borrowed) hipStreamDestroy(c->stream.own);
  Ctx *lender = c->lender;
  delete c;
  if (lender && --lender->borrowers == 0 && lender->zombie) free_ctx(lender);
}

int make_ctx(Map *m, Ctx **out, Ctx *share = nullptr, bool merge_only = false) {
  *out = nullptr;
  Ctx *c = new (std::nothrow) Ctx();
  SFM_CHECK(c, SFMLOC_ENOMEM, "out of host memory");
  c->map = m;
  int rc = SFMLOC_OK;
#define CTX_TRY(x)   \
  do {               \
    rc = (x);        \
    if (rc) {        \
      free_ctx(c);   \
      return rc;     \
    }                \
  } while (0)
#define CTX_HIP(x)                                              \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      set_error("%s -> %s", #x, hipGetErrorString(e_));         \
      free_ctx(c);                                              \
      return e_ == hipErrorOutOfMemory ? SFMLOC_ENOMEM : SFMLOC_EHIP; \
    }                                                           \
  } while (0)
  if (share) {  // no stream (hardware queue) of its own: work is queued on the lender's
    c->stream.own = share->stream.own;
    c->stream_borrowed = true;
    c->lender = share;
    ++share->borrowers;
  } else {
    CTX_HIP(hipStreamCreateWithFlags(&c->stream.own, hipStreamNonBlocking));
  }
  const uint64_t n_pad = (uint64_t)m->n_blocks * kBlockRows;
  uint64_t *acct = &c->hbm_bytes;
  c->merge_only = merge_only;
  // a context that only ever merges candidate parts and runs P3P (sfmloc_context_create_merge) has no use for the
  // per-bank-row arrays of the matching stages: 30 B per row and the 32 MB of the flagged-row pass
  const bool full = !merge_only;
  if (full) CTX_TRY(dev_alloc(acct, &c->d_part, (size_t)n_pad));
  CTX_TRY(dev_alloc(acct, &c->d_view_sel, (size_t)m->n_views + 1));
  CTX_TRY(dev_alloc(acct, &c->d_view_widx0, (size_t)m->n_views + 1));
  if (full) CTX_TRY(dev_alloc(acct, &c->d_block_list, (size_t)m->n_blocks));
  CTX_TRY(dev_alloc(acct, &c->d_view_count, (size_t)m->n_views + 1));  // + the phantom view (sfmloc_internal.h)
  if (full) CTX_TRY(dev_alloc(acct, &c->d_match_i, (size_t)m->n_rows));
  if (full) CTX_TRY(dev_alloc(acct, &c->d_match_key, (size_t)m->n_rows));
  if (full) CTX_TRY(dev_alloc(acct, &c->d_flagged, (size_t)n_pad));
  CTX_TRY(dev_alloc(acct, &c->d_n_flagged, (size_t)1));
  if (full) CTX_TRY(dev_alloc(acct, &c->d_flagmask, (size_t)m->n_blocks + 1));
  c->rows_chunk_cap = m->n_blocks < 4096 ? (m->n_blocks ? m->n_blocks : 1) : 4096;  // 4096 chunks = 262 k flagged rows, 16 MB
  if (full) {
    CTX_TRY(dev_alloc(acct, &c->d_rows_scratch, (size_t)c->rows_chunk_cap * 8 * 64));
    CTX_TRY(dev_alloc(acct, &c->d_rows_arrivals, (size_t)c->rows_chunk_cap));
    CTX_TRY(dev_alloc(acct, &c->d_flagged_desc, (size_t)c->rows_chunk_cap * 4 * 64));
    CTX_HIP(hipMemset(c->d_rows_arrivals, 0, (size_t)c->rows_chunk_cap * sizeof(uint32_t)));
  }
  CTX_TRY(dev_alloc(acct, &c->d_k1_counters, (size_t)2 * kK1CounterSlots));
  CTX_HIP(hipMemset(c->d_k1_counters, 0, 2 * kK1CounterSlots * sizeof(unsigned long long)));
  CTX_TRY(dev_alloc(acct, &c->d_geo_count, (size_t)m->n_views + 1));
  if (full) CTX_TRY(dev_alloc(acct, &c->d_geo_idx, (size_t)m->n_rows));
  CTX_TRY(dev_alloc(acct, &c->d_geo_model, ((size_t)m->n_views + 1) * 10));
  // everything a finished query reports lives in ONE device record laid out as HostResult: one D2H copy per query
  CTX_TRY(dev_alloc(acct, &c->d_result, sizeof(HostResult)));
  {
    HostResult *r = reinterpret_cast<HostResult *>(c->d_result);
    c->d_p3p_state = &r->state;
    c->d_pose = &r->pose;
    c->d_status = &r->status;
    c->d_view_stats = r->view_stats;
    c->d_pair_qfeat = r->pair_qfeat;
    c->d_pair_landmark = r->pair_landmark;
  }
  CTX_TRY(dev_alloc(acct, &c->d_cand_part, (size_t)kPartHeaderBytes + (size_t)c->cand_cap * sizeof(Candidate)));
  if (full) CTX_TRY(dev_alloc(acct, &c->d_geo_dist, (size_t)m->n_rows));
  CTX_TRY(dev_alloc(acct, &c->d_best64, (size_t)65536));
  CTX_TRY(dev_alloc(acct, &c->d_winner, (size_t)65536));
  CTX_TRY(dev_alloc(acct, &c->d_ms_n, (size_t)1));
  CTX_TRY(dev_alloc(acct, &c->d_ms_qfeat, (size_t)65536));
  CTX_TRY(dev_alloc(acct, &c->d_ms_landmark, (size_t)65536));
  CTX_TRY(dev_alloc(acct, &c->d_pt2d, (size_t)65536 * 2));
  CTX_TRY(dev_alloc(acct, &c->d_pt3d, (size_t)65536 * 3));
  CTX_TRY(dev_alloc(acct, &c->d_xn, (size_t)kP3pMaxN * 2));
  CTX_TRY(dev_alloc(acct, &c->d_logc_n, (size_t)kP3pMaxN + 1));
  CTX_TRY(dev_alloc(acct, &c->d_logc_k, (size_t)kP3pMaxN + 1));
  CTX_TRY(dev_alloc(acct, &c->d_vec_index, (size_t)kP3pMaxN));
  CTX_TRY(dev_alloc(acct, &c->d_best_inl, (size_t)kP3pMaxN));
  CTX_TRY(dev_alloc(acct, &c->d_hyp_nfa, (size_t)kP3pSlots));
  CTX_TRY(dev_alloc(acct, &c->d_hyp_err, (size_t)kP3pSlots));
  CTX_TRY(dev_alloc(acct, &c->d_hyp_model, (size_t)kP3pSlots * 12));
  CTX_TRY(dev_alloc(acct, &c->d_hyp_k, (size_t)kP3pSlots));
  CTX_TRY(dev_alloc(acct, &c->d_prep_models, (size_t)kP3pBatchMax * 48));
  CTX_TRY(dev_alloc(acct, &c->d_prep_nm, (size_t)kP3pBatchMax));
  CTX_TRY(dev_alloc(acct, &c->d_hyp_inl, (size_t)kP3pSlots * kP3pMaxN));
  CTX_TRY(dev_alloc(acct, &c->d_inlier_idx, (size_t)kP3pMaxN));
  // the regrowable P3P set as allocated above (ctx_p3p_reserve replaces it and keeps the accounts)
  c->p3p_bytes = (uint64_t)kP3pMaxN * (2 * sizeof(double) + 3 * sizeof(int32_t)) + 2 * ((uint64_t)kP3pMaxN + 1) * sizeof(float) +
                 (uint64_t)kP3pSlots * kP3pMaxN * sizeof(int32_t);
  if (m->bow_dim) {
    CTX_TRY(dev_alloc(acct, &c->d_bow_query, (size_t)m->bow_dim));
    CTX_TRY(dev_alloc(acct, &c->d_bow_dist, (size_t)m->n_views));
    CTX_TRY(dev_alloc(acct, &c->d_bow_cand, (size_t)m->n_views));
    CTX_TRY(dev_alloc(acct, &c->d_bow_sel, (size_t)m->n_views + 1));
  }
  CTX_HIP(hipHostMalloc((void **)&c->h_pinned, ((size_t)2 * m->n_views + m->n_blocks + 16) * sizeof(uint32_t),
                        hipHostMallocDefault));
  CTX_HIP(hipHostMalloc(&c->h_result, sizeof(HostResult), hipHostMallocDefault));
  memset(c->h_result, 0, sizeof(HostResult));
  CTX_HIP(hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
  CTX_HIP(hipMemsetAsync(c->d_view_count, 0, ((size_t)m->n_views + 1) * sizeof(uint32_t), c->stream));
  CTX_HIP(hipMemsetAsync(c->d_geo_count, 0, ((size_t)m->n_views + 1) * sizeof(uint32_t), c->stream));
  CTX_HIP(hipMemsetAsync(c->d_ms_n, 0, sizeof(uint32_t), c->stream));
  CTX_HIP(hipEventCreateWithFlags(&c->xev_out, hipEventDisableTiming));
  CTX_HIP(hipEventCreateWithFlags(&c->xev_in, hipEventDisableTiming));
  CTX_HIP(hipMemsetAsync(c->d_pose, 0, sizeof(Pose), c->stream));
  CTX_HIP(hipMemsetAsync(c->d_p3p_state, 0, sizeof(P3pState), c->stream));
  CTX_HIP(hipMemsetAsync(c->d_view_stats, 0, 3 * sizeof(uint32_t), c->stream));
  CTX_HIP(hipStreamSynchronize(c->stream));
  // (the two hipMemset above -- the arrival counters of k_hamming_rows, K1's statistics -- run on the NULL stream, which the
  // context's non-blocking stream is not ordered with, and hipMemset of device memory may return before it has run: a first
  // scan that overtook the memset found garbage arrival counters, merged nothing and left the head pass's seeds to K2 --
  // the one wrong first scan of round 4, profiles/r04_pytest_gpu_failed_run_excerpt.txt)
  CTX_HIP(hipStreamSynchronize(nullptr));
#undef CTX_TRY
#undef CTX_HIP
  *out = c;
  return SFMLOC_OK;
}

void free_map(Map *m) {
  if (!m) return;
  hipSetDevice(m->device);
  for (Ctx *c : m->pool) free_ctx(c);
  free_ctx(m->ctx0);
  void *ptrs[] = {m->d_bank,         m->d_view_off,   m->d_view_id, m->d_view_wh, m->d_kpt, m->d_row_landmark,
                  m->d_landmark_id,  m->d_landmark_X, m->d_bow,     m->d_L10,     m->d_ratio_cnt};
  for (void *p : ptrs)
    if (p) hipFree(p);
  delete m;
}

// ----- stages on a context --------------------------------------------------------------------------

// Every counter / flag the stages of one query start from, cleared by ONE launch instead of eight memsets (each a
// separate ~5 us dispatch on the query's critical path).  The stage functions keep their own memsets for callers
// that drive them one at a time (Ctx::cleared says which applies).
struct QueryResetBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(QueryResetArgs R) {
    query_reset_items(R, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);  // chain_device.h
  }
};
__global__ __launch_bounds__(256) void k_query_reset(QueryResetArgs R) {
  QueryResetBody::run(R);
}

QueryResetArgs make_reset_args(Ctx *c, const Query *q) {
  QueryResetArgs R;
  R.view_count = c->d_view_count;
  R.geo_count = c->d_geo_count;
  R.n_views = c->map->n_views;
  R.best64 = c->d_best64;
  R.nq = q->n ? q->n : 1;
  R.n_flagged = c->d_n_flagged;
  R.status = c->d_status;
  R.cand_header = reinterpret_cast<uint32_t *>(c->d_cand_part);
  R.view_stats = c->d_view_stats;
  R.ms_n = c->d_ms_n;
  R.fl_count = c->fl_count;
  return R;
}

int ctx_reset_for_query(Ctx *c, const Query *q) {
  Map *m = c->map;
  if (c->chain_done) {  // the shortlist's workgroup already did it (chain_after_shortlist)
    c->cleared = true;
    return SFMLOC_OK;
  }
  const uint32_t nq = q->n ? q->n : 1;
  const uint32_t n = m->n_views + 1 > nq ? m->n_views + 1 : nq;
  sfm_launch<QueryResetBody>(c, k_query_reset, dim3((n + 255) / 256), dim3(256), 0, make_reset_args(c, q));
  SFM_HIP(hipGetLastError());
  c->cleared = true;
  return SFMLOC_OK;
}

// The chain a shortlist kernel runs after the shortlist (reset + block list of the knn selected views in d_sel)
ChainArgs make_chain_args(Ctx *c, const Query *q, const uint32_t *d_sel, uint32_t n_sel) {
  Map *m = c->map;
  ChainArgs C{};  // (keys_out = null: no key list)
  C.enabled = 1;
  C.reset = make_reset_args(c, q);
  C.blocks.sel = d_sel;
  C.blocks.n_sel = n_sel;
  C.blocks.view_off = m->d_view_off;
  C.blocks.view_sel_out = c->d_view_sel;
  C.blocks.widx0 = c->d_view_widx0;
  C.blocks.block_list = c->d_block_list;
  C.blocks.bound = (uint32_t)std::min<uint64_t>(m->n_blocks, (uint64_t)n_sel * m->max_view_blocks);
  C.blocks.flagmask = c->d_flagmask;
  return C;
}

struct ClearedScope {  // the flag must not outlive the call that set it
  Ctx *c;
  ~ClearedScope() { c->cleared = false; }
};

// A short scan is cut into query slices (more waves, better balance, ~7 % more arithmetic) only when the GPU has
// nothing else to fill it with, i.e. when no other context of the map has work queued.
static void ctx_mark_busy(Ctx *c) {
  Map *m = c->map;
  c->k1_may_slice = m->busy_ctx.load(std::memory_order_relaxed) - (c->counted_busy ? 1 : 0) <= 0;
  if (!c->counted_busy) {
    c->counted_busy = true;
    m->busy_ctx.fetch_add(1, std::memory_order_relaxed);
  }
}
static void ctx_mark_idle(Ctx *c) {
  if (c->counted_busy) {
    c->counted_busy = false;
    c->map->busy_ctx.fetch_sub(1, std::memory_order_relaxed);
  }
}

int ctx_match_putative(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel, const uint32_t *d_sel = nullptr);
int ctx_localize_begin(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel, const uint32_t *d_sel = nullptr);

// d_sel: a selection that lives on the device (ascending view indices, n_sel of them; the BoW shortlist of
// sfmloc_localize_bow_begin) -- the block list is then built by a kernel and the host never sees the views
int ctx_match_putative(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel, const uint32_t *d_sel) {
  Map *m = c->map;
  SFM_CHECK(!c->merge_only, SFMLOC_EINVAL, "this context was created for sfmloc_merge_begin only (sfmloc_context_create_merge)");
  const bool all_views = (view_sel == nullptr && d_sel == nullptr);
  if (all_views) n_sel = m->n_views;
  SFM_CHECK(n_sel <= m->n_views, SFMLOC_EINVAL, "view selection: n_sel %u > n_views %u", n_sel, m->n_views);

  // selected views -> ascending list of the 64-row bank blocks they overlap, plus per selected view the
  // position of its first block in that list
  uint32_t n_work_blocks = m->n_blocks;
  c->last_blocks.clear();
  c->last_blocks_on_device = false;
  c->flagmask_zeroed = false;
  c->merge_is_deferred = false;  // (a merge left to a K3 that never ran belongs to an abandoned query)
  if (d_sel) {
    // launch bound the host can know: every selected view overlaps at most max_view_blocks blocks
    const uint64_t bound = std::min<uint64_t>(m->n_blocks, (uint64_t)n_sel * m->max_view_blocks);
    n_work_blocks = (uint32_t)bound;
    if (n_sel && c->chain_done) {
      c->flagmask_zeroed = true;  // (built by the shortlist's workgroup, chain_after_shortlist)
    } else if (n_sel) {
      int rc = launch_blocks_from_views(c, d_sel, n_sel, n_work_blocks);
      if (rc) return rc;
    }
    c->last_blocks_on_device = true;
  } else if (!all_views) {
    if (c->pinned_busy) SFM_HIP(hipEventSynchronize(c->pinned_busy));  // previous upload still reading the staging
    uint32_t *h_sel = c->h_pinned;
    uint32_t *h_w0 = c->h_pinned + m->n_views;
    uint32_t *h_blk = c->h_pinned + 2 * (size_t)m->n_views;
    uint32_t nb = 0;
    for (uint32_t k = 0; k < n_sel; ++k) {
      const uint32_t v = view_sel[k];
      SFM_CHECK(v < m->n_views, SFMLOC_EINVAL, "view selection: index %u out of range", v);
      SFM_CHECK(k == 0 || view_sel[k - 1] < v, SFMLOC_EINVAL, "view selection must be strictly ascending");
      h_sel[k] = v;
      h_w0[k] = 0;
      const uint32_t r0 = m->h_view_off[v], r1 = m->h_view_off[v + 1];
      if (r1 == r0) continue;
      const uint32_t b0 = r0 / kBlockRows, b1 = (r1 - 1) / kBlockRows;
      uint32_t b = b0;
      if (nb && h_blk[nb - 1] == b0) {  // the previous selected view ends in this view's first block
        h_w0[k] = nb - 1;
        b = b0 + 1;
      } else {
        h_w0[k] = nb;
      }
      for (; b <= b1; ++b) h_blk[nb++] = b;
    }
    n_work_blocks = nb;
    c->last_blocks.assign(h_blk, h_blk + nb);
    if (n_sel) {
      SFM_HIP(hipMemcpyAsync(c->d_view_sel, h_sel, n_sel * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      SFM_HIP(hipMemcpyAsync(c->d_view_widx0, h_w0, n_sel * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    if (nb) SFM_HIP(hipMemcpyAsync(c->d_block_list, h_blk, nb * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    if (!c->pinned_busy) SFM_HIP(hipEventCreateWithFlags(&c->pinned_busy, hipEventDisableTiming));
    SFM_HIP(hipEventRecord(c->pinned_busy, c->stream));
  }

  // query split: spread a short block list over the chip (partial top-2 are merged in K2); the partial
  // buffer holds n_blocks wave-blocks in all
  uint32_t split = 1;
  if (n_work_blocks && q->n >= 256) {
    const uint64_t want = (uint64_t)m->n_cu * 8;
    while (split < c->max_split && (uint64_t)n_work_blocks * split < want && q->n / (split * 2) >= 128 &&
           (uint64_t)n_work_blocks * split * 2 <= m->n_blocks)
      split *= 2;
  }

  if (!c->cleared) SFM_HIP(hipMemsetAsync(c->d_view_count, 0, ((size_t)m->n_views + 1) * sizeof(uint32_t), c->stream));
  c->last_split = split;
  c->last_nq = q->n;
  c->last_n_sel = n_sel;
  c->last_all_views = all_views;
  c->last_n_work_blocks = n_work_blocks;
  c->last_query = q;
  if (q->n == 0 || n_sel == 0) return SFMLOC_OK;

  int rc;
  {
    EventScope ev(c, SFMLOC_K_HAMMING);
    rc = launch_hamming_top2(c, q, n_work_blocks, !all_views, split);
  }
  if (rc) return rc;
  c->stats.hamming_pairs += (uint64_t)n_work_blocks * kBlockRows * q->n;
  c->stats.hamming_alg_bytes += (uint64_t)n_work_blocks * kBlockRows * 64 + (uint64_t)q->n * 64;
  {
    EventScope ev(c, SFMLOC_K_COMPACT);
    rc = launch_merge_ratio_compact(c, q, n_sel, all_views, split, n_work_blocks);
  }
  return rc;
}

int check_stage(Ctx *c, Query *q, const char *who) {
  SFM_CHECK(c && q, SFMLOC_EINVAL, "%s: null argument", who);
  Map *m = c->map;
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "%s: query belongs to another map", who);
  SFM_CHECK(m->have_geometry, SFMLOC_EINVAL,
            "%s: the map was created without view sizes / keypoints / landmarks / intrinsics", who);
  SFM_CHECK(q->n == 0 || q->d_kpt, SFMLOC_EINVAL, "%s: the query was created without keypoints", who);
  SFM_CHECK(q->n == 0 || (q->width > 0 && q->height > 0), SFMLOC_EINVAL, "%s: query image size missing", who);
  SFM_CHECK(c->last_query == q, SFMLOC_EINVAL, "%s: run sfmloc_match_putative with this query first", who);
  return SFMLOC_OK;
}

int ctx_geometric_filter(Ctx *c, Query *q) {
  Map *m = c->map;
  if (!c->cleared) {
    SFM_HIP(hipMemsetAsync(c->d_geo_count, 0, ((size_t)m->n_views + 1) * sizeof(uint32_t), c->stream));
    SFM_HIP(hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
  }
  if (q->n == 0 || c->last_n_sel == 0) return SFMLOC_OK;
  EventScope ev(c, SFMLOC_K_FMATRIX);
  int rc = launch_fmatrix_filter(c, q, c->last_n_sel, c->last_all_views);
  if (rc) return rc;
  // -gm (MatchUtils.cpp:413-415): the surviving views' matches are re-derived under the estimated F
  if (m->params.guided_matching) rc = launch_guided_matching(c, q, c->last_n_sel, c->last_all_views);
  return rc;
}

int ctx_match_set(Ctx *c, Query *q) {
  EventScope ev(c, SFMLOC_K_MATCHSET);
  return launch_match_set(c, q, c->last_n_sel, c->last_all_views);
}

// Enqueues P3P AC-RANSAC rounds.  The device-side state machine turns surplus rounds into no-ops, so a fixed
// number is enqueued without synchronising.
int ctx_resection_enqueue(Ctx *c, bool first_call) {
  int rc = SFMLOC_OK;
  if (first_call) {
    rc = launch_p3p_init(c);
    if (rc) return rc;
    // the form of this query's rounds (acransac.hip, at P3pShared: the small form holds sets of at most 512
    // correspondences in a fraction of the registers): a prediction from the map's last queries -- or a certainty, when
    // the query has no more features than that --; ctx_resection_wait corrects it if the set turns out larger
    static const int env_small = [] { const char *e = getenv("SFMLOC_P3P_SMALL"); return e ? atoi(e) : 1; }();
    c->p3p_small = env_small == 2 || (env_small == 1 && (c->p3p_query_n <= 512 ||
                                                         c->map->p3p_small_credit.load(std::memory_order_relaxed) >= 8));
    // The sequential form (acransac.hip, k_p3p_seq: the whole AC-RANSAC as ONE launch of one workgroup, no speculative
    // hypotheses, no rounds) -- built, bit-exact, measured and NOT the default: one workgroup walks a headline query's 410
    // iterations in 2.2-2.5 ms where the rounds take 0.25 (profiles/r04_k5_sequential_form.txt).  SFMLOC_P3P_SEQ: 0 never
    // (default), 1 while the GPU is shared, 2 always.
    static const int env_seq = [] { const char *e = getenv("SFMLOC_P3P_SEQ"); return e ? atoi(e) : 0; }();
    c->p3p_seq = env_seq == 2 || (env_seq == 1 && (!c->k1_may_slice || c->stream.gang != nullptr));
    if (c->p3p_seq) {
      rc = launch_p3p_seq(c);
      if (rc == SFMLOC_OK) rc = launch_p3p_finish(c);
      return rc;
    }
  }
  // typically 6-8 rounds end the stage (one per improvement of the model); rounds enqueued past the end return at
  // once but still cost two launches each, so the first call queues 9 and ctx_resection_wait adds more if needed
  static const int env_batch = [] {  // tuning hooks: hypotheses per later round / rounds queued by the first call
    const char *e = getenv("SFMLOC_P3P_BATCH");
    const int v = e ? atoi(e) : 0;
    // 256 per later round: as many rounds as with 512 (an improvement of the model comes early in a round or not at
    // all), 3-5 % more queries per second because fewer speculative hypotheses are evaluated for nothing
    return (v >= 16 && v <= kP3pBatchMax) ? v : 256;
  }();
  // the first round's size while the GPU is shared (a query alone: 64): after the geometric filter the first hypothesis is
  // nearly always the one that switches sampling to its inliers, and everything behind it is thrown away
  static const int env_first_shared = [] {
    const char *e = getenv("SFMLOC_P3P_FIRST_BATCH");
    const int v = e ? atoi(e) : 0;
    return (v >= 4 && v <= 256) ? v : 64;
  }();
  static const int env_rounds = [] {
    const char *e = getenv("SFMLOC_P3P_ROUNDS");
    const int v = e ? atoi(e) : 0;
    return (v >= 1 && v <= 64) ? v : 9;
  }();
  // (in a gang session a surplus round costs a fraction of a launch, and a member that needs more than were queued gets
  // them alone on the gang's stream: three more up front)
  const int rounds = first_call ? (c->stream.gang ? std::max(env_rounds, 12) : env_rounds) : 6;
  for (int r = 0; r < rounds && rc == SFMLOC_OK; ++r)
    rc = launch_p3p_round(c, (first_call && r == 0) ? (c->k1_may_slice ? 64 : env_first_shared) : env_batch);
  if (rc == SFMLOC_OK) rc = launch_p3p_finish(c);  // pose + inlier pairs once the state says "done"; a no-op before
  return rc;
}

}  // namespace

// Grows the P3P workspace of a context to hold `n` correspondences (a query has at most one per feature).  Rare: only a
// query with more than kP3pMaxN features gets here.  The new set is allocated first and swapped in only when every
// allocation has succeeded: on failure the context keeps its old arrays and capacity and the caller gets SFMLOC_ENOMEM
// (the old code freed first and left null pointers behind a stale p3p_cap).  The stream is drained before
// the old arrays are freed (kernels of the previous query may still read them); a context that is recording for a gang
// session never gets here with work recorded -- the reserve is the first thing a query's stage does -- but if it does,
// reading c->stream issues what was recorded first (gang.h), so stream order is still call order.
namespace {
// test hook (sfmloc_debug_fail_p3p_alloc(k)): the k-th allocation of the NEXT regrowth fails; one shot, consumed atomically
// by the regrowth that sees it (no environment variable: a stray one must not be able to fail a production allocation)
std::atomic<int> g_test_fail_alloc{-1};
}
int ctx_p3p_reserve(Ctx *c, uint32_t n) {
  c->p3p_query_n = n;
  if (n <= c->p3p_cap) return SFMLOC_OK;
  uint32_t cap = c->p3p_cap;
  while (cap < n) cap <<= 1;
  // hypothesis inlier lists: kP3pBatchMax lists of kP3pMaxN, or (more correspondences) kP3pLargeBatch lists of cap
  const size_t large_batch = 64;  // acransac.hip kP3pLargeBatch
  const size_t hyp = std::max<size_t>((size_t)kP3pSlots * kP3pMaxN, large_batch * cap);
  const size_t bytes[12] = {(size_t)cap * 2 * sizeof(double),      ((size_t)cap + 1) * sizeof(float),
                            ((size_t)cap + 1) * sizeof(float),     (size_t)cap * sizeof(int32_t),
                            (size_t)cap * sizeof(int32_t),         hyp * sizeof(int32_t),
                            (size_t)cap * sizeof(uint32_t),        (size_t)cap * sizeof(uint32_t),
                            (size_t)cap * sizeof(uint32_t),        large_batch * cap * sizeof(uint64_t),
                            large_batch * cap * sizeof(uint32_t),  ((size_t)cap / 2 + 2) * sizeof(double)};
  const int test_fail_alloc = g_test_fail_alloc.exchange(-1, std::memory_order_relaxed);
  void *fresh[12] = {nullptr};
  uint64_t total = 0;
  for (int i = 0; i < 12; ++i) {
    hipError_t err = (i == test_fail_alloc) ? hipErrorOutOfMemory : hipMalloc(&fresh[i], bytes[i]);
    if (err != hipSuccess) {
      if (i != test_fail_alloc) (void)hipGetLastError();
      for (int k = 0; k < i; ++k) (void)hipFree(fresh[k]);
      set_error("P3P workspace for %u correspondences: allocation %d of 12 (%zu bytes) failed: %s", cap, i, bytes[i],
                hipGetErrorString(err));
      return SFMLOC_ENOMEM;
    }
    total += bytes[i];
  }
  SFM_HIP(hipStreamSynchronize(c->stream));
  void *old[12] = {c->d_xn,        c->d_logc_n,        c->d_logc_k,           c->d_vec_index,  c->d_best_inl,   c->d_hyp_inl,
                   c->d_inlier_idx, c->d_pair_qfeat_big, c->d_pair_landmark_big, c->d_p3p_ws_key, c->d_p3p_ws_idx, c->d_p3p_terms};
  for (void *p : old)
    if (p) (void)hipFree(p);
  c->d_xn = (double *)fresh[0];
  c->d_logc_n = (float *)fresh[1];
  c->d_logc_k = (float *)fresh[2];
  c->d_vec_index = (int32_t *)fresh[3];
  c->d_best_inl = (int32_t *)fresh[4];
  c->d_hyp_inl = (int32_t *)fresh[5];
  c->d_inlier_idx = (uint32_t *)fresh[6];
  c->d_pair_qfeat_big = (uint32_t *)fresh[7];
  c->d_pair_landmark_big = (uint32_t *)fresh[8];
  c->d_p3p_ws_key = (uint64_t *)fresh[9];
  c->d_p3p_ws_idx = (uint32_t *)fresh[10];
  c->d_p3p_terms = (double *)fresh[11];
  c->d_pair_qfeat = c->d_pair_qfeat_big;  // no longer inside the HostResult record
  c->d_pair_landmark = c->d_pair_landmark_big;
  // accounting: the regrown set replaces the previous one (the first set was counted by make_ctx through dev_alloc)
  const uint64_t before = c->p3p_bytes;
  c->p3p_bytes = total;
  c->hbm_bytes += total - before;
  if (c->map) c->map->hbm_bytes += total - before;
  c->p3p_cap = cap;
  return SFMLOC_OK;
}

namespace {

// The host copy of the result record is written by k_p3p_finish itself (p3p_publish), which every call of
// ctx_resection_enqueue ends with; a copy is only needed when the chain stops before K5 (measurements).
int ctx_fetch_result(Ctx *c, bool no_p3p = false) {
  if (!no_p3p) return SFMLOC_OK;
  HostResult *h = reinterpret_cast<HostResult *>(c->h_result);
  SFM_HIP(hipMemcpyAsync(h, c->d_result, sizeof(HostResult), hipMemcpyDeviceToHost, c->stream));
  return SFMLOC_OK;
}

int ctx_resection_wait(Ctx *c) {
  HostResult *h = reinterpret_cast<HostResult *>(c->h_result);
  for (int guard = 0; guard < 64; ++guard) {
    SFM_HIP(hipStreamSynchronize(c->stream));
    c->stream.dirty = false;
    if (h->state.done) {
      // what the next queries' rounds look like (launch_p3p_round): this one's number of correspondences
      if (h->state.n > 512) c->map->p3p_wide_credit.store(64, std::memory_order_relaxed);
      else if (c->map->p3p_wide_credit.load(std::memory_order_relaxed) > 0) c->map->p3p_wide_credit.fetch_sub(1, std::memory_order_relaxed);
      if (h->state.n > 512) c->map->p3p_small_credit.store(0, std::memory_order_relaxed);
      else if (c->map->p3p_small_credit.load(std::memory_order_relaxed) < 64) c->map->p3p_small_credit.fetch_add(1, std::memory_order_relaxed);
      return SFMLOC_OK;
    }
    // (the small rounds of a set that turned out larger than the form holds all returned at once: the full form now)
    if (c->p3p_small && h->state.n > 512) c->p3p_small = false;
    // (a set the sequential launch is not built for -- more correspondences than its waves hold -- came back untouched)
    c->p3p_seq = false;
    int rc;
    {
      EventScope ev(c, SFMLOC_K_P3P);
      rc = ctx_resection_enqueue(c, false);
    }
    if (rc) return rc;
    rc = ctx_fetch_result(c);
    if (rc) return rc;
  }
  set_error("P3P AC-RANSAC did not finish (iteration %d of %d)", h->state.iter, h->state.n_iter);
  return SFMLOC_EHIP;
}

static int ctx_localize_begin_impl(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel, const uint32_t *d_sel);

int ctx_localize_begin(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel, const uint32_t *d_sel) {
  SFM_CHECK(c->in_flight == nullptr, SFMLOC_EINVAL, "sfmloc_localize_begin: context already has a query in flight");
  ctx_mark_busy(c);
  const int rc = ctx_localize_begin_impl(c, q, view_sel, n_sel, d_sel);
  if (rc) ctx_mark_idle(c);
  return rc;
}

// SFMLOC_DIAG_STOP_AFTER = 1 (putative matches) / 2 (geometric filter) / 3 (2D-3D set): measurements only -- the chain
// ends there and sfmloc_localize_end returns an empty pose (bench.py marks such a line as not the metric)
static int diag_stop_after() {
  static const int v = [] {
    const char *e = getenv("SFMLOC_DIAG_STOP_AFTER");
    return e ? atoi(e) : 0;
  }();
  return v;
}

static int ctx_localize_begin_impl(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel, const uint32_t *d_sel) {
  c->t_begin = now_s();
  ClearedScope cs{c};
  int rc = ctx_p3p_reserve(c, q->n);
  if (rc) return rc;
  rc = ctx_reset_for_query(c, q);
  if (rc) return rc;
  const int stop = diag_stop_after();
  c->defer_merge = stop != 1;  // K3 follows on this context: its per-view workgroups build their own match lists
  rc = ctx_match_putative(c, q, view_sel, n_sel, d_sel);
  c->defer_merge = false;
  if (rc) return rc;
  rc = check_stage(c, q, "sfmloc_localize");
  if (rc) return rc;
  if (stop != 1) {
    rc = ctx_geometric_filter(c, q);
    if (rc) return rc;
  }
  if (stop == 0 || stop > 2) {
    rc = ctx_match_set(c, q);
    if (rc) return rc;
  }
  if (stop == 0) {
    EventScope ev(c, SFMLOC_K_P3P);
    rc = ctx_resection_enqueue(c, true);
  }
  if (rc) return rc;
  rc = ctx_fetch_result(c, stop != 0);
  if (rc) return rc;
  c->in_flight = q;
  return SFMLOC_OK;
}

int ctx_localize_end(Ctx *c, sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark, uint32_t cap) {
  SFM_CHECK(c->in_flight != nullptr, SFMLOC_EINVAL, "sfmloc_localize_end: no query in flight on this context");
  c->in_flight = nullptr;
  ctx_mark_idle(c);
  if (diag_stop_after()) {  // truncated chain (measurements): nothing to wait for but the stream
    SFM_HIP(hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof(*out));
    return SFMLOC_OK;
  }
  int rc = ctx_resection_wait(c);
  if (rc) return rc;
  HostResult *h = reinterpret_cast<HostResult *>(c->h_result);
  *out = h->pose;
  out->status |= h->status;
  out->n_putative_views = (int32_t)h->view_stats[0];
  out->n_geometric_views = (int32_t)h->view_stats[1];
  // the form of the next queries' K3 (launch_geometric_filter): was this one's largest view one for the 1 024-match
  // register form?  (A query with a still larger view gains nothing from it: the launches run one after the other, and
  // the block-wide form of its largest view is what it waits for either way -- measured on the dense lab frames, +0.2 ms.)
  if (h->view_stats[2] > 1024u && h->view_stats[2] <= 2048u) c->map->k3_huge_credit.store(64, std::memory_order_relaxed);
  else if (c->map->k3_huge_credit.load(std::memory_order_relaxed) > 0) c->map->k3_huge_credit.fetch_sub(1, std::memory_order_relaxed);
  if (h->view_stats[2] > 1024u) c->map->k3_big_credit.store(0, std::memory_order_relaxed);
  else if (h->view_stats[2] > 512u) c->map->k3_big_credit.store(64, std::memory_order_relaxed);
  else if (c->map->k3_big_credit.load(std::memory_order_relaxed) > 0) c->map->k3_big_credit.fetch_sub(1, std::memory_order_relaxed);
  // (none of these can be reached with data the reference accepts: a view's matches beyond 65 536, an internal
  // inconsistency of the candidate part, more correspondences than the query has features)
  SFM_CHECK((out->status & 1) == 0, SFMLOC_ECAP, "a view has more than 65536 putative matches");
  SFM_CHECK((out->status & 2) == 0, SFMLOC_ECAP, "more than %u 2D-3D candidates (match-set workspace)", c->cand_cap);
  SFM_CHECK((out->status & 4) == 0, SFMLOC_ECAP, "more than %u 2D-3D correspondences (P3P workspace)", c->p3p_cap);
  if (out->ok && out->n_inliers > 0 && (pair_qfeat || pair_landmark)) {
    const uint32_t k = (uint32_t)out->n_inliers;
    SFM_CHECK(cap >= k, SFMLOC_ECAP, "pair buffers hold %u entries, %u inliers", cap, k);
    if (c->p3p_cap > (uint32_t)kP3pMaxN) {  // the pair lists live outside the HostResult record
      if (pair_qfeat) SFM_HIP(hipMemcpy(pair_qfeat, c->d_pair_qfeat, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
      if (pair_landmark) SFM_HIP(hipMemcpy(pair_landmark, c->d_pair_landmark, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    } else {
      if (pair_qfeat) memcpy(pair_qfeat, h->pair_qfeat, k * sizeof(uint32_t));
      if (pair_landmark) memcpy(pair_landmark, h->pair_landmark, k * sizeof(uint32_t));
    }
  }
  // the reference's `times` (LocalizeEngine.cc:643-658): with params.profile = 1 every stage of this query was
  // bracketed by HIP events on its stream -- selectBow = K8, putMatch = K1 + K2, geoMatch = K3, PnP = 2D-3D set + P3P
  // (the reference's PnP bucket runs from the end of geometricMatch to the end of Localize), others = the rest of the
  // wall time begin -> end.  Without profiling only the total is known and it goes to `others`.
  for (int i = 0; i < 7; ++i) out->stage_seconds[i] = 0.0;
  const double wall = now_s() - c->t_begin;
  if (c->map->params.profile == 1) {
    double per[SFMLOC_K_COUNT] = {0};
    rc = drain_events(c, per);
    if (rc) return rc;
    out->stage_seconds[1] = per[SFMLOC_K_BOW] * 1e-3;
    out->stage_seconds[3] = (per[SFMLOC_K_HAMMING] + per[SFMLOC_K_COMPACT]) * 1e-3;
    out->stage_seconds[4] = per[SFMLOC_K_FMATRIX] * 1e-3;
    out->stage_seconds[5] = (per[SFMLOC_K_MATCHSET] + per[SFMLOC_K_P3P]) * 1e-3;
    double sum = 0.0;
    for (int i = 0; i < 6; ++i) sum += out->stage_seconds[i];
    out->stage_seconds[6] = wall > sum ? wall - sum : 0.0;
  } else {
    out->stage_seconds[6] = wall;
  }
  return SFMLOC_OK;
}

}  // namespace

// gang.h: issue what the members of a gang session have recorded -- the heads of their lists that are the same kernel
// on the same grid as ONE launch, a head without a partner as a plain launch
int gang_flush(GangState *g) {
  int rc = SFMLOC_OK;
  bool oom = false;
  for (GangMember *m : g->members) oom |= m->gang_oom;
  if (oom) {  // a chain with a launch missing must not run at all: drop the session's records
    for (GangMember *m : g->members) {
      m->gang_recs.clear();
      m->gang_head = 0;
      m->gang_oom = false;
    }
    set_error("gang session: out of host memory while recording launches; the launches still recorded were dropped");
    return SFMLOC_ENOMEM;
  }
  for (;;) {
    // the members advance in step: always the earliest position any member still has to issue, and there every member
    // whose record is the same kernel on the same grid in ONE launch (dynamic LDS: the largest request serves all)
    size_t pos = SIZE_MAX;
    for (GangMember *m : g->members)
      if (m->gang_head < m->gang_recs.size() && m->gang_head < pos) pos = m->gang_head;
    if (pos == SIZE_MAX) break;
    GangRec *grp[kGangMembers];
    int n = 0;
    GangRec *lead = nullptr;
    uint32_t shmem = 0;
    for (GangMember *m : g->members) {
      if (m->gang_head != pos || pos >= m->gang_recs.size()) continue;
      GangRec *r = &m->gang_recs[pos];
      if (lead && !(r->key == lead->key && r->grid.x == lead->grid.x && r->grid.y == lead->grid.y &&
                    r->grid.z == lead->grid.z && r->block.x == lead->block.x && r->block.y == lead->block.y &&
                    r->block.z == lead->block.z))
        continue;
      if (!lead) lead = r;
      grp[n++] = r;
      shmem = r->shmem > shmem ? r->shmem : shmem;
      ++m->gang_head;
      if (n == lead->cap) break;
    }
    const bool together = n > 1 && lead->grid.z == 1;
    if (together) {
      lead->shmem = shmem;
      lead->launch_many(grp, n, g->stream);
    } else {
      for (int i = 0; i < n; ++i) grp[i]->launch_one(*grp[i], g->stream);
    }
    g->launches += together ? 1 : n;
    g->gang_launches += together ? 1 : 0;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("gang launch (%d members, grid %u x %u, %u threads, %u B LDS): %s", n, lead->grid.x, lead->grid.y,
                lead->block.x, lead->shmem, hipGetErrorString(e));
      rc = SFMLOC_EHIP;
    }
  }
  for (GangMember *m : g->members) {
    m->gang_recs.clear();
    m->gang_head = 0;
  }
  return rc;
}

int gang_open(GangMember *const *members, int n) {
  GangMember *lead = members[0];
  if (!lead->gang_owned) {
    lead->gang_owned = new (std::nothrow) GangState();
    SFM_CHECK(lead->gang_owned, SFMLOC_ENOMEM, "out of host memory");
    SFM_HIP(hipEventCreateWithFlags(&lead->gang_owned->done, hipEventDisableTiming));
  }
  GangState *g = lead->gang_owned;
  SFM_CHECK(lead->stream.own, SFMLOC_EINVAL, "gang session: the leading member has no stream");
  g->stream = lead->stream.own;
  g->members.clear();
  for (int i = 0; i < n; ++i) {
    GangMember *c = members[i];
    c->ever_ganged = true;
    // (own == nullptr: a member that has no stream of its own yet -- an extractor that only ever worked in sessions)
    if (i > 0 && c->stream.own && c->stream.dirty && c->stream.own != g->stream) {  // the member's own earlier work comes first
      if (!c->gang_ev) SFM_HIP(hipEventCreateWithFlags(&c->gang_ev, hipEventDisableTiming));
      SFM_HIP(hipEventRecord(c->gang_ev, c->stream.own));
      SFM_HIP(hipStreamWaitEvent(g->stream, c->gang_ev, 0));
      c->stream.dirty = false;
    }
    g->members.push_back(c);
  }
  for (GangMember *c : g->members) c->stream.gang = g;
  return SFMLOC_OK;
}

int gang_close(GangMember *lead) {
  GangState *g = lead->stream.gang;
  if (!g) return SFMLOC_OK;
  const int rc = gang_flush(g);
  for (GangMember *c : g->members) c->stream.gang = nullptr;
  lead->stream.dirty = true;
  // the members' own streams continue after the gang's work
  bool any_own = false;
  for (size_t i = 1; i < g->members.size(); ++i)
    any_own |= g->members[i]->stream.own && g->members[i]->stream.own != g->stream;
  if (any_own) {
    SFM_HIP(hipEventRecord(g->done, g->stream));
    for (size_t i = 1; i < g->members.size(); ++i)
      if (g->members[i]->stream.own && g->members[i]->stream.own != g->stream) {
        SFM_HIP(hipStreamWaitEvent(g->members[i]->stream.own, g->done, 0));
        // the member's stream now carries a dependency on the session's work: a later session under ANOTHER leader
        // must order itself after it (gang_open records and waits only for dirty members) -- ADVICE r02
        g->members[i]->stream.dirty = true;
      }
  }
  return rc;  // (gang_flush has set the message)
}

void gang_member_free(GangMember *m) {
  if (m->gang_ev) hipEventDestroy(m->gang_ev);
  if (m->gang_owned) {
    if (m->gang_owned->done) hipEventDestroy(m->gang_owned->done);
    delete m->gang_owned;
  }
  m->gang_ev = nullptr;
  m->gang_owned = nullptr;
}

int match_putative_on(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel) {
  return ctx_match_putative(c, q, view_sel, n_sel);
}
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
  p->dist_ratio = 0.6f;         // localization.cpp:70
  p->ransac_round = 200;        // localization.cpp:71
  p->geom_precision = 4.0;      // localization.cpp:81
  p->bow_knn = 0;               // localization.cpp:73
  p->min_putative = 16;         // localization.cpp:56
  p->min_resection_points = 8;  // localization.cpp:57
  p->min_inliers = 10;          // localization.cpp:58
  p->p3p_max_iteration = 4096;  // OpenMVG Image_Localizer_Match_Data default
  p->seed = 0x5f3759df12345678ull;
  p->refine_pose = 0;
  p->device = 0;
  p->profile = 0;
  p->exact_rows = 0;
  p->guided_matching = 0;       // -gm: false (localization.cpp:82, computeFeaturesAndMatches.cpp:63)
}

static int map_create_impl(const sfmloc_map_desc *d, const sfmloc_params *params, sfmloc_map **out);

int sfmloc_map_create(const sfmloc_map_desc *d, const sfmloc_params *params, sfmloc_map **out) {
  try {  // host-side containers are sized by the caller's counts: no exception may cross the C ABI
    return map_create_impl(d, params, out);
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_map_create: out of host memory");
    return SFMLOC_ENOMEM;
  } catch (const std::exception &e) {
    set_error("sfmloc_map_create: %s", e.what());
    return SFMLOC_EINVAL;
  }
}

static int map_create_impl(const sfmloc_map_desc *d, const sfmloc_params *params, sfmloc_map **out) {
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
    SFM_CHECK(d->view_id[v] < (1u << 24), SFMLOC_EINVAL, "sfmloc_map_create: view id %u >= 2^24", d->view_id[v]);
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
  hipStream_t s = nullptr;  // upload stream
#define SFM_TRY(x)                   \
  do {                               \
    rc = (x);                        \
    if (rc) {                        \
      if (s) hipStreamDestroy(s);    \
      free_map(m);                   \
      return rc;                     \
    }                                \
  } while (0)
#define MAP_HIP(x)                                                    \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      set_error("%s -> %s", #x, hipGetErrorString(e_));               \
      if (s) hipStreamDestroy(s);                                     \
      free_map(m);                                                    \
      return e_ == hipErrorOutOfMemory ? SFMLOC_ENOMEM : SFMLOC_EHIP; \
    }                                                                 \
  } while (0)
  MAP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  m->n_rows = d->n_rows;
  m->n_blocks = (uint32_t)((d->n_rows + kBlockRows - 1) / kBlockRows);
  m->max_view_blocks = 0;
  for (uint32_t v = 0; v < d->n_views; ++v) {
    const uint32_t r0 = d->view_off[v], r1 = d->view_off[v + 1];
    if (r1 > r0) m->max_view_blocks = std::max(m->max_view_blocks, (r1 - 1) / kBlockRows - r0 / kBlockRows + 1);
    m->max_view_rows = std::max(m->max_view_rows, r1 - r0);
  }
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
  SFM_CHECK(d->intrinsic_type == 0 || d->intrinsic_type == 3, SFMLOC_EINVAL,
            "sfmloc_map_create: intrinsic_type %u (0 = pinhole, 3 = pinhole_radial_k3)", d->intrinsic_type);
  m->intrinsic_type = d->intrinsic_type;
  uint64_t *acct = &m->hbm_bytes;

  // bank: upload row-major chunks and re-tile on the device
  const uint64_t n_pad = (uint64_t)m->n_blocks * kBlockRows;
  SFM_TRY(dev_alloc(acct, &m->d_bank, (size_t)n_pad * 4));
  if (n_pad) {
    MAP_HIP(hipMemsetAsync(m->d_bank, 0, n_pad * 64, s));
    const uint64_t chunk = 16ull << 20;  // rows per staging chunk (1 GiB)
    uint4 *d_stage = nullptr;
    const uint64_t stage_rows = std::min<uint64_t>(chunk, d->n_rows);
    if (stage_rows) MAP_HIP(hipMalloc((void **)&d_stage, stage_rows * 64));
    for (uint64_t r0 = 0; r0 < d->n_rows; r0 += chunk) {
      const uint64_t n = std::min<uint64_t>(chunk, d->n_rows - r0);
      hipError_t ce = hipMemcpyAsync(d_stage, d->desc + r0 * 64, n * 64, hipMemcpyHostToDevice, s);
      if (ce == hipSuccess) {
        rc = launch_tile_bank(d_stage, r0, n, m->d_bank, s);
        if (!rc) ce = hipStreamSynchronize(s);
      }
      if (ce != hipSuccess || rc) {
        if (ce != hipSuccess) {
          set_error("bank upload: %s", hipGetErrorString(ce));
          rc = SFMLOC_EHIP;
        }
        hipFree(d_stage);
        hipStreamDestroy(s);
        free_map(m);
        return rc;
      }
    }
    if (d_stage) hipFree(d_stage);
  }

  {  // view tables + the phantom (empty) view at index n_views (sfmloc_internal.h)
    std::vector<uint32_t> off(d->view_off, d->view_off + d->n_views + 1), id(d->view_id, d->view_id + d->n_views);
    off.push_back((uint32_t)d->n_rows);
    id.push_back(0xFFFFFFu);
    SFM_TRY(dev_upload(acct, &m->d_view_off, off.data(), off.size(), s));
    SFM_TRY(dev_upload(acct, &m->d_view_id, id.data(), id.size(), s));
    if (d->view_wh) {
      std::vector<uint32_t> wh(d->view_wh, d->view_wh + 2 * (size_t)d->n_views);
      wh.push_back(1);
      wh.push_back(1);
      SFM_TRY(dev_upload(acct, &m->d_view_wh, wh.data(), wh.size(), s));
    }
    MAP_HIP(hipStreamSynchronize(s));  // the staging vectors go out of scope
  }
  if (d->kpt_xy) SFM_TRY(dev_upload(acct, (float **)&m->d_kpt, d->kpt_xy, (size_t)d->n_rows * 2, s));
  if (d->row_landmark) {
    SFM_TRY(dev_upload(acct, &m->d_row_landmark, d->row_landmark, (size_t)d->n_rows, s));
    SFM_TRY(dev_upload(acct, &m->d_landmark_id, d->landmark_id, (size_t)d->n_landmarks, s));
    SFM_TRY(dev_upload(acct, &m->d_landmark_X, d->landmark_X, (size_t)d->n_landmarks * 3, s));
  }
  if (d->bow && d->bow_dim) {
    m->bow_dim = d->bow_dim;
    SFM_TRY(dev_upload(acct, &m->d_bow, d->bow, (size_t)d->n_views * d->bow_dim, s));
  }
  m->have_geometry = d->view_wh && d->kpt_xy && d->row_landmark && d->focal > 0.0;
  SFM_TRY(dev_alloc(acct, &m->d_L10, (size_t)65538));
  SFM_TRY(launch_fill_log10(m->d_L10, 65538, s));
  {
    uint16_t tab[513];
    build_ratio_table(p.dist_ratio, tab);
    SFM_TRY(dev_alloc(acct, &m->d_ratio_cnt, (size_t)513));
    MAP_HIP(hipMemcpyAsync(m->d_ratio_cnt, tab, sizeof(tab), hipMemcpyHostToDevice, s));
    MAP_HIP(hipStreamSynchronize(s));  // tab lives on the stack
    m->ratio_cnt_for = p.dist_ratio;
  }
  MAP_HIP(hipStreamSynchronize(s));
  hipStreamDestroy(s);
  s = nullptr;
  SFM_TRY(make_ctx(m, &m->ctx0));
  m->hbm_bytes += m->ctx0->hbm_bytes;
#undef SFM_TRY
#undef MAP_HIP
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

int sfmloc_context_create(sfmloc_map *map, sfmloc_context **out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_context_create: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  Ctx *c = nullptr;
  int rc = make_ctx(m, &c);
  if (rc) return rc;
  m->pool.push_back(c);
  m->hbm_bytes += c->hbm_bytes;
  *out = reinterpret_cast<sfmloc_context *>(c);
  return SFMLOC_OK;
}

int sfmloc_context_create_sharing(sfmloc_map *map, sfmloc_context *lender, sfmloc_context **out) {
  SFM_CHECK(map && lender && out, SFMLOC_EINVAL, "sfmloc_context_create_sharing: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *l = reinterpret_cast<Ctx *>(lender);
  SFM_CHECK(l->map == m, SFMLOC_EINVAL, "sfmloc_context_create_sharing: the lender belongs to another map");
  SFM_HIP(hipSetDevice(m->device));
  Ctx *c = nullptr;
  int rc = make_ctx(m, &c, l);
  if (rc) return rc;
  m->pool.push_back(c);
  m->hbm_bytes += c->hbm_bytes;
  *out = reinterpret_cast<sfmloc_context *>(c);
  return SFMLOC_OK;
}

int sfmloc_context_create_merge(sfmloc_map *map, sfmloc_context *lender, sfmloc_context **out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_context_create_merge: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *l = reinterpret_cast<Ctx *>(lender);
  SFM_CHECK(!l || l->map == m, SFMLOC_EINVAL, "sfmloc_context_create_merge: the lender belongs to another map");
  SFM_HIP(hipSetDevice(m->device));
  Ctx *c = nullptr;
  int rc = make_ctx(m, &c, l, /*merge_only=*/true);
  if (rc) return rc;
  m->pool.push_back(c);
  m->hbm_bytes += c->hbm_bytes;
  *out = reinterpret_cast<sfmloc_context *>(c);
  return SFMLOC_OK;
}

void sfmloc_context_destroy(sfmloc_context *ctx) {
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  if (!c) return;
  Map *m = c->map;
  hipSetDevice(m->device);
  ctx_mark_idle(c);
  auto it = std::find(m->pool.begin(), m->pool.end(), c);
  if (it != m->pool.end()) m->pool.erase(it);
  auto ib = std::find(m->batch_ctx.begin(), m->batch_ctx.end(), c);
  if (ib != m->batch_ctx.end()) m->batch_ctx.erase(ib);
  m->hbm_bytes -= c->hbm_bytes;
  free_ctx(c);
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
  hipStream_t s = m->ctx0->stream;
  if (n_pad) {
    e = hipMalloc((void **)&q->d_desc, n_pad * 64);
    if (e == hipSuccess) e = hipMemsetAsync(q->d_desc, 0, n_pad * 64, s);
    if (e == hipSuccess) e = hipMemcpyAsync(q->d_desc, desc, (size_t)n * 64, hipMemcpyHostToDevice, s);
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
      if (e == hipSuccess) e = hipMemcpyAsync(q->d_kpt, kpt_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, s);
      if (e == hipSuccess)
        e = hipMemcpyAsync(q->d_kpt6, k6.data(), (size_t)n * sizeof(float2), hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
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

void sfmloc_feat_round_trip(const float *kpt_xy, uint64_t n_values, float *out) {
  char buf[64];
  for (uint64_t i = 0; i < n_values; ++i) {  // (the same two lines as in sfmloc_query_create)
    snprintf(buf, sizeof(buf), "%.6g", (double)kpt_xy[i]);
    out[i] = strtof(buf, nullptr);
  }
}

int sfmloc_query_create_view(sfmloc_map *map, const void *desc_dev, const void *kpt_dev, const void *kpt6_dev,
                             const void *bow_dev, uint32_t n, uint32_t width, uint32_t height, sfmloc_query **out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_query_create_view: null argument");
  *out = nullptr;
  SFM_CHECK(n == 0 || (desc_dev && kpt_dev && kpt6_dev), SFMLOC_EINVAL, "sfmloc_query_create_view: null device pointer");
  SFM_CHECK(n <= SFMLOC_MAX_QUERY_ROWS, SFMLOC_EINVAL, "sfmloc_query_create_view: %u query descriptors > %u", n,
            SFMLOC_MAX_QUERY_ROWS);
  SFM_CHECK(((uintptr_t)desc_dev & 15) == 0 && ((uintptr_t)kpt_dev & 7) == 0 && ((uintptr_t)kpt6_dev & 7) == 0 &&
                ((uintptr_t)bow_dev & 3) == 0,
            SFMLOC_EINVAL, "sfmloc_query_create_view: misaligned device pointer (descriptors 16, keypoints 8, bow 4 bytes)");
  Query *q = new (std::nothrow) Query();
  SFM_CHECK(q, SFMLOC_ENOMEM, "out of host memory");
  q->map = reinterpret_cast<Map *>(map);
  q->n = n;
  q->width = width;
  q->height = height;
  q->is_view = true;
  q->d_desc = reinterpret_cast<uint4 *>(const_cast<void *>(desc_dev));
  q->d_kpt = reinterpret_cast<float2 *>(const_cast<void *>(kpt_dev));
  q->d_kpt6 = reinterpret_cast<float2 *>(const_cast<void *>(kpt6_dev));
  q->d_bow = reinterpret_cast<float *>(const_cast<void *>(bow_dev));
  *out = reinterpret_cast<sfmloc_query *>(q);
  return SFMLOC_OK;
}

void sfmloc_query_destroy(sfmloc_query *query) {
  Query *q = reinterpret_cast<Query *>(query);
  if (!q) return;
  if (q->map) {
    hipSetDevice(q->map->device);
    // only the contexts that worked on THIS query can still be reading it: the others keep running (a server destroys a
    // query per request while other users' queries are in flight)
    std::vector<Ctx *> all = q->map->pool;
    if (q->map->ctx0) all.push_back(q->map->ctx0);
    for (Ctx *c : all)
      if (c->last_query == q || c->in_flight == q) {
        hipStreamSynchronize(c->stream);
        if (c->last_query == q) c->last_query = nullptr;
      }
  }
  if (!q->is_view) {  // (a view's arrays are the caller's)
    if (q->d_desc) hipFree(q->d_desc);
    if (q->d_kpt) hipFree(q->d_kpt);
    if (q->d_kpt6) hipFree(q->d_kpt6);
    if (q->d_bow) hipFree(q->d_bow);
  }
  delete q;
}

int sfmloc_query_set_bow(sfmloc_query *query, const float *query_bow) {
  SFM_CHECK(query && query_bow, SFMLOC_EINVAL, "sfmloc_query_set_bow: null argument");
  Query *q = reinterpret_cast<Query *>(query);
  Map *m = q->map;
  SFM_CHECK(m && m->bow_dim > 0, SFMLOC_EINVAL, "sfmloc_query_set_bow: the map has no .bow vectors");
  SFM_CHECK(!q->is_view, SFMLOC_EINVAL, "sfmloc_query_set_bow: a view's BoW vector is given to sfmloc_query_create_view");
  SFM_HIP(hipSetDevice(m->device));
  if (!q->d_bow) SFM_HIP(hipMalloc((void **)&q->d_bow, (size_t)m->bow_dim * sizeof(float)));
  SFM_HIP(hipMemcpy(q->d_bow, query_bow, (size_t)m->bow_dim * sizeof(float), hipMemcpyHostToDevice));
  return SFMLOC_OK;
}

// ----- stage-level API on the map's own context -------------------------------------------------------

int sfmloc_match_putative(sfmloc_map *map, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel) {
  SFM_CHECK(map && query, SFMLOC_EINVAL, "sfmloc_match_putative: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_match_putative: query belongs to another map");
  SFM_HIP(hipSetDevice(m->device));
  return ctx_match_putative(m->ctx0, q, view_sel, n_sel);
}

int sfmloc_putative_read(sfmloc_map *map, uint32_t *view_count, uint32_t *match_i, uint32_t *match_j,
                         uint32_t *match_d, uint64_t cap) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_putative_read: null map");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *c = m->ctx0;
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(c->stream));
  std::vector<uint32_t> cnt(m->n_views);
  SFM_HIP(hipMemcpy(cnt.data(), c->d_view_count, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (view_count) memcpy(view_count, cnt.data(), cnt.size() * sizeof(uint32_t));
  if (match_i || match_j || match_d) {
    SFM_CHECK(cap >= m->n_rows, SFMLOC_ECAP, "sfmloc_putative_read: cap %llu < n_rows %llu", (unsigned long long)cap,
              (unsigned long long)m->n_rows);
    std::vector<uint32_t> hi(m->n_rows), hk(m->n_rows);
    SFM_HIP(hipMemcpy(hi.data(), c->d_match_i, hi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SFM_HIP(hipMemcpy(hk.data(), c->d_match_key, hk.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
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
  Ctx *c = m->ctx0;
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(c->stream));
  for (uint64_t r = 0; r < m->n_rows; ++r) {
    if (best0) best0[r] = SFMLOC_NOMATCH;
    if (best1) best1[r] = SFMLOC_NOMATCH;
  }
  if (c->last_nq == 0 || c->last_n_sel == 0 || c->last_split == 0 || c->last_n_work_blocks == 0) return SFMLOC_OK;
  const uint64_t nwb = c->last_n_work_blocks;
  std::vector<uint2> part((size_t)c->last_split * nwb * 64);
  SFM_HIP(hipMemcpy(part.data(), c->d_part, part.size() * sizeof(uint2), hipMemcpyDeviceToHost));
  std::vector<unsigned long long> mask;
  if (c->last_screened) {  // only the rows the screening scan could not reject have a pair; the others stay NOMATCH
    mask.resize(nwb);
    SFM_HIP(hipMemcpy(mask.data(), c->d_flagmask, nwb * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  auto push = [](uint32_t &b0, uint32_t &b1, uint32_t k) {
    if (k < b0) {
      b1 = b0;
      b0 = k;
    } else if (k < b1) {
      b1 = k;
    }
  };
  if (c->last_blocks_on_device) {  // the list was built on the device (BoW shortlist); padding names no block
    c->last_blocks.resize(nwb);
    SFM_HIP(hipMemcpy(c->last_blocks.data(), c->d_block_list, nwb * sizeof(uint32_t), hipMemcpyDeviceToHost));
    c->last_blocks_on_device = false;
  }
  for (uint64_t w = 0; w < nwb; ++w) {
    const uint32_t blk = c->last_all_views ? (uint32_t)w : c->last_blocks[w];
    if (blk == 0xFFFFFFFFu) continue;
    for (uint32_t l = 0; l < kBlockRows; ++l) {
      const uint64_t r = (uint64_t)blk * kBlockRows + l;
      if (r >= m->n_rows) break;
      if (c->last_screened && !((mask[w] >> l) & 1ull)) continue;
      uint32_t b0 = SFMLOC_NOMATCH, b1 = SFMLOC_NOMATCH;
      for (uint32_t s = 0; s < c->last_split; ++s) {
        const uint2 p = part[((size_t)s * nwb + w) * 64 + l];
        push(b0, b1, p.x);
        push(b0, b1, p.y);
      }
      if (best0) best0[r] = b0;
      if (best1) best1[r] = b1;
    }
  }
  return SFMLOC_OK;
}

int sfmloc_geometric_filter(sfmloc_map *map, sfmloc_query *query) {
  SFM_CHECK(map && query, SFMLOC_EINVAL, "sfmloc_geometric_filter: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  int rc = check_stage(m->ctx0, q, "sfmloc_geometric_filter");
  if (rc) return rc;
  SFM_HIP(hipSetDevice(m->device));
  return ctx_geometric_filter(m->ctx0, q);
}

int sfmloc_geometric_read(sfmloc_map *map, uint32_t *geo_count, uint32_t *geo_idx, uint64_t cap) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_geometric_read: null map");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *c = m->ctx0;
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(c->stream));
  std::vector<uint32_t> cnt(m->n_views);
  SFM_HIP(hipMemcpy(cnt.data(), c->d_geo_count, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (geo_count) memcpy(geo_count, cnt.data(), cnt.size() * sizeof(uint32_t));
  if (geo_idx) {
    SFM_CHECK(!c->geo_is_pairs, SFMLOC_EINVAL,
              "sfmloc_geometric_read: guided matches are not indices into the putative lists; use sfmloc_geometric_read_pairs");
    SFM_CHECK(cap >= m->n_rows, SFMLOC_ECAP, "sfmloc_geometric_read: cap too small");
    std::vector<uint32_t> gi(m->n_rows);
    SFM_HIP(hipMemcpy(gi.data(), c->d_geo_idx, gi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t v = 0; v < m->n_views; ++v)
      for (uint32_t k = 0; k < cnt[v]; ++k) geo_idx[m->h_view_off[v] + k] = gi[m->h_view_off[v] + k];
  }
  int st = 0;
  SFM_HIP(hipMemcpy(&st, c->d_status, sizeof(int), hipMemcpyDeviceToHost));
  SFM_CHECK((st & 1) == 0, SFMLOC_ECAP, "a view has more than 65536 putative matches");
  return SFMLOC_OK;
}

int sfmloc_geometric_read_pairs(sfmloc_map *map, uint32_t *geo_count, uint32_t *geo_i, uint32_t *geo_j, uint64_t cap) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_geometric_read_pairs: null map");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *c = m->ctx0;
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(c->stream));
  std::vector<uint32_t> cnt(m->n_views);
  SFM_HIP(hipMemcpy(cnt.data(), c->d_geo_count, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (geo_count) memcpy(geo_count, cnt.data(), cnt.size() * sizeof(uint32_t));
  if (geo_i || geo_j) {
    SFM_CHECK(cap >= m->n_rows, SFMLOC_ECAP, "sfmloc_geometric_read_pairs: cap too small");
    std::vector<uint32_t> gi(m->n_rows), gj;
    SFM_HIP(hipMemcpy(gi.data(), c->d_geo_idx, gi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> pi, pk;
    if (c->geo_is_pairs) {
      gj.resize(m->n_rows);
      SFM_HIP(hipMemcpy(gj.data(), c->d_geo_j, gj.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    } else {  // indices into the putative lists: resolve them
      pi.resize(m->n_rows);
      pk.resize(m->n_rows);
      SFM_HIP(hipMemcpy(pi.data(), c->d_match_i, pi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      SFM_HIP(hipMemcpy(pk.data(), c->d_match_key, pk.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    for (uint32_t v = 0; v < m->n_views; ++v) {
      const uint32_t off = m->h_view_off[v];
      for (uint32_t k = 0; k < cnt[v]; ++k) {
        const uint32_t g = gi[off + k];
        if (geo_i) geo_i[off + k] = c->geo_is_pairs ? g : pi[off + g];
        if (geo_j) geo_j[off + k] = c->geo_is_pairs ? gj[off + k] : (pk[off + g] & 0xFFFFu);
      }
    }
  }
  return SFMLOC_OK;
}

int sfmloc_match_set(sfmloc_map *map, sfmloc_query *query) {
  SFM_CHECK(map && query, SFMLOC_EINVAL, "sfmloc_match_set: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  int rc = check_stage(m->ctx0, q, "sfmloc_match_set");
  if (rc) return rc;
  SFM_HIP(hipSetDevice(m->device));
  rc = ctx_p3p_reserve(m->ctx0, q->n);  // K5's start runs at the end of this stage (k_match_set_finish)
  if (rc) return rc;
  return ctx_match_set(m->ctx0, q);
}

int sfmloc_match_set_read(sfmloc_map *map, uint32_t *n, uint32_t *qfeat, uint32_t *landmark_id, double *pt2d,
                          double *pt3d, uint32_t cap) {
  SFM_CHECK(map && n, SFMLOC_EINVAL, "sfmloc_match_set_read: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *c = m->ctx0;
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(c->stream));
  uint32_t k = 0;
  SFM_HIP(hipMemcpy(&k, c->d_ms_n, sizeof(uint32_t), hipMemcpyDeviceToHost));
  *n = k;
  int st = 0;
  SFM_HIP(hipMemcpy(&st, c->d_status, sizeof(int), hipMemcpyDeviceToHost));
  SFM_CHECK((st & 2) == 0, SFMLOC_ECAP, "more than %u 2D-3D candidates (match-set workspace)", c->cand_cap);
  if (k == 0) return SFMLOC_OK;
  SFM_CHECK(cap >= k, SFMLOC_ECAP, "sfmloc_match_set_read: cap %u < %u", cap, k);
  if (qfeat) SFM_HIP(hipMemcpy(qfeat, c->d_ms_qfeat, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (landmark_id) SFM_HIP(hipMemcpy(landmark_id, c->d_ms_landmark, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (pt2d) SFM_HIP(hipMemcpy(pt2d, c->d_pt2d, (size_t)k * 2 * sizeof(double), hipMemcpyDeviceToHost));
  if (pt3d) SFM_HIP(hipMemcpy(pt3d, c->d_pt3d, (size_t)k * 3 * sizeof(double), hipMemcpyDeviceToHost));
  return SFMLOC_OK;
}

int sfmloc_resection(sfmloc_map *map, sfmloc_query *query) {
  SFM_CHECK(map && query, SFMLOC_EINVAL, "sfmloc_resection: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  Ctx *c = m->ctx0;
  int rc = check_stage(c, q, "sfmloc_resection");
  if (rc) return rc;
  SFM_HIP(hipSetDevice(m->device));
  rc = ctx_p3p_reserve(c, q->n);
  if (rc) return rc;
  {
    EventScope ev(c, SFMLOC_K_P3P);
    rc = ctx_resection_enqueue(c, true);
  }
  if (rc) return rc;
  rc = ctx_fetch_result(c);
  if (rc) return rc;
  return ctx_resection_wait(c);
}

int sfmloc_pose_read(sfmloc_map *map, sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark,
                     uint32_t *inlier_idx, uint32_t cap) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_pose_read: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *c = m->ctx0;
  SFM_HIP(hipSetDevice(m->device));
  SFM_HIP(hipStreamSynchronize(c->stream));
  SFM_HIP(hipMemcpy(out, c->d_pose, sizeof(Pose), hipMemcpyDeviceToHost));
  SFM_CHECK((out->status & 4) == 0, SFMLOC_ECAP, "more than %u 2D-3D correspondences (P3P workspace)", c->p3p_cap);
  if (out->ok && out->n_inliers > 0) {
    const uint32_t k = (uint32_t)out->n_inliers;
    if (pair_qfeat || pair_landmark || inlier_idx)
      SFM_CHECK(cap >= k, SFMLOC_ECAP, "sfmloc_pose_read: cap %u < %u inliers", cap, k);
    if (pair_qfeat) SFM_HIP(hipMemcpy(pair_qfeat, c->d_pair_qfeat, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (pair_landmark)
      SFM_HIP(hipMemcpy(pair_landmark, c->d_pair_landmark, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (inlier_idx) SFM_HIP(hipMemcpy(inlier_idx, c->d_inlier_idx, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  return SFMLOC_OK;
}

// ----- whole queries --------------------------------------------------------------------------------

int sfmloc_localize_begin(sfmloc_context *ctx, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel) {
  SFM_CHECK(ctx && query, SFMLOC_EINVAL, "sfmloc_localize_begin: null argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  Query *q = reinterpret_cast<Query *>(query);
  SFM_CHECK(q->map == c->map, SFMLOC_EINVAL, "sfmloc_localize_begin: query belongs to another map");
  SFM_HIP(hipSetDevice(c->map->device));
  return ctx_localize_begin(c, q, view_sel, n_sel);
}

int sfmloc_localize_bow_begin(sfmloc_context *ctx, sfmloc_query *query, const float *query_bow, uint32_t knn,
                              const uint32_t *cand_views, uint32_t n_cand) {
  SFM_CHECK(ctx && query, SFMLOC_EINVAL, "sfmloc_localize_bow_begin: null argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  Query *q = reinterpret_cast<Query *>(query);
  Map *m = c->map;
  SFM_CHECK(query_bow || q->d_bow, SFMLOC_EINVAL,
            "sfmloc_localize_bow_begin: no BoW vector (pass one, or make it resident with sfmloc_query_set_bow)");
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_localize_bow_begin: query belongs to another map");
  SFM_CHECK(m->bow_dim > 0 && m->d_bow, SFMLOC_EINVAL, "sfmloc_localize_bow_begin: the map has no .bow vectors");
  SFM_CHECK(c->in_flight == nullptr, SFMLOC_EINVAL, "sfmloc_localize_bow_begin: context already has a query in flight");
  SFM_CHECK(!c->merge_only, SFMLOC_EINVAL, "sfmloc_localize_bow_begin: this context was created for sfmloc_merge_begin only");
  if (!cand_views) n_cand = m->n_views;
  SFM_CHECK(n_cand <= m->n_views, SFMLOC_EINVAL, "sfmloc_localize_bow_begin: n_cand > n_views");
  SFM_HIP(hipSetDevice(m->device));
  // "if (knnbow > 0 && viewList.size() > knnbow)" (localization.cpp:346, LocalizeEngine.cc:342): otherwise every
  // candidate is matched
  if (knn == 0 || n_cand <= knn) return ctx_localize_begin(c, q, cand_views, cand_views ? n_cand : 0);
  if (cand_views) {
    for (uint32_t i = 0; i < n_cand; ++i) {
      SFM_CHECK(cand_views[i] < m->n_views, SFMLOC_EINVAL, "sfmloc_localize_bow_begin: view index out of range");
      SFM_CHECK(i == 0 || cand_views[i - 1] < cand_views[i], SFMLOC_EINVAL,
                "sfmloc_localize_bow_begin: candidate views must be strictly ascending");
    }
    SFM_HIP(hipMemcpyAsync(c->d_bow_cand, cand_views, n_cand * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  if (query_bow)
    SFM_HIP(hipMemcpyAsync(c->d_bow_query, query_bow, m->bow_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
  int rc;
  {
    EventScope ev(c, SFMLOC_K_BOW);
    // the workgroup that finishes the shortlist also clears the query's counters and builds the block list
    const ChainArgs chain = make_chain_args(c, q, c->d_bow_sel, knn);
    rc = launch_bow_select(c, query_bow ? c->d_bow_query : q->d_bow, cand_views ? c->d_bow_cand : nullptr,
                           n_cand, knn, c->d_bow_dist, c->d_bow_sel, &chain);
  }
  if (rc) return rc;
  // K8 leaves the knn views in ascending order in d_bow_sel; everything downstream reads the selection on the device
  uint32_t dummy = 0;
  c->chain_done = true;
  rc = ctx_localize_begin(c, q, &dummy, knn, c->d_bow_sel);
  c->chain_done = false;
  return rc;
}

int sfmloc_localize_end(sfmloc_context *ctx, sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark,
                        uint32_t cap) {
  SFM_CHECK(ctx && out, SFMLOC_EINVAL, "sfmloc_localize_end: null argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_HIP(hipSetDevice(c->map->device));
  return ctx_localize_end(c, out, pair_qfeat, pair_landmark, cap);
}

// ----- sharded maps: one process per GPU, bank sharded by view ------------------------------

uint64_t sfmloc_part_bytes(uint32_t cap) { return (uint64_t)kPartHeaderBytes + (uint64_t)cap * sizeof(Candidate); }

int sfmloc_shard_begin(sfmloc_context *ctx, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel) {
  SFM_CHECK(ctx && query, SFMLOC_EINVAL, "sfmloc_shard_begin: null argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  Query *q = reinterpret_cast<Query *>(query);
  SFM_CHECK(q->map == c->map, SFMLOC_EINVAL, "sfmloc_shard_begin: query belongs to another map");
  SFM_CHECK(c->in_flight == nullptr, SFMLOC_EINVAL, "sfmloc_shard_begin: context has a query in flight");
  SFM_HIP(hipSetDevice(c->map->device));
  ctx_mark_busy(c);  // until sfmloc_context_sync
  ClearedScope cs{c};
  int rc = ctx_reset_for_query(c, q);
  if (rc) return rc;
  c->defer_merge = true;  // K3 follows on this context
  rc = ctx_match_putative(c, q, view_sel, n_sel);
  c->defer_merge = false;
  if (rc) return rc;
  rc = check_stage(c, q, "sfmloc_shard_begin");
  if (rc) return rc;
  rc = ctx_geometric_filter(c, q);
  if (rc) return rc;
  EventScope ev(c, SFMLOC_K_MATCHSET);
  return launch_emit_candidates(c, q, c->last_n_sel, c->last_all_views);
}

int sfmloc_shard_export(sfmloc_context *ctx, void *dst_dev, uint32_t cap) {
  SFM_CHECK(ctx && dst_dev && cap > 0, SFMLOC_EINVAL, "sfmloc_shard_export: bad argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_HIP(hipSetDevice(c->map->device));
  // header (with the true count: the merging side flags count > cap) + the candidates that exist, at most cap
  return launch_export_part(c, dst_dev, cap < c->cand_cap ? cap : c->cand_cap);
}

int sfmloc_shard_bow_keys(sfmloc_context *ctx, sfmloc_query *query, const float *query_bow, uint32_t knn,
                          void *keys_dev) {
  SFM_CHECK(ctx && query && keys_dev && knn > 0, SFMLOC_EINVAL, "sfmloc_shard_bow_keys: bad argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  Query *q = reinterpret_cast<Query *>(query);
  Map *m = c->map;
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_shard_bow_keys: query belongs to another map");
  SFM_CHECK(m->bow_dim > 0 && m->d_bow, SFMLOC_EINVAL, "sfmloc_shard_bow_keys: the map has no .bow vectors");
  SFM_CHECK(query_bow || q->d_bow, SFMLOC_EINVAL, "sfmloc_shard_bow_keys: no BoW vector");
  SFM_CHECK(c->in_flight == nullptr, SFMLOC_EINVAL, "sfmloc_shard_bow_keys: context has a query in flight");
  SFM_CHECK(knn <= 1024, SFMLOC_EINVAL, "sfmloc_shard_bow_keys: knn %u > 1024", knn);
  SFM_HIP(hipSetDevice(m->device));
  ctx_mark_busy(c);
  if (query_bow)
    SFM_HIP(hipMemcpyAsync(c->d_bow_query, query_bow, m->bow_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
  EventScope ev(c, SFMLOC_K_BOW);
  return launch_bow_keys(c, query_bow ? c->d_bow_query : q->d_bow, knn, c->d_bow_dist, c->d_bow_cand,
                         reinterpret_cast<unsigned long long *>(keys_dev));
}

int sfmloc_shard_begin_bow(sfmloc_context *ctx, sfmloc_query *query, const void *keys_dev, uint32_t n_parts,
                           uint64_t part_stride_keys, uint32_t knn) {
  SFM_CHECK(ctx && query && keys_dev && n_parts > 0 && knn > 0, SFMLOC_EINVAL, "sfmloc_shard_begin_bow: bad argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  Query *q = reinterpret_cast<Query *>(query);
  Map *m = c->map;
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_shard_begin_bow: query belongs to another map");
  SFM_CHECK(c->in_flight == nullptr, SFMLOC_EINVAL, "sfmloc_shard_begin_bow: context has a query in flight");
  SFM_CHECK(!c->merge_only, SFMLOC_EINVAL, "sfmloc_shard_begin_bow: this context was created for sfmloc_merge_begin only");
  if (part_stride_keys == 0) part_stride_keys = knn;
  SFM_CHECK(part_stride_keys >= knn, SFMLOC_EINVAL, "sfmloc_shard_begin_bow: part_stride_keys < knn");
  SFM_CHECK(m->bow_dim > 0 && c->d_bow_sel, SFMLOC_EINVAL, "sfmloc_shard_begin_bow: the map has no BoW vectors");
  SFM_HIP(hipSetDevice(m->device));
  ctx_mark_busy(c);  // until sfmloc_context_sync
  ClearedScope cs{c};
  // this shard's part of the global knn best: ascending local view indices, padded with the phantom view up to
  // n_pad = min(knn, n_views) entries -- the launch sizes below depend on n_pad only, never on the outcome
  const uint32_t n_pad = knn < m->n_views ? knn : m->n_views;
  int rc;
  {
    EventScope ev(c, SFMLOC_K_BOW);
    // the workgroup that merges the key lists also clears the query's counters and builds the block list
    const ChainArgs chain = make_chain_args(c, q, c->d_bow_sel, n_pad);
    rc = launch_bow_merge_select(c, reinterpret_cast<const unsigned long long *>(keys_dev), n_parts,
                                 part_stride_keys, knn, n_pad, c->d_bow_sel, n_pad ? &chain : nullptr);
  }
  if (rc) return rc;
  c->chain_done = n_pad != 0;
  rc = ctx_reset_for_query(c, q);
  uint32_t dummy = 0;
  c->defer_merge = true;  // K3 follows on this context
  if (!rc) rc = ctx_match_putative(c, q, &dummy, n_pad, c->d_bow_sel);
  c->defer_merge = false;
  c->chain_done = false;
  if (rc) return rc;
  rc = check_stage(c, q, "sfmloc_shard_begin_bow");
  if (rc) return rc;
  rc = ctx_geometric_filter(c, q);
  if (rc) return rc;
  EventScope ev(c, SFMLOC_K_MATCHSET);
  return launch_emit_candidates(c, q, c->last_n_sel, c->last_all_views);
}

int sfmloc_context_signal(sfmloc_context *ctx, void *hip_stream) {
  SFM_CHECK(ctx, SFMLOC_EINVAL, "sfmloc_context_signal: null context");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_HIP(hipSetDevice(c->map->device));
  SFM_HIP(hipEventRecord(c->xev_out, c->stream));
  SFM_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(hip_stream), c->xev_out, 0));
  return SFMLOC_OK;
}

int sfmloc_context_wait(sfmloc_context *ctx, void *hip_stream) {
  SFM_CHECK(ctx, SFMLOC_EINVAL, "sfmloc_context_wait: null context");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_HIP(hipSetDevice(c->map->device));
  SFM_HIP(hipEventRecord(c->xev_in, reinterpret_cast<hipStream_t>(hip_stream)));
  SFM_HIP(hipStreamWaitEvent(c->stream, c->xev_in, 0));
  return SFMLOC_OK;
}

int sfmloc_gang_begin(sfmloc_context *const *ctxs, uint32_t n) {
  SFM_CHECK(ctxs && n >= 1 && n <= (uint32_t)kGangMembers, SFMLOC_EINVAL, "sfmloc_gang_begin: 1..%d contexts", kGangMembers);
  Ctx *lead = reinterpret_cast<Ctx *>(ctxs[0]);
  SFM_CHECK(lead, SFMLOC_EINVAL, "sfmloc_gang_begin: null context");
  for (uint32_t i = 0; i < n; ++i) {
    Ctx *c = reinterpret_cast<Ctx *>(ctxs[i]);
    SFM_CHECK(c && c->map == lead->map, SFMLOC_EINVAL, "sfmloc_gang_begin: the contexts belong to different maps");
    SFM_CHECK(c->stream.gang == nullptr, SFMLOC_EINVAL, "sfmloc_gang_begin: context %u is already in a gang session", i);
    for (uint32_t j = 0; j < i; ++j) SFM_CHECK(ctxs[j] != ctxs[i], SFMLOC_EINVAL, "sfmloc_gang_begin: context listed twice");
  }
  // (stage brackets are events on the stream around every launch: with profiling on, the session records nothing and the
  // members simply run one after the other)
  if (n == 1 || lead->map->params.profile != 0) return SFMLOC_OK;
  SFM_HIP(hipSetDevice(lead->map->device));
  GangMember *ms[kGangMembers];
  for (uint32_t i = 0; i < n; ++i) ms[i] = reinterpret_cast<Ctx *>(ctxs[i]);
  return gang_open(ms, (int)n);
}

int sfmloc_gang_end(sfmloc_context *const *ctxs, uint32_t n) {
  SFM_CHECK(ctxs && n >= 1 && n <= (uint32_t)kGangMembers && ctxs[0], SFMLOC_EINVAL, "sfmloc_gang_end: 1..%d contexts", kGangMembers);
  Ctx *lead = reinterpret_cast<Ctx *>(ctxs[0]);
  GangState *g = lead->stream.gang;
  if (!g) return SFMLOC_OK;  // the session recorded nothing (one member, or profiling)
  SFM_CHECK(g == lead->gang_owned && g->members.size() == n, SFMLOC_EINVAL, "sfmloc_gang_end: not the contexts of the session");
  SFM_HIP(hipSetDevice(lead->map->device));
  return gang_close(lead);
}

int sfmloc_gang_counters(sfmloc_context *lead_ctx, uint64_t *launches, uint64_t *gang_launches) {
  SFM_CHECK(lead_ctx, SFMLOC_EINVAL, "sfmloc_gang_counters: null context");
  Ctx *c = reinterpret_cast<Ctx *>(lead_ctx);
  if (launches) *launches = c->gang_owned ? c->gang_owned->launches : 0;
  if (gang_launches) *gang_launches = c->gang_owned ? c->gang_owned->gang_launches : 0;
  return SFMLOC_OK;
}

int sfmloc_context_sync(sfmloc_context *ctx) {
  SFM_CHECK(ctx, SFMLOC_EINVAL, "sfmloc_context_sync: null context");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_HIP(hipSetDevice(c->map->device));
  SFM_HIP(hipStreamSynchronize(c->stream));
  c->stream.dirty = false;
  if (c->in_flight == nullptr) ctx_mark_idle(c);
  return SFMLOC_OK;
}

uint64_t sfmloc_packed_bytes(uint32_t n_queries, uint32_t budget) { return packed_part_bytes(n_queries, budget); }

int sfmloc_shard_export_packed(sfmloc_context *ctx, void *packed_dev, uint32_t n_queries, uint32_t budget,
                               uint32_t query_index) {
  SFM_CHECK(ctx && packed_dev && n_queries > 0 && query_index < n_queries, SFMLOC_EINVAL,
            "sfmloc_shard_export_packed: bad argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_HIP(hipSetDevice(c->map->device));
  return launch_export_packed(c, packed_dev, n_queries, budget, query_index);
}

static int merge_begin_impl(sfmloc_context *ctx, sfmloc_query *query, const void *parts_dev, uint32_t n_parts,
                            uint32_t cap, uint64_t part_stride, uint32_t packed_b, uint32_t packed_qi);

int sfmloc_merge_begin(sfmloc_context *ctx, sfmloc_query *query, const void *parts_dev, uint32_t n_parts,
                       uint32_t cap, uint64_t part_stride) {
  return merge_begin_impl(ctx, query, parts_dev, n_parts, cap, part_stride, 0, 0);
}

int sfmloc_merge_begin_packed(sfmloc_context *ctx, sfmloc_query *query, const void *packed_dev, uint32_t n_parts,
                              uint64_t part_stride, uint32_t n_queries, uint32_t budget, uint32_t query_index) {
  SFM_CHECK(n_queries > 0 && query_index < n_queries && budget > 0, SFMLOC_EINVAL,
            "sfmloc_merge_begin_packed: bad argument");
  if (part_stride == 0) part_stride = packed_part_bytes(n_queries, budget);
  SFM_CHECK(part_stride >= packed_part_bytes(n_queries, budget), SFMLOC_EINVAL,
            "sfmloc_merge_begin_packed: part_stride too small");
  SFM_CHECK((uint64_t)n_parts * budget < (1ull << 32), SFMLOC_EINVAL, "sfmloc_merge_begin_packed: n_parts * budget >= 2^32");
  return merge_begin_impl(ctx, query, packed_dev, n_parts, budget, part_stride, n_queries, query_index);
}

static int merge_begin_impl(sfmloc_context *ctx, sfmloc_query *query, const void *parts_dev, uint32_t n_parts,
                            uint32_t cap, uint64_t part_stride, uint32_t packed_b, uint32_t packed_qi) {
  SFM_CHECK(ctx && query && parts_dev && n_parts > 0 && cap > 0, SFMLOC_EINVAL, "sfmloc_merge_begin: bad argument");
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  Query *q = reinterpret_cast<Query *>(query);
  Map *m = c->map;
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_merge_begin: query belongs to another map");
  SFM_CHECK(c->in_flight == nullptr, SFMLOC_EINVAL, "sfmloc_merge_begin: context has a query in flight");
  SFM_CHECK(m->focal > 0.0, SFMLOC_EINVAL, "sfmloc_merge_begin: the map has no intrinsic");
  SFM_CHECK(q->n == 0 || q->d_kpt, SFMLOC_EINVAL, "sfmloc_merge_begin: the query was created without keypoints");
  if (part_stride == 0) part_stride = sfmloc_part_bytes(cap);
  SFM_CHECK(packed_b || part_stride >= sfmloc_part_bytes(cap), SFMLOC_EINVAL, "sfmloc_merge_begin: part_stride too small");
  SFM_HIP(hipSetDevice(m->device));
  c->t_begin = now_s();
  // (counted like a begun localisation: with other contexts' work queued -- the next batch's stage 1, always, in the
  // pipeline of dist.py -- K5 takes its shared-GPU round sizes; sfmloc_localize_end uncounts)
  ctx_mark_busy(c);
  {
    const int rcr = ctx_p3p_reserve(c, q->n);
    if (rcr) {
      ctx_mark_idle(c);
      return rcr;
    }
  }
  int rc;
  {
    EventScope ev(c, SFMLOC_K_MATCHSET);
    rc = launch_select_candidates(c, q, reinterpret_cast<const unsigned char *>(parts_dev), n_parts, part_stride, cap,
                                  packed_b, packed_qi, /*reset_status=*/true);
  }
  if (!rc) {
    EventScope ev(c, SFMLOC_K_P3P);
    rc = ctx_resection_enqueue(c, true);
  }
  if (!rc) rc = ctx_fetch_result(c);
  if (rc) {
    ctx_mark_idle(c);
    return rc;
  }
  c->in_flight = q;
  return SFMLOC_OK;
}

int sfmloc_localize(sfmloc_map *map, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel,
                    sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark, uint32_t cap) {
  SFM_CHECK(map && query && out, SFMLOC_EINVAL, "sfmloc_localize: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Query *q = reinterpret_cast<Query *>(query);
  SFM_CHECK(q->map == m, SFMLOC_EINVAL, "sfmloc_localize: query belongs to another map");
  SFM_HIP(hipSetDevice(m->device));
  int rc = ctx_localize_begin(m->ctx0, q, view_sel, n_sel);
  if (rc) {
    m->ctx0->in_flight = nullptr;
    return rc;
  }
  return ctx_localize_end(m->ctx0, out, pair_qfeat, pair_landmark, cap);
}

int sfmloc_localize_bow(sfmloc_map *map, sfmloc_query *query, const float *query_bow, uint32_t knn,
                        const uint32_t *cand_views, uint32_t n_cand, sfmloc_pose *out, uint32_t *pair_qfeat,
                        uint32_t *pair_landmark, uint32_t cap) {
  SFM_CHECK(map && query && out, SFMLOC_EINVAL, "sfmloc_localize_bow: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  int rc = sfmloc_localize_bow_begin(reinterpret_cast<sfmloc_context *>(m->ctx0), query, query_bow, knn, cand_views, n_cand);
  if (rc) {
    m->ctx0->in_flight = nullptr;
    return rc;
  }
  return ctx_localize_end(m->ctx0, out, pair_qfeat, pair_landmark, cap);
}

int sfmloc_localize_batch(sfmloc_map *map, sfmloc_query *const *queries, uint32_t n, uint32_t n_contexts,
                          sfmloc_pose *poses, uint32_t *pair_qfeat, uint32_t *pair_landmark, uint32_t pair_stride) {
  SFM_CHECK(map && (n == 0 || (queries && poses)), SFMLOC_EINVAL, "sfmloc_localize_batch: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  if (n_contexts == 0) n_contexts = 4;
  if (n_contexts > n) n_contexts = n ? n : 1;
  while (m->batch_ctx.size() < n_contexts) {
    Ctx *c = nullptr;
    int rc = make_ctx(m, &c);
    if (rc) return rc;
    m->pool.push_back(c);
    m->batch_ctx.push_back(c);
    m->hbm_bytes += c->hbm_bytes;
  }
  int first_err = SFMLOC_OK;
  auto finish = [&](uint32_t i) {
    Ctx *c = m->batch_ctx[i % n_contexts];
    int rc = ctx_localize_end(c, &poses[i], pair_qfeat ? pair_qfeat + (size_t)i * pair_stride : nullptr,
                              pair_landmark ? pair_landmark + (size_t)i * pair_stride : nullptr, pair_stride);
    if (rc && !first_err) first_err = rc;
  };
  for (uint32_t i = 0; i < n; ++i) {
    if (i >= n_contexts) finish(i - n_contexts);
    Ctx *c = m->batch_ctx[i % n_contexts];
    Query *q = reinterpret_cast<Query *>(queries[i]);
    int rc = (q && q->map == m) ? ctx_localize_begin(c, q, nullptr, 0) : SFMLOC_EINVAL;
    if (rc) {
      if (!first_err) {
        first_err = rc;
        if (!(q && q->map == m)) set_error("sfmloc_localize_batch: query %u is null or belongs to another map", i);
      }
      // drain what is in flight and stop
      for (uint32_t k = (i >= n_contexts ? i - n_contexts + 1 : 0); k < i; ++k) finish(k);
      return first_err;
    }
  }
  for (uint32_t k = (n >= n_contexts ? n - n_contexts : 0); k < n; ++k) finish(k);
  return first_err;
}

// ----- BoW shortlist ---------------------------------------------------------------------------------------

int sfmloc_bow_select(sfmloc_map *map, const float *query_bow, const uint32_t *cand_views, uint32_t n_cand,
                      uint32_t k, uint32_t *out_sel, uint32_t *n_out) {
  SFM_CHECK(map && query_bow && out_sel && n_out, SFMLOC_EINVAL, "sfmloc_bow_select: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *c = m->ctx0;
  SFM_CHECK(m->bow_dim > 0 && m->d_bow, SFMLOC_EINVAL, "sfmloc_bow_select: the map has no .bow vectors");
  if (!cand_views) n_cand = m->n_views;
  SFM_CHECK(n_cand <= m->n_views, SFMLOC_EINVAL, "sfmloc_bow_select: n_cand > n_views");
  // CV_Assert(knn < viewList.size()) (BoFUtils.cpp:30)
  SFM_CHECK(k < n_cand, SFMLOC_EINVAL, "sfmloc_bow_select: knn %u must be smaller than the %u candidate views", k,
            n_cand);
  SFM_HIP(hipSetDevice(m->device));
  if (cand_views) {
    for (uint32_t i = 0; i < n_cand; ++i) {
      SFM_CHECK(cand_views[i] < m->n_views, SFMLOC_EINVAL, "sfmloc_bow_select: view index out of range");
      SFM_CHECK(i == 0 || cand_views[i - 1] < cand_views[i], SFMLOC_EINVAL,
                "sfmloc_bow_select: candidate views must be strictly ascending");
    }
    SFM_HIP(hipMemcpyAsync(c->d_bow_cand, cand_views, n_cand * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  SFM_HIP(hipMemcpyAsync(c->d_bow_query, query_bow, m->bow_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
  int rc;
  {
    EventScope ev(c, SFMLOC_K_BOW);
    rc = launch_bow_select(c, c->d_bow_query, cand_views ? c->d_bow_cand : nullptr, n_cand, k,
                           c->d_bow_dist, c->d_bow_sel);
  }
  if (rc) return rc;
  SFM_HIP(hipMemcpyAsync(out_sel, c->d_bow_sel, k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  SFM_HIP(hipStreamSynchronize(c->stream));
  *n_out = k;
  return SFMLOC_OK;
}

int sfmloc_bow_distances(sfmloc_map *map, const float *query_bow, float *out_dist) {
  SFM_CHECK(map && query_bow && out_dist, SFMLOC_EINVAL, "sfmloc_bow_distances: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  Ctx *c = m->ctx0;
  SFM_CHECK(m->bow_dim > 0 && m->d_bow, SFMLOC_EINVAL, "sfmloc_bow_distances: the map has no .bow vectors");
  SFM_HIP(hipSetDevice(m->device));
  if (m->n_views == 0) return SFMLOC_OK;
  SFM_HIP(hipMemcpyAsync(c->d_bow_query, query_bow, m->bow_dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
  int rc;
  {
    EventScope ev(c, SFMLOC_K_BOW);
    rc = launch_bow_select(c, c->d_bow_query, nullptr, m->n_views, 1, c->d_bow_dist, c->d_bow_sel);
  }
  if (rc) return rc;
  // the kernel stores the float32 distance's bit pattern (non-negative floats order like their bits)
  SFM_HIP(hipMemcpyAsync(out_dist, c->d_bow_dist, (size_t)m->n_views * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  SFM_HIP(hipStreamSynchronize(c->stream));
  return SFMLOC_OK;
}

int sfmloc_bof_create(const sfmloc_bof_desc *d, int device, sfmloc_bof **out) {
  SFM_CHECK(d && out, SFMLOC_EINVAL, "sfmloc_bof_create: null argument");
  *out = nullptr;
  SFM_CHECK(d->K > 0 && d->centers && d->in_dim > 0 && d->in_dim <= 128, SFMLOC_EINVAL,
            "sfmloc_bof_create: bad model (K %d, in_dim %d)", d->K, d->in_dim);
  SFM_CHECK(d->pyramid_level >= 1 && d->pyramid_level <= 3, SFMLOC_EINVAL, "PYRAMID_LEVEL must be 1..3");
  SFM_CHECK(d->n_pca == 0 || (d->pca_mean && d->pca_eigvec && d->pca_eigval && d->n_pca <= 128), SFMLOC_EINVAL,
            "sfmloc_bof_create: PCA arrays missing");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  SFM_CHECK(e == hipSuccess && ndev > 0, SFMLOC_ENODEV, "no HIP device visible; this library has no CPU fallback");
  SFM_HIP(hipSetDevice(device));
  BofModel *b = new (std::nothrow) BofModel();
  SFM_CHECK(b, SFMLOC_ENOMEM, "out of host memory");
  b->device = device;
  b->K = d->K;
  b->in_dim = d->in_dim;
  b->n_pca = d->n_pca;
  b->cdim = d->n_pca > 0 ? d->n_pca : d->in_dim;
  b->resized = d->resized_image_size;
  b->levels = d->use_spatial_pyramid ? d->pyramid_level : 1;
  b->norm_type = d->norm_type;
  b->cells = 0;
  for (int l = 0; l < b->levels; ++l) b->cells += (l == 0) ? 1 : (l == 2 ? 3 : (l + 1) * (l + 1));
  uint64_t acct = 0;
  int rc = SFMLOC_OK;
  // (the model's own stream is created by the first sfmloc_bof_compute: a model inside a sfmloc_imgbow works on that
  // object's stream and needs none -- an idle stream still shares a hardware queue with streams that work; the uploads
  // here go through the null stream)
  if (!rc) rc = dev_upload(&acct, &b->d_centers, d->centers, (size_t)d->K * b->cdim, b->stream);
  if (!rc && d->n_pca > 0) {
    rc = dev_upload(&acct, &b->d_pca_mean, d->pca_mean, (size_t)d->in_dim, b->stream);
    if (!rc) rc = dev_upload(&acct, &b->d_pca_evec, d->pca_eigvec, (size_t)d->n_pca * d->in_dim, b->stream);
    if (!rc) rc = dev_upload(&acct, &b->d_pca_eval, d->pca_eigval, (size_t)d->n_pca, b->stream);
  }
  if (!rc) rc = dev_alloc(&acct, &b->d_counts, (size_t)b->K * b->cells);
  if (!rc) rc = dev_alloc(&acct, &b->d_out, (size_t)b->K * b->cells);
  if (!rc && hipStreamSynchronize(b->stream) != hipSuccess) rc = SFMLOC_EHIP;
  if (rc) {
    sfmloc_bof_destroy(reinterpret_cast<sfmloc_bof *>(b));
    return rc;
  }
  *out = reinterpret_cast<sfmloc_bof *>(b);
  return SFMLOC_OK;
}

void sfmloc_bof_destroy(sfmloc_bof *bof) {
  BofModel *b = reinterpret_cast<BofModel *>(bof);
  if (!b) return;
  hipSetDevice(b->device);
  if (b->stream) hipStreamSynchronize(b->stream);
  void *ptrs[] = {b->d_centers, b->d_pca_mean, b->d_pca_evec, b->d_pca_eval, b->d_counts, b->d_out, b->d_desc, b->d_kxy};
  for (void *p : ptrs)
    if (p) hipFree(p);
  if (b->stream) hipStreamDestroy(b->stream);
  delete b;
}

int sfmloc_bof_dim(const sfmloc_bof *bof) {
  const BofModel *b = reinterpret_cast<const BofModel *>(bof);
  return b ? b->K * b->cells : 0;
}

int sfmloc_bof_compute(sfmloc_bof *bof, const float *desc, const float *kpt_xy, uint32_t n, double *out_bow) {
  SFM_CHECK(bof && out_bow && (n == 0 || (desc && kpt_xy)), SFMLOC_EINVAL, "sfmloc_bof_compute: null argument");
  BofModel *b = reinterpret_cast<BofModel *>(bof);
  SFM_HIP(hipSetDevice(b->device));
  if ((int)n > b->cap_n) {
    if (b->d_desc) hipFree(b->d_desc);
    if (b->d_kxy) hipFree(b->d_kxy);
    b->d_desc = nullptr;
    b->d_kxy = nullptr;
    SFM_HIP(hipMalloc((void **)&b->d_desc, (size_t)n * b->in_dim * sizeof(float)));
    SFM_HIP(hipMalloc((void **)&b->d_kxy, (size_t)n * 2 * sizeof(float)));
    b->cap_n = (int)n;
  }
  if (!b->stream) SFM_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  if (n) {
    SFM_HIP(hipMemcpyAsync(b->d_desc, desc, (size_t)n * b->in_dim * sizeof(float), hipMemcpyHostToDevice, b->stream));
    SFM_HIP(hipMemcpyAsync(b->d_kxy, kpt_xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, b->stream));
  }
  int rc = launch_bof(b, b->stream, b->d_desc, b->d_kxy, (int)n, b->d_counts, b->d_out, nullptr);  // (stream: created above)
  if (rc) return rc;
  SFM_HIP(hipMemcpyAsync(out_bow, b->d_out, (size_t)b->K * b->cells * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  SFM_HIP(hipStreamSynchronize(b->stream));
  return SFMLOC_OK;
}

// ----- a rank's whole batch per call --------------------------------------------
// dist.py drove a batch with one ctypes call per query and stage (~2 ms of a 17 ms batch at N = 8); these take the
// batch's arrays and run the same per-query calls in gang sessions of `gang` contexts: query i goes to context
// i mod n_ctx, the contexts [g, g + gang) of every n_ctx consecutive queries form one session (the first of them leads),
// the sessions take turns so that their streams stay equally loaded -- exactly the order dist.HipShardCompute._rounds
// queued them in, so the results are the same bits.
}  // extern "C"
namespace {
template <class PerQuery>
int shard_batch_rounds(sfmloc_context *const *ctxs, uint32_t n_ctx, uint32_t gang, uint32_t n_queries, const char *who,
                       PerQuery per_query) {
  SFM_CHECK(ctxs && n_ctx > 0, SFMLOC_EINVAL, "%s: no contexts", who);
  if (gang == 0) gang = 1;
  if (gang > (uint32_t)kGangMembers) gang = kGangMembers;
  for (uint32_t base = 0; base < n_queries; base += n_ctx)
    for (uint32_t g0 = 0; g0 < n_ctx; g0 += gang) {
      uint32_t m = 0;
      for (uint32_t k = g0; k < n_ctx && k < g0 + gang && base + k < n_queries; ++k) ++m;
      if (m == 0) continue;
      int rc = m > 1 ? sfmloc_gang_begin(ctxs + g0, m) : SFMLOC_OK;
      for (uint32_t k = 0; k < m && rc == SFMLOC_OK; ++k) rc = per_query(ctxs[g0 + k], base + g0 + k);
      const int rc_end = m > 1 ? sfmloc_gang_end(ctxs + g0, m) : SFMLOC_OK;
      if (rc == SFMLOC_OK) rc = rc_end;
      if (rc) return rc;
    }
  return SFMLOC_OK;
}
}  // namespace
extern "C" {

int sfmloc_shard_batch_bow_keys(sfmloc_context *const *ctxs, uint32_t n_ctx, uint32_t gang, sfmloc_query *const *queries,
                                uint32_t n_queries, uint32_t knn, void *keys_dev) {
  SFM_CHECK(queries && keys_dev && knn > 0, SFMLOC_EINVAL, "sfmloc_shard_batch_bow_keys: bad argument");
  return shard_batch_rounds(ctxs, n_ctx, gang, n_queries, "sfmloc_shard_batch_bow_keys", [&](sfmloc_context *c, uint32_t i) {
    return sfmloc_shard_bow_keys(c, queries[i], nullptr, knn, static_cast<unsigned char *>(keys_dev) + (size_t)i * knn * 8);
  });
}

int sfmloc_shard_batch_begin_bow(sfmloc_context *const *ctxs, uint32_t n_ctx, uint32_t gang, sfmloc_query *const *queries,
                                 uint32_t n_queries, const void *keys_all_dev, uint32_t n_parts, uint32_t knn,
                                 void *packed_dev, uint32_t budget) {
  SFM_CHECK(queries && keys_all_dev && packed_dev && knn > 0 && n_parts > 0, SFMLOC_EINVAL,
            "sfmloc_shard_batch_begin_bow: bad argument");
  // keys_all [n_parts][n_queries][knn]: query i's lists start at i * knn keys, one part every n_queries * knn keys
  return shard_batch_rounds(ctxs, n_ctx, gang, n_queries, "sfmloc_shard_batch_begin_bow", [&](sfmloc_context *c, uint32_t i) {
    int rc = sfmloc_shard_begin_bow(c, queries[i], static_cast<const unsigned char *>(keys_all_dev) + (size_t)i * knn * 8,
                                    n_parts, (uint64_t)n_queries * knn, knn);
    if (!rc) rc = sfmloc_shard_export_packed(c, packed_dev, n_queries, budget, i);
    return rc;
  });
}

int sfmloc_shard_batch_begin(sfmloc_context *const *ctxs, uint32_t n_ctx, uint32_t gang, sfmloc_query *const *queries,
                             uint32_t n_queries, void *packed_dev, uint32_t budget) {
  SFM_CHECK(queries && packed_dev, SFMLOC_EINVAL, "sfmloc_shard_batch_begin: bad argument");
  return shard_batch_rounds(ctxs, n_ctx, gang, n_queries, "sfmloc_shard_batch_begin", [&](sfmloc_context *c, uint32_t i) {
    int rc = sfmloc_shard_begin(c, queries[i], nullptr, 0);
    if (!rc) rc = sfmloc_shard_export_packed(c, packed_dev, n_queries, budget, i);
    return rc;
  });
}

// stage 2 of up to kGangMembers queries in ONE session: context k takes query query_index[k] of the batch
int sfmloc_merge_batch_begin(sfmloc_context *const *ctxs, uint32_t n, sfmloc_query *const *queries, const uint32_t *query_index,
                             const void *packed_all_dev, uint32_t n_parts, uint64_t part_stride, uint32_t n_queries,
                             uint32_t budget) {
  SFM_CHECK(ctxs && queries && query_index && packed_all_dev && n > 0 && n <= (uint32_t)kGangMembers, SFMLOC_EINVAL,
            "sfmloc_merge_batch_begin: 1..%d contexts", kGangMembers);
  int rc = n > 1 ? sfmloc_gang_begin(ctxs, n) : SFMLOC_OK;
  for (uint32_t k = 0; k < n && rc == SFMLOC_OK; ++k)
    rc = sfmloc_merge_begin_packed(ctxs[k], queries[k], packed_all_dev, n_parts, part_stride, n_queries, budget, query_index[k]);
  const int rc_end = n > 1 ? sfmloc_gang_end(ctxs, n) : SFMLOC_OK;
  return rc ? rc : rc_end;
}

void sfmloc_debug_fail_p3p_alloc(int k) { sfmloc::g_test_fail_alloc.store(k, std::memory_order_relaxed); }

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
  SFM_HIP(hipStreamSynchronize(m->ctx0->stream));
  for (Ctx *c : m->pool) SFM_HIP(hipStreamSynchronize(c->stream));
  return SFMLOC_OK;
}

int sfmloc_stats_read(sfmloc_map *map, sfmloc_kernel_stats *out) {
  SFM_CHECK(map && out, SFMLOC_EINVAL, "sfmloc_stats_read: null argument");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  memset(out, 0, sizeof(*out));
  std::vector<Ctx *> all = m->pool;
  all.push_back(m->ctx0);
  for (Ctx *c : all) {
    SFM_HIP(hipStreamSynchronize(c->stream));
    int rc = drain_events(c);
    if (rc) return rc;
    for (int k = 0; k < SFMLOC_K_COUNT; ++k) {
      out->total_ms[k] += c->stats.total_ms[k];
      out->launches[k] += c->stats.launches[k];
    }
    out->hamming_pairs += c->stats.hamming_pairs;
    out->hamming_alg_bytes += c->stats.hamming_alg_bytes;
    unsigned long long k1s[2 * kK1CounterSlots], k1c[2] = {0, 0};
    SFM_HIP(hipMemcpy(k1s, c->d_k1_counters, sizeof(k1s), hipMemcpyDeviceToHost));
    for (int q = 0; q < kK1CounterSlots; ++q) {
      k1c[0] += k1s[2 * q];
      k1c[1] += k1s[2 * q + 1];
    }
    out->hamming_lane_ops += c->stats.hamming_lane_ops + 64ull * k1c[0] * (uint64_t)c->k1_finish_ops;
    out->hamming_pairs_finished += 64ull * k1c[0];
    out->hamming_rows_flagged += k1c[1];
  }
  return SFMLOC_OK;
}

int sfmloc_set_profile(sfmloc_map *map, int level) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_set_profile: null map");
  SFM_CHECK(level >= 0 && level <= 2, SFMLOC_EINVAL, "sfmloc_set_profile: level %d (0 off, 1 every stage, 2 K1 only)", level);
  reinterpret_cast<Map *>(map)->params.profile = level;
  return SFMLOC_OK;
}

int sfmloc_stats_reset(sfmloc_map *map) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_stats_reset: null map");
  Map *m = reinterpret_cast<Map *>(map);
  SFM_HIP(hipSetDevice(m->device));
  std::vector<Ctx *> all = m->pool;
  all.push_back(m->ctx0);
  for (Ctx *c : all) {
    SFM_HIP(hipStreamSynchronize(c->stream));
    int rc = drain_events(c);
    if (rc) return rc;
    memset(&c->stats, 0, sizeof(c->stats));
    SFM_HIP(hipMemset(c->d_k1_counters, 0, 2 * kK1CounterSlots * sizeof(unsigned long long)));
  }
  return SFMLOC_OK;
}

}  // extern "C"
