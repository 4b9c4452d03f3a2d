// Internal declarations shared by the HIP translation units of libsfmloc_hip.so.
// Nothing here is part of the C ABI (include/sfmloc.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/sfmloc.h"
#include "chain_device.h"
#include "gang.h"

namespace sfmloc {

void set_error(const char *fmt, ...);

#define SFM_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      ::sfmloc::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? SFMLOC_ENODEV           \
             : (e_ == hipErrorOutOfMemory)                            ? SFMLOC_ENOMEM          \
                                                                      : SFMLOC_EHIP;           \
    }                                                                                          \
  } while (0)

#define SFM_CHECK(cond, code, ...)     \
  do {                                 \
    if (!(cond)) {                     \
      ::sfmloc::set_error(__VA_ARGS__); \
      return (code);                   \
    }                                  \
  } while (0)

// ---------------------------------------------------------------------------
// HBM layout of the descriptor bank ("tiled64").
// The .desc payload is row major, 64 B per descriptor.  A lane that owns one bank
// row needs all 64 B of it in registers; loading that straight from the row-major
// image makes every wave-instruction touch 64 different 64-B segments.  The bank is
// therefore re-tiled ONCE at map load: rows are grouped in blocks of 64 (one per
// lane of a wavefront) and each block is stored as 4 planes of 64 x 16 B:
//     tiled[(block*4 + plane)*64 + lane] = row(block*64 + lane).uint4[plane]
// so that the four global_load_dwordx4 a wave issues for a block are each one
// contiguous, fully coalesced 1 KiB.
// ---------------------------------------------------------------------------
constexpr uint32_t kBlockRows = 64;
constexpr int kK1CounterSlots = 64;  // Ctx::d_k1_counters
constexpr int kP3pMaxN = 4096;     // 2D-3D correspondences the P3P LDS sort holds
constexpr int kP3pBatchMax = 512;  // hypotheses evaluated per round
// result slots of a round: one per hypothesis (at most kP3pBatchMax), or (wide launches, at most kP3pSlots / 4 = 128
// hypotheses -- their nominal size since round 4) four, one per model.  A slot's inlier list is kP3pMaxN ints: 8 MB per
// context (16 MB while wide rounds could be 256 hypotheses: ADVICE r03)
constexpr int kP3pSlots = 512;
constexpr uint32_t kPartHeaderBytes = 16;  // candidate part: {u32 n_cand, pad[3]} then the candidates

using Pose = sfmloc_pose;

// one 2D-3D candidate (K4); `order` sorts candidates of one query feature exactly as the reference's
// sequential scan would meet them (distance, then view id, then position in the view's list)
struct Candidate {
  uint64_t order;
  uint32_t qfeat;
  uint32_t landmark_id;
  double X[3];
};

struct P3pState {
  int n, iter, n_iter, n_reserve, n_index, identity, n_in, done, rounds, status;
  unsigned arrive;  // workgroups of the current round that have delivered their hypothesis (k_p3p_round)
  int finished;     // the epilogue (pose, inlier pairs) has run (k_p3p_finish)
  int batch_limit;  // hypotheses the current round evaluates (<= the launch's grid); see p3p_next_batch_limit
  int switch_iter;  // iteration at which sampling switched to the best model's inliers
  // hypotheses prep_iter .. prep_iter + prep_n - 1 have their P3P models in P3pArgs::prep_models already: the workgroup
  // that replayed the previous round solved them, one hypothesis per lane (acransac.hip "Prepared ahead")
  int prep_iter, prep_n;
  // the lowest hypothesis index of the current round whose model is known to change the index set (NFA below the round's
  // starting NFA and below 0), or ~0u: a workgroup that STARTS after that is known -- while the GPU is shared a round's
  // workgroups trickle in -- evaluates nothing, because the replay throws everything behind that hypothesis away
  unsigned first_hit;
  double min_nfa, errmax;
  double model[12];
};

struct HostResult;
struct P3pArgs {
  P3pState *state;
  Pose *result;
  const HostResult *record;  // the context's result record in device memory (state, pose, status, inlier pairs) ...
  HostResult *host;          // ... and its pinned host copy, which k_p3p_finish fills itself (no copy launch)
  const uint32_t *ms_n, *ms_qfeat, *ms_landmark;
  const double *pt2d, *pt3d;
  double *xn;
  const double *L10;
  float *logc_n, *logc_k;
  int32_t *vec_index, *best_inl;
  double *hyp_nfa;
  int *hyp_k;
  double *hyp_err, *hyp_model;
  int32_t *hyp_inl;
  double *prep_models;  // [kP3pBatchMax][4][12] models of the coming round's hypotheses ("Prepared ahead"), and how many
  int *prep_nm;         // roots each has; prep_ahead = 0: every workgroup solves its own hypothesis
  int prep_ahead;
  uint32_t *pair_qfeat, *pair_landmark, *inlier_idx;
  uint64_t *ws_key;  // [kP3pLargeBatch * max_n] sort segments for more than kP3pMaxN correspondences, or null
  uint32_t *ws_idx;
  double *ws_terms;  // [max_n / 2 + 1]
  double focal, ppx, ppy;
  int max_iteration, min_resection_points, min_inliers, max_n, refine_pose;
  int nfa_filter;      // 1 = models that cannot beat the round's starting NFA are not sorted (acransac.hip, "The NFA filter")
  int nfa_filter_min_p;  // ... for next_pow2(n) >= this
  int skip_overtaken;  // 1: workgroups / waves give up behind a known index-changing hypothesis (P3pState::first_hit)
  int adaptive_batch;  // other queries share the GPU: trade rounds for fewer speculative hypotheses
  int adapt_quarters, adapt_floor;  // next batch = max(floor, quarters/4 * iterations since the switch)
  uint64_t seed;
  uint32_t stream;
};

// what a finished query copies back in one asynchronous transfer
struct HostResult {
  P3pState state;
  Pose pose;
  int status;
  uint32_t view_stats[3];  // views with >= min_putative matches, views passing the F filter, the largest view above 512
  uint32_t pair_qfeat[kP3pMaxN];
  uint32_t pair_landmark[kP3pMaxN];
};

struct Ctx;

// Static, read-only after creation: the map as it sits in HBM.
struct Map {

  int device = 0;
  int n_cu = 256;
  sfmloc_params params{};

  uint64_t n_rows = 0;
  uint32_t n_blocks = 0;  // ceil(n_rows / 64)
  uint32_t max_view_blocks = 0;  // most 64-row blocks any one view overlaps (launch bound of a device-side selection)
  uint32_t max_view_rows = 0;    // most rows of any one view
  std::atomic<int> busy_ctx{0};  // contexts with work queued (begin .. end / sync): K1 slices a short scan only when alone
  // > 0 while recent queries had more than 512 2D-3D correspondences: K5's rounds are then launched wide (four
  // workgroups per hypothesis, acransac.hip).  Set to 64 by a finished query that had, counted down by the others.
  std::atomic<int> p3p_wide_credit{0};
  std::atomic<int> k3_huge_credit{0};  // ... with 1 025 .. 2 048 (the wide form's 2 048-match instance, a query alone)
  std::atomic<int> k3_big_credit{0};  // finished queries ago that one had a view with more than 512 putative matches (64 = just now)
  // finished queries in a row whose 2D-3D set had at most 512 correspondences (acransac.hip kP3pSmallN): from 8 on a
  // query's P3P rounds are queued in the small form (ctx_resection_enqueue)
  std::atomic<int> p3p_small_credit{0};
  uint32_t n_views = 0;
  uint32_t n_landmarks = 0;
  std::vector<uint32_t> h_view_id, h_view_off, h_view_wh;
  std::vector<double> h_view_center;     // [n_views*3], maps opened from sfm_data.json only
  std::vector<std::string> h_view_file;  // view filenames, same
  uint4 *d_bank = nullptr;         // tiled64, n_blocks*64 rows (zero padded)
  // The view tables carry one extra, EMPTY view at index n_views ("phantom": no rows, id 0xFFFFFF): a device-built
  // selection whose length the host cannot know (the sharded BoW shortlist) is padded with it, and every kernel
  // treats it as a view without descriptors.  Per-view arrays of a context are sized n_views + 1 for the same reason.
  uint32_t *d_view_off = nullptr;  // [n_views+2]  (view_off[n_views+1] = view_off[n_views] = n_rows)
  uint32_t *d_view_id = nullptr;   // [n_views+1]
  uint32_t *d_view_wh = nullptr;   // [(n_views+1)*2]
  float2 *d_kpt = nullptr;         // [n_rows]
  int32_t *d_row_landmark = nullptr;
  uint32_t *d_landmark_id = nullptr;
  double *d_landmark_X = nullptr;
  double focal = 0, ppx = 0, ppy = 0, k1 = 0, k2 = 0, k3 = 0;
  uint32_t intrinsic_type = 0;  // 0 pinhole, 3 pinhole_radial_k3
  uint32_t bow_dim = 0;
  float *d_bow = nullptr;
  double *d_L10 = nullptr;          // [65538] log10(i)
  uint16_t *d_ratio_cnt = nullptr;  // [513]
  float ratio_cnt_for = -1.0f;      // ratio the table was built for
  bool have_geometry = false;       // kpts + landmarks + view sizes + intrinsic were supplied
  uint64_t hbm_bytes = 0;

  Ctx *ctx0 = nullptr;              // the handle's own context (stage-level API)
  std::vector<Ctx *> pool;          // every other context of this map (owned by the map)
  std::vector<Ctx *> batch_ctx;     // the subset sfmloc_localize_batch round-robins over
};

// Everything one in-flight query needs: a stream and the workspace of every stage.  Several contexts
// of one map run concurrently (the latency-bound RANSAC stages of one query overlap the VALU-bound
// Hamming kernel of another).
struct Ctx : GangMember {  // (gang.h: stream, gang_recs, gang_head)
  Map *map = nullptr;
  bool stream_borrowed = false;    // stream.own belongs to `lender` (sfmloc_context_create_sharing)
  Ctx *lender = nullptr;
  bool merge_only = false;         // no workspace of the matching stages (sfmloc_context_create_merge)
  int borrowers = 0;               // contexts working on this one's stream
  bool zombie = false;             // destroyed while lent out: freed with its last borrower
  uint64_t hbm_bytes = 0;

  // --- putative stage ---
  uint32_t max_split = 8;
  uint2 *d_part = nullptr;          // [n_blocks*64] partial (best0,best1), laid out [split][work block][lane]
  uint32_t *d_view_sel = nullptr;   // [n_views] selected view indices
  uint32_t *d_view_widx0 = nullptr; // [n_views] work-block index of each selected view's first bank block
  uint32_t *d_block_list = nullptr; // [n_blocks]
  uint32_t *h_pinned = nullptr;     // pinned staging: view_sel | view_widx0 | block_list
  uint32_t *d_view_count = nullptr; // [n_views]
  uint32_t *d_match_i = nullptr;    // [n_rows]
  uint32_t *d_match_key = nullptr;  // [n_rows]  (d0<<16)|j0
  uint2 *d_flagged = nullptr;       // [n_blocks*64] rows the screening kernel could not reject {part index, bank row}
  uint32_t *d_n_flagged = nullptr;
  unsigned long long *d_flagmask = nullptr;  // [n_blocks] per work block: rows whose d_part slot is valid (screened scan)
  bool last_screened = false;
  uint2 *d_rows_scratch = nullptr;      // [rows_chunk_cap][8 slices][64] partial top-2 of the flagged-row pass
  uint32_t *d_rows_arrivals = nullptr;  // [rows_chunk_cap] arrival counters of that pass (self-resetting)
  uint4 *d_flagged_desc = nullptr;      // [rows_chunk_cap][4][64] descriptors of the flagged rows (tiled like the bank)
  uint32_t rows_chunk_cap = 0;          // chunks of 64 flagged rows the sliced pass has scratch for (rest: fallback)
  // [kK1CounterSlots][2] finished wave-pairs, flagged rows (since stats reset): a wave adds to slot (its block & 63) -- one
  // pair of words for every wave of a scan was 300 000 atomics on the same address per full-bank scan
  unsigned long long *d_k1_counters = nullptr;
  int k1_finish_ops = 0;

  // --- geometric stages ---
  uint32_t *d_geo_count = nullptr;  // [n_views]
  uint32_t *d_geo_idx = nullptr;    // [n_rows]
  double *d_geo_model = nullptr;    // [(n_views+1)*10] per view: AC-RANSAC's F (normalised frame, 9) + errorMax
  // guided matching (-gm): allocated on first use
  uint32_t *d_geo_j = nullptr;        // [n_rows] query feature of each guided match (geo_idx then holds the map feature)
  uint32_t *d_guided_row = nullptr;   // [n_rows] per bank row: its guided query feature or SFMLOC_NOMATCH
  bool geo_is_pairs = false;          // the last geometric stage left (i, j) lists, not indices into the putative lists
  // K3 for views with more putative matches than its LDS form holds (acransac.hip k_fmatrix_large): allocated on first use
  uint64_t *fl_key = nullptr;
  uint32_t *fl_idx = nullptr, *fl_count = nullptr, *fl_list = nullptr;
  void *d_k3_static = nullptr;          // K3's per-context argument block on the device (FFilterStatic, acransac.hip) ...
  unsigned char k3_static_host[256] = {};  // ... and what was last written there
  bool k3_static_valid = false;
  void *d_k3_spec = nullptr;            // k_fmatrix_fast's wide form: the first batch's results per view slot (lazily)
  unsigned int *d_k3_arrive = nullptr;  // ... and the arrivals
  int32_t *fl_vec_index = nullptr, *fl_best_inl = nullptr;
  float *fl_logc_n = nullptr, *fl_logc_k = nullptr;
  int fl_slot_m = 0;
  int *d_status = nullptr;
  unsigned char *d_cand_part = nullptr;  // this context's candidate part: header + cand_cap candidates (one per query
                                         // feature at most, so 2^16 can never overflow)
  uint32_t cand_cap = 1u << 16;
  uint16_t *d_geo_dist = nullptr;        // [n_rows] per geometric match: its featDist, or 0xFFFF = not a candidate
  unsigned long long *d_best64 = nullptr;  // [65536]
  uint32_t *d_winner = nullptr;            // [65536]
  uint32_t *d_ms_n = nullptr, *d_ms_qfeat = nullptr, *d_ms_landmark = nullptr;  // [65536]
  double *d_pt2d = nullptr, *d_pt3d = nullptr;                                  // [65536*2], [65536*3]
  double *d_xn = nullptr;                                                       // [kP3pMaxN*2]
  float *d_logc_n = nullptr, *d_logc_k = nullptr;                               // [kP3pMaxN+1]
  int32_t *d_vec_index = nullptr, *d_best_inl = nullptr;                        // [kP3pMaxN]
  double *d_hyp_nfa = nullptr, *d_hyp_err = nullptr, *d_hyp_model = nullptr, *d_prep_models = nullptr;
  int *d_prep_nm = nullptr;
  int *d_hyp_k = nullptr;
  int32_t *d_hyp_inl = nullptr;  // [kP3pBatchMax * kP3pMaxN]
  uint32_t *d_pair_qfeat = nullptr, *d_pair_landmark = nullptr, *d_inlier_idx = nullptr;  // [p3p_cap]
  // the P3P arrays hold p3p_cap correspondences: kP3pMaxN to begin with, regrown (ctx_p3p_reserve) when a query with
  // more features arrives; beyond kP3pMaxN the pair lists leave HostResult for buffers of their own and the sort of a
  // hypothesis gets a global-memory segment
  uint32_t p3p_cap = kP3pMaxN;
  uint64_t p3p_bytes = 0;  // bytes of the regrowable P3P arrays currently held (part of hbm_bytes)
  bool p3p_small = false;  // this query's P3P rounds go out in the small form (decided when the first ones are queued)
  bool p3p_seq = false;    // this query's AC-RANSAC went out in the sequential form (one launch; acransac.hip k_p3p_seq)
  uint32_t p3p_query_n = 0;  // features of the query K5 is about to run for (ctx_p3p_reserve): launch shape of the rounds
  uint32_t *d_pair_qfeat_big = nullptr, *d_pair_landmark_big = nullptr;
  uint64_t *d_p3p_ws_key = nullptr;
  uint32_t *d_p3p_ws_idx = nullptr;
  double *d_p3p_terms = nullptr;
  P3pState *d_p3p_state = nullptr;
  Pose *d_pose = nullptr;
  unsigned char *d_result = nullptr;  // one HostResult record; the six pointers below alias into it
  uint32_t *d_view_stats = nullptr;  // [3] HostResult::view_stats
  float *d_bow_query = nullptr;      // [bow_dim]
  uint32_t *d_bow_dist = nullptr;    // [n_views]
  uint32_t *d_bow_cand = nullptr;    // [n_views]
  uint32_t *d_bow_sel = nullptr;     // [n_views]
  void *h_result = nullptr;  // pinned: what a finished query copies back in one go (capi.hip HostResult)
  hipEvent_t pinned_busy = nullptr;  // recorded after the last upload out of h_pinned
  hipEvent_t xev_out = nullptr, xev_in = nullptr;  // ordering against a caller's stream (sfmloc_context_signal / _wait)

  // state of the last putative call
  uint32_t last_split = 0;
  uint32_t last_nq = 0;
  uint32_t last_n_sel = 0;
  bool last_all_views = false;
  uint32_t last_n_work_blocks = 0;
  std::vector<uint32_t> last_blocks;  // host copy of the block list (empty = all)
  bool last_blocks_on_device = false; // the list was built by k_blocks_from_views: fetch it when a reader needs it
  bool flagmask_zeroed = false;       // k_blocks_from_views has just cleared d_flagmask for the coming scan
  bool counted_busy = false;          // this context is counted in Map::busy_ctx
  bool k1_may_slice = true;           // no other context had work queued when this query began
  bool p3p_init_fused = false;        // k_match_set_finish has run K5's initialisation for the query in hand
  bool chain_done = false;            // the shortlist kernel already cleared the counters and built the block list
  bool defer_merge = false;           // the caller runs K3 right after K1 on this context: K2 may be left to K3
  bool merge_is_deferred = false;     // ... and was: launch_fmatrix_filter passes deferred_merge to k_fmatrix_fast
  MergeMaskedArgs deferred_merge{};
  bool cleared = false;  // k_query_reset already cleared this query's counters: the stages skip their own memsets
  struct Query *last_query = nullptr;  // query of the last putative call
  struct Query *in_flight = nullptr;   // query of a begun, not yet ended, localisation
  double t_begin = 0.0;

  // --- measurement ---
  sfmloc_kernel_stats stats{};
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
};

// A launch of a kernel that has a gang form (Body: the kernel's body, gang.h): issued at once on the context's stream,
// or recorded when the context is a member of a gang session.  `single` is the kernel itself.
template <class... Ts, size_t... Is, class... As>
inline void gang_store_args(void *dst, std::index_sequence<Is...>, As &&...as) {
  new (dst) Tup<Ts...>{TupLeaf<Is, Ts>{static_cast<Ts>(as)}...};
}
template <class Body, class... Ts, class... As>
inline void sfm_launch(GangMember *c, void (*single)(Ts...), dim3 grid, dim3 block, uint32_t shmem, As &&...as) {
  if (!c->stream.gang) {
    hipLaunchKernelGGL(single, grid, block, shmem, (hipStream_t)c->stream, static_cast<Ts>(as)...);
    return;
  }
  try {  // (a record is ~2 KB; no exception may cross the C ABI: the session's end reports it -- gang_flush)
    c->gang_recs.emplace_back();
  } catch (const std::bad_alloc &) {
    c->gang_oom = true;
    return;
  }
  GangRec &r = c->gang_recs.back();
  r.key = GangLaunch<Body, Ts...>::key();
  r.cap = GangLaunch<Body, Ts...>::kCap;
  r.single = reinterpret_cast<const void *>(single);
  r.launch_one = &GangLaunch<Body, Ts...>::one;
  r.launch_many = &GangLaunch<Body, Ts...>::many;
  r.grid = grid;
  r.block = block;
  r.shmem = shmem;
  gang_store_args<Ts...>(r.args, std::index_sequence_for<Ts...>{}, as...);
}

struct Query {
  Map *map = nullptr;
  uint32_t n = 0, width = 0, height = 0;
  uint4 *d_desc = nullptr;  // [n_pad*4], row major, zero padded to a multiple of 64 rows
  float2 *d_kpt = nullptr;   // full precision (locFeat, AKAZEOpenCV.cpp:77-79): used for pt2D
  float2 *d_kpt6 = nullptr;  // after the .feat text round trip (6 significant digits): used by the F-matrix filter
  float *d_bow = nullptr;    // the query's BoW vector, resident (sfmloc_query_set_bow), [bow_dim] or null
  std::vector<float> h_kpt;
  bool is_view = false;  // the device arrays belong to the caller (sfmloc_query_create_view)
};

// capi.hip: K1 + K2 of the selected views against q on context c (sfmloc_match_putative's body)
int match_putative_on(Ctx *c, Query *q, const uint32_t *view_sel, uint32_t n_sel);

// hamming.hip
int launch_tile_bank(const uint4 *d_rows, uint64_t row0, uint64_t n_rows_chunk, uint4 *d_bank, hipStream_t s);
int launch_hamming_top2(Ctx *c, const Query *q, uint32_t n_work_blocks, bool use_list, uint32_t split);
int launch_blocks_from_views(Ctx *c, const uint32_t *d_sel, uint32_t n_sel, uint32_t bound);
int launch_merge_ratio_compact(Ctx *c, const Query *q, uint32_t n_sel, bool all_views, uint32_t split,
                               uint32_t n_work_blocks);


// bow.hip
struct BofModel {
  int device = 0;
  int K = 0, cdim = 0, in_dim = 0, n_pca = 0, resized = 300, levels = 2, norm_type = 2, cells = 5;
  float *d_centers = nullptr, *d_pca_mean = nullptr, *d_pca_evec = nullptr, *d_pca_eval = nullptr;
  // workspace of one call
  uint32_t *d_counts = nullptr;
  double *d_out = nullptr;
  float *d_desc = nullptr, *d_kxy = nullptr;
  int cap_n = 0;
  hipStream_t stream = nullptr;
};
struct ChainArgs;  // chain_device.h: what the shortlist's workgroup does on top of the shortlist (may be null)
int launch_bow_select(Ctx *c, const float *d_query, const uint32_t *d_cand, uint32_t n_cand,
                      uint32_t k, uint32_t *d_dist_bits, uint32_t *d_out_sel, const ChainArgs *chain = nullptr);
// sharded shortlist: this shard's k best as sortable keys (distance bits << 32 | global view id), and
// the shard's part of the global k best among n_parts key lists
int launch_bow_keys(Ctx *c, const float *d_query, uint32_t k, uint32_t *d_dist_bits, uint32_t *d_sel_tmp,
                    unsigned long long *d_keys_out);
int launch_bow_merge_select(Ctx *c, const unsigned long long *d_keys, uint32_t n_parts,
                            uint64_t part_stride_keys, uint32_t k, uint32_t n_pad, uint32_t *d_sel_out,
                            const ChainArgs *chain = nullptr);
// d_desc8 non-null: the descriptors are rows of 64 bytes (a .desc row, the first in_dim bytes used) instead of floats
int launch_bof(const BofModel *b, hipStream_t s, const float *d_desc, const float *d_kxy, int n, uint32_t *d_counts,
               double *d_out, float *d_out_f32, const uint8_t *d_desc8 = nullptr);
int launch_bof_member(const BofModel *b, GangMember *m, const float *d_kxy, int n, uint32_t *d_counts, double *d_out,
                      float *d_out_f32, const uint8_t *d_desc8);

// dense.hip: the resize + gray + min-max front end with its tables resident (sfmloc_imgbow; sfmloc_dense_gray builds a
// temporary one).  src: h x w x channels (3 = BGR, 1 = gray: a colour read of a gray file has three equal channels,
// for which BGR2GRAY is the identity, so one channel is resized).
struct DenseGrayPlan {
  int w = 0, h = 0, channels = 3, size = 300;
  int *d_xo = nullptr, *d_yo = nullptr;
  short *d_xa = nullptr, *d_ya = nullptr;
  unsigned int *d_mm = nullptr;
};
int dense_gray_plan_create(DenseGrayPlan *p, int w, int h, int channels, int size);
void dense_gray_plan_destroy(DenseGrayPlan *p);
int dense_gray_enqueue(const DenseGrayPlan *p, hipStream_t s, const uint8_t *d_src, uint8_t *d_gray_out);
int dense_gray_enqueue_member(const DenseGrayPlan *p, GangMember *m, const uint8_t *d_src, uint8_t *d_gray_out);

// akaze.hip: an extractor's device-resident pieces, for callers inside the library (imgbow.hip)
struct Akaze;
uint8_t *akaze_gray_dev(Akaze *a);           // [h*w] the image build_scale_space works on
uint8_t *akaze_desc_dev(Akaze *a);           // [n x 64] descriptors of the last describe
hipStream_t akaze_stream_now(Akaze *a);      // the stream its work is queued on (its own or a context's)
GangMember *akaze_member(Akaze *a);          // the extractor as a gang member (its launches can be recorded for a session)
// scale space of the image ALREADY in akaze_gray_dev (written on akaze_stream_now) + orientation and M-LDB at the
// device-resident keypoints d_kin [n x 4] (x, y, size, class_id); asynchronous
int akaze_compute_resident(Akaze *a, const float *d_kin, unsigned int n, int need_levels /*0: all*/);

// acransac.hip
int launch_fill_log10(double *d_L10, int n, hipStream_t s);
int launch_debug_math(int op, const double *d_in, int n, int in_stride, double *d_out, int out_stride, hipStream_t s);
// min_putative < 0: the map's params.min_putative (the query path's >=16 rule, localization.cpp:408-415)
int launch_fmatrix_filter(Ctx *c, const Query *q, uint32_t n_sel, bool all_views, int min_putative = -1);
int launch_match_set(Ctx *c, const Query *q, uint32_t n_sel, bool all_views);
// guided.hip: GeometricFilter_FMatrix_AC::Geometry_guided_matching for the views that passed K3 (their geo lists are
// replaced by (map feature, query feature) lists in d_geo_idx / d_geo_j)
int ensure_guided_workspace(Ctx *c);
int launch_guided_matching(Ctx *c, const Query *q, uint32_t n_sel, bool all_views);
int launch_emit_candidates(Ctx *c, const Query *q, uint32_t n_sel, bool all_views);
int launch_export_part(Ctx *c, void *dst_dev, uint32_t cap);
// packed_b > 0: `parts` are packed batch parts of packed_b queries (acransac.hip PartLayout), cap = their budget,
// and this query is number packed_qi of the batch
int launch_select_candidates(Ctx *c, const Query *q, const unsigned char *parts, uint32_t n_parts,
                             uint64_t part_bytes, uint32_t cap, uint32_t packed_b = 0, uint32_t packed_qi = 0,
                             bool reset_status = false);  // (also clear the context's status word: sfmloc_merge_begin)
int launch_export_packed(Ctx *c, void *dst_dev, uint32_t n_queries, uint32_t budget, uint32_t qi);
uint64_t packed_part_bytes(uint32_t n_queries, uint32_t budget);
int ctx_p3p_reserve(Ctx *c, uint32_t n_query_rows);  // capi.hip: grow the P3P workspace to a query's feature count
int launch_merge_masked_now(Ctx *c, uint32_t n_sel);  // hamming.hip: the deferred K2 as a launch of its own
int launch_p3p_init(Ctx *c);
int launch_p3p_round(Ctx *c, int batch);
int launch_p3p_seq(Ctx *c);     // the sequential form: the query's whole AC-RANSAC in one workgroup (GPU shared)
int launch_p3p_finish(Ctx *c);

}  // namespace sfmloc
