/*
 * sfmloc.h -- C ABI of the MI355X-native query-localisation hot path.
 *
 * This is the drop-in boundary for ONE path of hulop/SfMLocalization: what
 * OpenMVGLocalization_AKAZE's per-query loop (reference
 * OpenMVGLocalization_AKAZE/src/localization.cpp:285-586) and its in-process
 * twin LocalizeEngine::localize (VisionLocalizeServer/src/LocalizeEngine.cc:288-661)
 * do between "query descriptors are known" and "pose is known".
 * The reference has no FFI for this path (it is a CLI + files, or a C++ class);
 * the entry points below are what a binding for it would call.  Each one
 * cites the reference interface it replaces.
 *
 * Conventions
 *   - plain C, opaque handles, caller-owned output buffers, no torch types;
 *   - every function returns an int status: 0 = ok, <0 = error, text via
 *     sfmloc_last_error() (thread local);
 *   - "localisation failed" is NOT an error: status 0 with n_inliers == 0
 *     (reference: failure JSON without "t", localization.cpp:419-446,511-542);
 *   - one handle = one map (or one shard of a map) on one GPU; calls on one
 *     handle are serialised by the caller (the reference object is not
 *     re-entrant either, LocalizeEngine.cc:398-399,627-628);
 *   - there is no CPU fallback: without a HIP device every compute entry point
 *     returns SFMLOC_ENODEV.
 */
#ifndef SFMLOC_H
#define SFMLOC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFMLOC_ABI_VERSION 2   /* 2: sfmloc_params.guided_matching */
#define SFMLOC_DESC_BYTES 64          /* FileUtils.cpp:77-92: 61 M-LDB bytes + 3 zero bytes */
#define SFMLOC_NOMATCH 0xFFFFFFFFu
#define SFMLOC_MAX_QUERY_ROWS 65535u  /* query feature index is packed into 16 bits of the match key */

enum {
  SFMLOC_OK = 0,
  SFMLOC_EINVAL = -1, /* bad argument / shape */
  SFMLOC_ENODEV = -2, /* no HIP device: the product path has no CPU fallback */
  SFMLOC_EHIP = -3,   /* HIP runtime error */
  SFMLOC_EIO = -4,    /* file contract violated */
  SFMLOC_ECAP = -5,   /* caller buffer too small */
  SFMLOC_ENOMEM = -6
};

const char *sfmloc_last_error(void);
int sfmloc_abi_version(void);
/* number of HIP devices visible (0 on a CPU-only box); never fails */
int sfmloc_device_count(void);

/* ------------------------------------------------------------------------- */
/* Parameters: the reference's CLI keys (localization.cpp:64-82) and its      */
/* compile-time constants (localization.cpp:56-58).                            */
/* ------------------------------------------------------------------------- */
typedef struct sfmloc_params {
  float dist_ratio;         /* -f fDistRatio, default 0.6 */
  int ransac_round;         /* -r ransacRound (F-matrix AC-RANSAC iterations), default 200; callers pass 25 */
  double geom_precision;    /* -g geomLimit, default 4.0 px */
  int bow_knn;              /* -k knnbow, 0 = no shortlist.  Recorded with the map for the callers (the command-line tools
                             * and engine mirrors read it back); the library itself takes the shortlist length as an
                             * argument of sfmloc_bow_select / sfmloc_localize_bow_begin, because the reference decides it
                             * per call (LocalizeEngine.cc:337-340: only when the map has more views than knn) */
  int min_putative;         /* MINUM_NUMBER_OF_POINT_PUTATIVE_MATCH = 16 */
  int min_resection_points; /* MINUM_NUMBER_OF_POINT_RESECTION = 8 (test is ">") */
  int min_inliers;          /* MINUM_NUMBER_OF_INLIER_RESECTION = 10 (test is ">") */
  int p3p_max_iteration;    /* OpenMVG Image_Localizer_Match_Data::max_iteration default 4096 */
  uint64_t seed;            /* counter-based RNG seed (the reference's RNG is unseeded) */
  int refine_pose;          /* north-star extension A13; 0 = reference-equivalent output */
  int device;               /* HIP device ordinal */
  int profile;              /* 1 = bracket each stage with HIP events on the handle's stream, 2 = the Hamming scan only */
  int exact_rows;           /* 1 = keep the exact (nearest, second) pair of EVERY bank row (sfmloc_putative_read_rows);
                               0 = rows the screening kernel proves rejected are not finished (same matches) */
  int guided_matching;      /* -gm guidedMatch (localization.cpp:82,183; computeFeaturesAndMatches.cpp:63,89; the map
                               builder's default, ReconstructParam.py:71): every pair that passes the F-matrix AC-RANSAC
                               has its matches replaced by OpenMVG's Geometry_guided_matching under the estimated F
                               (MatchUtils.cpp:413-415) */
} sfmloc_params;

void sfmloc_default_params(sfmloc_params *p);

/* ------------------------------------------------------------------------- */
/* Map (= what the reference loads once: localization.cpp:238-280).            */
/* Host arrays are copied to HBM; the caller may free them after the call.     */
/* Views are the reconstruction's images in ascending view id; their           */
/* descriptors are the rows [view_off[v], view_off[v+1]) of the bank, in .desc */
/* file order (FileUtils.cpp:94-103).                                          */
/* ------------------------------------------------------------------------- */
typedef struct sfmloc_map_desc {
  uint32_t n_views;
  const uint32_t *view_id;      /* [n_views] strictly ascending */
  const uint32_t *view_off;     /* [n_views+1] row offsets, view_off[0] = 0 */
  const uint32_t *view_wh;      /* [n_views*2] image width,height (sfm_data views) or NULL */
  uint64_t n_rows;              /* = view_off[n_views] */
  const uint8_t *desc;          /* [n_rows*64] .desc payload rows */
  const float *kpt_xy;          /* [n_rows*2] .feat x,y or NULL (needed from the F-matrix stage on) */
  const int32_t *row_landmark;  /* [n_rows] landmark slot of (view,feat) or -1, or NULL */
  uint32_t n_landmarks;
  const uint32_t *landmark_id;  /* [n_landmarks] structure keys */
  const double *landmark_X;     /* [n_landmarks*3] */
  /* intrinsic id 0 (localization.cpp:484-487): pinhole or pinhole_radial_k3 */
  double focal, ppx, ppy, k1, k2, k3;
  uint32_t bow_dim;             /* 0 = no BoW */
  const float *bow;             /* [n_views*bow_dim] .bow vectors as f32 (BoFUtils.cpp:43-45) or NULL */
  /* 0 = pinhole: pt2D = the query keypoint (get_ud_pixel is the identity); 3 = pinhole_radial_k3: pt2D =
   * cam2ima(remove_disto(ima2cam(p))) with (k1, k2, k3) (localization.cpp:484-487).  A radial camera with zero
   * coefficients still goes through remove_disto's bisection, so the type is explicit. */
  uint32_t intrinsic_type;
} sfmloc_map_desc;

typedef struct sfmloc_map sfmloc_map;

/* replaces: Load(sfm_data) + structureToMapViewFeatTo3D + HuloSfMRegionsProvider::load
 * (localization.cpp:238-248,274-280), fed from memory */
int sfmloc_map_create(const sfmloc_map_desc *desc, const sfmloc_params *params, sfmloc_map **out);
void sfmloc_map_destroy(sfmloc_map *map);

/* replaces the same start-up sequence fed from disk: <sfm_dir>/sfm_data.json (cereal JSON: views, intrinsic 0,
 * extrinsics, structure) and <match_dir>/<basename>.{desc,feat[,bow]} of every view that has a pose
 * (localization.cpp:236-280,337-341; file layouts: SURVEY.md Appendix A).  Error text follows the reference's
 * ("The input sfm_data.json file ... cannot be read.").  SFMLOC_EIO when the contract is violated. */
int sfmloc_open(const char *sfm_dir, const char *match_dir, const sfmloc_params *params, sfmloc_map **out);

/* Host-only half of sfmloc_open (no GPU needed): parses the same files and reports what it found. */
typedef struct sfmloc_scan_info {
  uint32_t n_views_total, n_views_posed;
  uint64_t n_rows;
  uint32_t n_landmarks, n_observations, bow_dim;
  double focal, ppx, ppy, k1, k2, k3;
  uint64_t desc_fnv1a;     /* FNV-1a over all descriptor bytes in bank order */
  double kpt_sum;          /* sum of all keypoint coordinates */
  int64_t row_landmark_sum;
} sfmloc_scan_info;
int sfmloc_scan(const char *sfm_dir, const char *match_dir, sfmloc_scan_info *info);

/* Packed map file: everything sfmloc_open reads from sfm_data.json and the per-view .desc / .feat / .bow
 * files, as ONE binary written once by sfmloc_pack (host only) -- a server then opens a 10 000-view map at disk speed
 * instead of parsing a large JSON and 20 000 small files.  sfmloc_open_packed(path) gives the same map as
 * sfmloc_open(sfm_dir, match_dir); sfmloc_scan_packed is the host-only check (same fields as sfmloc_scan).  The file
 * is little-endian and versioned by its 8-byte magic; SFMLOC_EIO for anything else. */
int sfmloc_pack(const char *sfm_dir, const char *match_dir, const char *out_path);
int sfmloc_scan_packed(const char *path, sfmloc_scan_info *info);
int sfmloc_open_packed(const char *path, const sfmloc_params *params, sfmloc_map **out);

/* The view table of an sfm_data.json on its own (host only): what ExtFeatAndMatch iterates over before any
 * reconstruction exists (computeFeaturesAndMatches.cpp:118-126, AKAZEOpenCV.cpp:122-131).  Views come in id order
 * (Views is a std::map); image_path = root_path / file name and stays valid until sfmloc_view_list_close. */
typedef struct sfmloc_view_list sfmloc_view_list;
int sfmloc_view_list_open(const char *sfm_data_json, sfmloc_view_list **out, uint32_t *n_views);
int sfmloc_view_list_get(const sfmloc_view_list *list, uint32_t k, uint32_t *view_id, uint32_t *width,
                         uint32_t *height, const char **image_path);
void sfmloc_view_list_close(sfmloc_view_list *list);

/* view table of a map: ids [n_views], row offsets [n_views+1], camera centres [n_views*3] (only for maps opened
 * from sfm_data.json; used for the dead-reckoning restriction getLocalViews, SfMDataUtils.cpp:210-227) */
int sfmloc_map_views(const sfmloc_map *map, uint32_t *view_id, uint32_t *view_off, double *center);
/* image width, height of every view [n_views*2] (sfm_data views; the default size of a query without an image) */
int sfmloc_map_view_sizes(const sfmloc_map *map, uint32_t *wh);

typedef struct sfmloc_map_info {
  uint64_t n_rows;
  uint32_t n_views;
  uint32_t n_landmarks;
  uint64_t hbm_bytes; /* device memory held by the handle */
  int device;
} sfmloc_map_info;
int sfmloc_map_get_info(const sfmloc_map *map, sfmloc_map_info *info);

/* ------------------------------------------------------------------------- */
/* Query (= the output of extractAKAZESingleImg, AKAZEOpenCV.cpp:37-113).      */
/* Uploading it is separate from matching so that callers (and bench.py) can   */
/* have inputs resident in HBM before the timed region.                        */
/* ------------------------------------------------------------------------- */
typedef struct sfmloc_query sfmloc_query;

int sfmloc_query_create(sfmloc_map *map, const uint8_t *desc /*[n*64]*/, const float *kpt_xy /*[n*2] or NULL*/,
                        uint32_t n, uint32_t width, uint32_t height, sfmloc_query **out);
void sfmloc_query_destroy(sfmloc_query *q);
/* The query's BoW vector [bow_dim of the map] made resident with the query (what calcBoF produced for it,
 * LocalizeEngine.cc:337-340): sfmloc_localize_bow_begin / sfmloc_shard_bow_keys then take query_bow = NULL and no
 * per-call upload happens.  Synchronous. */
int sfmloc_query_set_bow(sfmloc_query *q, const float *query_bow);
/* A query over arrays that already ARE in device memory and stay the caller's -- e.g. slices of the buffer an all-gather
 * of extracted features wrote (images in on several ranks: the rank that owns a query extracts it, every rank matches
 * it).  Nothing is allocated or copied; the arrays must stay valid and unchanged while the query is in use.
 *   desc_dev  uint8 [n_pad * 64], row major, n_pad = n rounded up to a multiple of 64, the rows beyond n ZERO; 16-byte aligned
 *   kpt_dev   float [n * 2] as extracted;  kpt6_dev  float [n * 2] after sfmloc_feat_round_trip;  8-byte aligned
 *   bow_dev   float [bow_dim of the map], or NULL (then the calls that need one take a host pointer as usual)
 * sfmloc_feat_round_trip: host helper, the `.feat` text round trip of the keypoints (6 significant digits,
 * AKAZEOpenCV.cpp:80-81 / :106-111) that sfmloc_query_create applies itself. */
int sfmloc_query_create_view(sfmloc_map *map, const void *desc_dev, const void *kpt_dev, const void *kpt6_dev,
                             const void *bow_dev, uint32_t n, uint32_t width, uint32_t height, sfmloc_query **out);
void sfmloc_feat_round_trip(const float *kpt_xy, uint64_t n_values, float *out);

/* ------------------------------------------------------------------------- */
/* Stage A6+A7: putative matching.                                             */
/* replaces hulo::matchAKAZEToQuery (MatchUtils.cpp:283-367) with an exact     */
/* 2-NN in place of the reference's LSH, then the "< 16 matches" filter        */
/* (localization.cpp:408-415).                                                 */
/*   view_sel: indices into the map's view table (NOT view ids), strictly      */
/*   ascending, or NULL for all views (localization.cpp:386-392).              */
/* Results stay on the device for the next stage; sfmloc_putative_read copies  */
/* them out.  Asynchronous on the handle's stream.                             */
/* ------------------------------------------------------------------------- */
int sfmloc_match_putative(sfmloc_map *map, sfmloc_query *q, const uint32_t *view_sel, uint32_t n_sel);

/* Blocks until the stream is idle, then copies out
 *   view_count[n_views]          matches per view (0 for unselected views), BEFORE the >=16 filter
 *   match_i/match_j/match_d[cap] per-view lists, view v's list starting at row offset view_off[v]
 *                                (so cap must be >= n_rows), in ascending map-feature index i.
 * Any pointer may be NULL. */
int sfmloc_putative_read(sfmloc_map *map, uint32_t *view_count, uint32_t *match_i, uint32_t *match_j,
                         uint32_t *match_d, uint64_t cap);

/* Per-row view of the same stage (parity tests): key[r] = (d0<<16)|j0 of the nearest query
 * descriptor and of the second nearest, for every bank row that was searched, SFMLOC_NOMATCH elsewhere. */
int sfmloc_putative_read_rows(sfmloc_map *map, uint32_t *best0 /*[n_rows]*/, uint32_t *best1 /*[n_rows]*/);

int sfmloc_sync(sfmloc_map *map);

/* ------------------------------------------------------------------------- */
/* Stage A8: geometric filter.                                                 */
/* replaces hulo::geometricMatch (MatchUtils.cpp:372-420) = OpenMVG            */
/* GeometricFilter_FMatrix_AC(geomPrec, ransacRound) per (view, query) pair,   */
/* on the result of the last sfmloc_match_putative call; views with fewer than */
/* params.min_putative matches are dropped first (localization.cpp:408-415).   */
/* Query keypoints are the .feat-rounded ones (6 significant digits), map      */
/* keypoints come from the map's .feat (no undistortion: MatchUtils.cpp:381).  */
/* ------------------------------------------------------------------------- */
int sfmloc_geometric_filter(sfmloc_map *map, sfmloc_query *q);
/* geo_count[n_views]; geo_idx[n_rows]: view v's inliers as indices into ITS putative list, stored at
 * view_off[v], in AC-RANSAC's inlier order (ascending residual).  Synchronises. */
int sfmloc_geometric_read(sfmloc_map *map, uint32_t *geo_count, uint32_t *geo_idx, uint64_t cap);
/* The same stage as (map feature i, query feature j) pairs, view v's list at view_off[v]: with guided matching these ARE
 * the result (one match per map feature, ascending i; not a subset of the putative list, so sfmloc_geometric_read
 * refuses), without it the putative matches the indices name.  Synchronises. */
int sfmloc_geometric_read_pairs(sfmloc_map *map, uint32_t *geo_count, uint32_t *geo_i, uint32_t *geo_j, uint64_t cap);

/* ------------------------------------------------------------------------- */
/* Stage A9+A10: 2D-3D match set.                                              */
/* replaces hulo::matchProviderToMatchSet (SfMDataUtils.cpp:59-125) and the    */
/* pt2D/pt3D assembly (localization.cpp:479-501).                              */
/* ------------------------------------------------------------------------- */
int sfmloc_match_set(sfmloc_map *map, sfmloc_query *q);
int sfmloc_match_set_read(sfmloc_map *map, uint32_t *n, uint32_t *qfeat, uint32_t *landmark_id, double *pt2d,
                          double *pt3d, uint32_t cap);

/* ------------------------------------------------------------------------- */
/* Stage A11+A12: resection and pose.                                          */
/* replaces sfm::SfM_Localizer::Localize (localization.cpp:504-509; P3P        */
/* AC-RANSAC, max_iteration = params.p3p_max_iteration), the inlier gate       */
/* (localization.cpp:511) and KRt_From_P / t_out = -R^T t (:544-547).          */
/* ------------------------------------------------------------------------- */
typedef struct sfmloc_pose {
  int32_t ok;                /* 1 = localised: what the reference signals by writing "t" into the result JSON */
  int32_t n_inliers;         /* resection_data.vec_inliers.size() */
  int32_t n_matches_2d3d;    /* mapFeatTo3DFeat.size() (cpt, localization.cpp:502) */
  int32_t iterations;        /* AC-RANSAC iterations actually run */
  int32_t status;            /* bit 0/1/2: a capacity of the device workspace was exceeded */
  int32_t n_putative_views;  /* views with >= min_putative matches */
  int32_t n_geometric_views; /* views that passed the F-matrix filter */
  int32_t reserved;
  double nfa;                /* AC-RANSAC minimum NFA (log10) */
  double error_max;          /* resection_data.error_max (pixels) */
  double P[12];              /* projection_matrix, row-major 3x4 */
  double K[9], R[9], t[3];   /* KRt_From_P, row-major */
  double center[3];          /* t_out = -R^T t: the "t" of the result JSON */
  double stage_seconds[7];   /* LocalizeEngine.cc:643-658 buckets: selectBeacon, selectBow, extFeat, putMatch,
                                geoMatch, PnP, others.  Filled from the per-stage HIP events of this query when
                                params.profile == 1 (selectBow = K8, putMatch = K1+K2, geoMatch = K3, PnP = 2D-3D set +
                                P3P, others = the rest of the call's wall time; selectBeacon / extFeat are the caller's);
                                otherwise only others = the call's wall time */
} sfmloc_pose;

int sfmloc_resection(sfmloc_map *map, sfmloc_query *q);
/* pair_*: the "pair" list of the result JSON (localization.cpp:132-141): (query feature, landmark id) per
 * inlier; inlier_idx: indices into the 2D-3D match set.  Synchronises. */
int sfmloc_pose_read(sfmloc_map *map, sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark,
                     uint32_t *inlier_idx, uint32_t cap);

/* The whole per-query path: putative -> >=16 filter -> F-matrix filter -> 2D-3D set -> P3P -> pose
 * (localization.cpp:395-547 / LocalizeEngine.cc:423-585).  One host synchronisation at the end. */
int sfmloc_localize(sfmloc_map *map, sfmloc_query *q, const uint32_t *view_sel, uint32_t n_sel, sfmloc_pose *out,
                    uint32_t *pair_qfeat, uint32_t *pair_landmark, uint32_t cap);

/* Concurrent queries.  A context is a HIP stream plus the workspace of one in-flight query; contexts of one
 * map run concurrently on the GPU (the reference serves concurrent users with one engine per (user, map),
 * localizeImage.cc:71-104 -- here they share one copy of the map).  _begin enqueues the whole path and
 * returns; _end waits for it.  One query per context at a time.  Contexts are owned by the map. */
typedef struct sfmloc_context sfmloc_context;
int sfmloc_context_create(sfmloc_map *map, sfmloc_context **out);
void sfmloc_context_destroy(sfmloc_context *ctx);
int sfmloc_localize_begin(sfmloc_context *ctx, sfmloc_query *q, const uint32_t *view_sel, uint32_t n_sel);
/* localization.cpp:346-368 / LocalizeEngine.cc:333-361 in one asynchronous call: shortlist the `knn` candidate views
 * nearest to `query_bow` (exactly as sfmloc_bow_select) and localise on them.  The shortlist stays on the device --
 * the bank blocks to scan are derived from it by a kernel -- so nothing waits for the host between the two stages.
 * As in the reference the shortlist applies only when more than `knn` candidates remain; cand_views NULL = all
 * views.  query_bow NULL = the vector made resident by sfmloc_query_set_bow.  Finish with sfmloc_localize_end.
 * Same result as sfmloc_bow_select + sfmloc_localize_begin, bit for bit. */
int sfmloc_localize_bow_begin(sfmloc_context *ctx, sfmloc_query *query, const float *query_bow, uint32_t knn,
                              const uint32_t *cand_views, uint32_t n_cand);
/* the same, synchronous, on the map's own context (as sfmloc_localize; the stage read-backs then refer to it) */
int sfmloc_localize_bow(sfmloc_map *map, sfmloc_query *query, const float *query_bow, uint32_t knn,
                        const uint32_t *cand_views, uint32_t n_cand, sfmloc_pose *out, uint32_t *pair_qfeat,
                        uint32_t *pair_landmark, uint32_t cap);
int sfmloc_localize_end(sfmloc_context *ctx, sfmloc_pose *out, uint32_t *pair_qfeat, uint32_t *pair_landmark,
                        uint32_t cap);
/* n queries against all views, n_contexts of them in flight (0 = 4).  poses[n]; pair buffers may be NULL,
 * else query i's pairs start at i*pair_stride. */
int sfmloc_localize_batch(sfmloc_map *map, sfmloc_query *const *queries, uint32_t n, uint32_t n_contexts,
                          sfmloc_pose *poses, uint32_t *pair_qfeat, uint32_t *pair_landmark, uint32_t pair_stride);

/* ------------------------------------------------------------------------- */
/* Sharded maps: one process per GPU, each holding a contiguous   */
/* range of views.  Every bank row's decision depends only on the replicated     */
/* query, so putative matching and the F-matrix filter are shard-local; what is   */
/* exchanged is each shard's list of 2D-3D CANDIDATES (a "part": 16-byte header   */
/* {u32 count} + cap 40-byte candidates whose order key carries the global view   */
/* id).  The caller moves parts between ranks (torch.distributed all-gather over  */
/* RCCL) and hands the concatenation to sfmloc_merge_begin, which reproduces      */
/* matchProviderToMatchSet + Localize of the unsharded map bit for bit.           */
/*   dst_dev / parts_dev are DEVICE pointers owned by the caller.                 */
/* ------------------------------------------------------------------------- */
uint64_t sfmloc_part_bytes(uint32_t cap);
int sfmloc_shard_begin(sfmloc_context *ctx, sfmloc_query *q, const uint32_t *view_sel, uint32_t n_sel);
/* writes the part's header (true count) and its first min(count, cap) candidates to dst_dev; bytes beyond are left
 * as they were (the merging side reads `count` entries only) */
int sfmloc_shard_export(sfmloc_context *ctx, void *dst_dev, uint32_t cap);
int sfmloc_context_sync(sfmloc_context *ctx);
/* Ordering against a stream of the caller (e.g. the one its collective runs on) without blocking the host:
 *   _signal: everything queued on the context so far happens before work queued on hip_stream from now on;
 *   _wait:   everything queued on hip_stream so far happens before work queued on the context from now on.
 * hip_stream is a hipStream_t of the same device (NULL = the default stream). */
int sfmloc_context_signal(sfmloc_context *ctx, void *hip_stream);
int sfmloc_context_wait(sfmloc_context *ctx, void *hip_stream);
/* Gang sessions: up to 32 contexts of one map take one query each through the same asynchronous calls
 * (sfmloc_shard_bow_keys, sfmloc_shard_begin[_bow] + sfmloc_shard_export_packed, sfmloc_localize[_bow]_begin), and every
 * kernel of the chain is launched once for all of them (or for as many as its kernel arguments hold: 8 to 32).  Between _begin and _end the calls on these contexts only record
 * their launches; _end issues them -- one launch per kernel, the members side by side in gridDim.z -- on the first
 * context's stream and orders the other members' streams after it.  Results are those of the same calls made one context
 * at a time, bit for bit: a member's arithmetic does not change, only the number of launches (a rank of an N-rank run
 * scans 1/N of the map per query but still issues every query's launches, and the device completes only so many
 * dependent launches per second).  One host thread drives a session; with params.profile != 0 a session records
 * nothing and the members run one after the other.
 *   _counters: launches issued by this leader's sessions so far, and how many of them carried more than one member. */
int sfmloc_gang_begin(sfmloc_context *const *ctxs, uint32_t n);
/* a context (workspace) WITHOUT a stream of its own: its work is queued on `lender`'s stream (a lender destroyed first
 * stops being usable but its stream lives until the last borrower is destroyed).  For
 * the members of a gang other than the first -- a device serves only so many hardware queues well, and a context that
 * only ever works inside gang sessions needs none. */
int sfmloc_context_create_sharing(sfmloc_map *map, sfmloc_context *lender, sfmloc_context **out);
/* a context for sfmloc_merge_begin[_packed] + sfmloc_localize_end ONLY (the owner's stage of a sharded query): none of the
 * per-bank-row workspace of the matching stages (30 B per descriptor of the map and 32 MB), so a rank can afford one per
 * query of a gang.  lender: NULL = a stream of its own, else as sfmloc_context_create_sharing.  Every other asynchronous
 * call on it fails with SFMLOC_EINVAL. */
int sfmloc_context_create_merge(sfmloc_map *map, sfmloc_context *lender, sfmloc_context **out);
int sfmloc_gang_end(sfmloc_context *const *ctxs, uint32_t n);
int sfmloc_gang_counters(sfmloc_context *lead_ctx, uint64_t *launches, uint64_t *gang_launches);
/* Sharded BoW shortlist (selectViewByBoF over a map split by view).  _shard_bow_keys ranks this shard's
 * views against the query's BoW vector (query_bow, or NULL for the resident one) and writes its knn best to keys_dev
 * [knn] as sortable 64-bit keys (float32 distance bits << 32 | view id; ~0 = padding).  The caller all-gathers the key
 * lists; _shard_begin_bow takes the n_parts lists (list p at keys_dev + p*part_stride_keys keys; 0 = back to back),
 * keeps the shard's part of the GLOBAL knn best (ties to the lower view id, as the unsharded sfmloc_bow_select) and
 * runs sfmloc_shard_begin on it -- all on the device, asynchronously.  knn <= 1024, n_parts*knn <= 8192. */
int sfmloc_shard_bow_keys(sfmloc_context *ctx, sfmloc_query *q, const float *query_bow, uint32_t knn, void *keys_dev);
int sfmloc_shard_begin_bow(sfmloc_context *ctx, sfmloc_query *q, const void *keys_dev, uint32_t n_parts,
                           uint64_t part_stride_keys, uint32_t knn);
/* part p starts at parts_dev + p*part_stride (part_stride = 0 means sfmloc_part_bytes(cap): back to back) */
int sfmloc_merge_begin(sfmloc_context *ctx, sfmloc_query *q, const void *parts_dev, uint32_t n_parts, uint32_t cap,
                       uint64_t part_stride);
/* ... then sfmloc_localize_end(ctx, ...) */
/* The exchange of a whole BATCH in one buffer per shard ("packed part"): header {u32 total, n_queries, budget, flags},
 * u32 count[n_queries], u32 offset[n_queries], then (16-byte aligned) the candidates of all the batch's queries back
 * to back -- a shard sends what it found (a few MB per 256-query batch) instead of n_queries fixed-capacity parts.
 *   sfmloc_packed_bytes(n_queries, budget)   size of one packed part holding up to `budget` candidates in all;
 *   sfmloc_shard_export_packed               appends the context's candidates (after sfmloc_shard_begin[_bow]) as query
 *                                            `query_index` of the batch; the caller zeroes the first 16 bytes of
 *                                            packed_dev before the batch's first export.  When the budget does not
 *                                            suffice the query is recorded empty, flags |= 1, and `total` still counts
 *                                            every candidate: after the all-gather every rank sees total > budget and
 *                                            repeats the batch with a larger budget;
 *   sfmloc_merge_begin_packed                sfmloc_merge_begin over the gathered packed parts (part p at packed_dev +
 *                                            p*part_stride; 0 = back to back) for the batch's query `query_index`. */
uint64_t sfmloc_packed_bytes(uint32_t n_queries, uint32_t budget);
int sfmloc_shard_export_packed(sfmloc_context *ctx, void *packed_dev, uint32_t n_queries, uint32_t budget,
                               uint32_t query_index);
int sfmloc_merge_begin_packed(sfmloc_context *ctx, sfmloc_query *q, const void *packed_dev, uint32_t n_parts,
                              uint64_t part_stride, uint32_t n_queries, uint32_t budget, uint32_t query_index);
/* A rank's whole batch per call: the per-query calls above for n_queries queries, in gang sessions of
 * `gang` contexts (query i on context i mod n_ctx; the contexts [g, g + gang) of every n_ctx consecutive queries form one
 * session) -- what a caller would otherwise issue as one foreign call per query and stage.  Asynchronous like the calls
 * they stand for; same results.
 *   _bow_keys      keys_dev [n_queries][knn] u64: every query's knn best views of this shard (resident BoW vectors)
 *   _begin_bow     keys_all_dev [n_parts][n_queries][knn]: the all-gathered keys; stage 1 of every query on its part of
 *                  the global shortlist + its candidates exported to packed_dev (sfmloc_packed_bytes(n_queries, budget))
 *   _begin         the same without a shortlist (every view of the shard)
 *   sfmloc_merge_batch_begin   stage 2 (2D-3D set + P3P) of n <= 32 queries in one session: context k takes query
 *                  query_index[k] of the batch, whose parts lie in packed_all_dev (n_parts parts, part_stride bytes
 *                  apart); finish each with sfmloc_localize_end */
int sfmloc_shard_batch_bow_keys(sfmloc_context *const *ctxs, uint32_t n_ctx, uint32_t gang, sfmloc_query *const *queries,
                                uint32_t n_queries, uint32_t knn, void *keys_dev);
int sfmloc_shard_batch_begin_bow(sfmloc_context *const *ctxs, uint32_t n_ctx, uint32_t gang, sfmloc_query *const *queries,
                                 uint32_t n_queries, const void *keys_all_dev, uint32_t n_parts, uint32_t knn,
                                 void *packed_dev, uint32_t budget);
int sfmloc_shard_batch_begin(sfmloc_context *const *ctxs, uint32_t n_ctx, uint32_t gang, sfmloc_query *const *queries,
                             uint32_t n_queries, void *packed_dev, uint32_t budget);
int sfmloc_merge_batch_begin(sfmloc_context *const *ctxs, uint32_t n, sfmloc_query *const *queries,
                             const uint32_t *query_index, const void *packed_all_dev, uint32_t n_parts,
                             uint64_t part_stride, uint32_t n_queries, uint32_t budget);

/* ------------------------------------------------------------------------- */
/* Stage A5: bag-of-words view shortlist.                                        */
/* sfmloc_bow_select replaces hulo::selectViewByBoF (BoFUtils.cpp:27-68) with the */
/* exact k-NN its FLANN KD-tree approximates: the k candidate views whose .bow    */
/* vector (as float32, BoFUtils.cpp:43-45) is nearest in L2 to the query's; ties  */
/* to the lower view; result ascending (the reference returns a std::set).        */
/* cand_views: ascending view-table indices or NULL for all; requires k < n_cand  */
/* (CV_Assert, BoFUtils.cpp:30); callers skip the shortlist when the candidate    */
/* set is not larger than k (localization.cpp:346).  Synchronises.                */
/* ------------------------------------------------------------------------- */
int sfmloc_bow_select(sfmloc_map *map, const float *query_bow, const uint32_t *cand_views, uint32_t n_cand,
                      uint32_t k, uint32_t *out_sel, uint32_t *n_out);
/* The distances sfmloc_bow_select ranks by: out_dist[v] = the float32 L2 distance of view v's .bow vector to the
 * query's (same kernel, same summation order).  For the sharded shortlist: every rank ranks its own
 * views, the ranks exchange their k best (distance, view id) pairs and keep the global k best; ties go to the lower
 * view id, as in the unsharded selection. */
int sfmloc_bow_distances(sfmloc_map *map, const float *query_bow, float *out_dist /*[n_views]*/);

/* The query's BoW vector from its dense local features: PcaWrapper::calcPcaProject (PcaWrapper.cpp:67-89) +
 * BoFSpatialPyramids::calcBoF (BoFSpatialPyramids.cpp:108-302) with an exact nearest-centre search.
 * Model = what BOWfile.yml / PCAfile.yml hold (BoFSpatialPyramids.cpp:57-83, PcaWrapper.cpp:53-65). */
typedef struct sfmloc_bof_desc {
  int K;                     /* "K" */
  int in_dim;                /* dense descriptor length (61 for M-LDB bytes as floats) */
  const float *centers;      /* "Centers" [K x (n_pca ? n_pca : in_dim)] */
  int resized_image_size;    /* "ResizedImageSize" (300) */
  int use_spatial_pyramid;   /* "UseSpatialPyramid" */
  int pyramid_level;         /* "PyramidLevel" (2) */
  int norm_type;             /* "NormBofFeatureType": 0 NONE, 1 L2, 2 L1 (square root) */
  int n_pca;                 /* "DimPCA", 0 = no PCA */
  const float *pca_mean;     /* "MeanPCA" [in_dim] */
  const float *pca_eigvec;   /* "EigenVectorsPCA" first n_pca rows [n_pca x in_dim] */
  const float *pca_eigval;   /* "EigenValuesPCA" first n_pca */
} sfmloc_bof_desc;
typedef struct sfmloc_bof sfmloc_bof;
int sfmloc_bof_create(const sfmloc_bof_desc *desc, int device, sfmloc_bof **out);
void sfmloc_bof_destroy(sfmloc_bof *bof);
int sfmloc_bof_dim(const sfmloc_bof *bof); /* K x pyramid cells (500) */
/* desc [n x in_dim] f32, kpt_xy [n x 2] in the resized image; out_bow [sfmloc_bof_dim] f64.  Synchronises. */
int sfmloc_bof_compute(sfmloc_bof *bof, const float *desc, const float *kpt_xy, uint32_t n, double *out_bow);

/* ------------------------------------------------------------------------- */
/* Stage A2 / A5a: AKAZE detection and full 486-bit M-LDB description.            */
/* sfmloc_akaze_detect_and_compute replaces                                        */
/*   cv::AKAZE::create(DESCRIPTOR_MLDB, 0, 3, thres, nOct, nOctLay)->detectAndCompute(gray, noArray(), kpts, desc) */
/* (AKAZEOpenCV.cpp:44-46,67); sfmloc_akaze_compute replaces extractor->compute(gray, keypoints, desc) on given  */
/* keypoints (DenseLocalFeatureWrapper.cpp:146; class_id = evolution level, octave 0).                           */
/* An extractor is bound to one image size (it owns the scale-space buffers).                                    */
/*   kpts [cap*6]: x, y, size (diameter), angle (radians), response, class_id                                    */
/*   desc64 [cap*64]: rows exactly as saveAKAZEBin stores them (61 M-LDB bytes + 3 zero bytes, FileUtils.cpp:77-92) */
/* ------------------------------------------------------------------------- */
typedef struct sfmloc_akaze sfmloc_akaze;
int sfmloc_akaze_create(int device, int width, int height, int n_octaves, int n_sublevels, float threshold,
                        sfmloc_akaze **out);
void sfmloc_akaze_destroy(sfmloc_akaze *ak);
int sfmloc_akaze_detect_and_compute(sfmloc_akaze *ak, const uint8_t *gray, float *kpts, uint8_t *desc64, uint32_t cap,
                                    uint32_t *n_out);
/* n images of ONE size at once, extractor i taking image i (1 <= n <= 32): the scale spaces, determinants and extrema of
 * all of them go out as one launch per kernel (a gang session on the first extractor's stream) -- these are launch-bound
 * kernels on small images, eight frames cost little more than one --, the candidates' refinement on the host and the
 * orientation + M-LDB launch follow per image.  kpts[i] / descs[i] / n_out[i] as in the single call; same results. */
int sfmloc_akaze_detect_and_compute_batch(sfmloc_akaze *const *aks, const uint8_t *const *grays, uint32_t n,
                                          float *const *kpts, uint8_t *const *descs, uint32_t cap, uint32_t *n_out);
/* kin [n*4]: x, y, size, class_id */
/* detectAndCompute whose outputs stay on the device, laid out as a query block (what extractAKAZESingleImg hands
 * matchAKAZEToQuery, AKAZEOpenCV.cpp:37-113, without the .feat / .desc files in between): descriptor rows [n x 64]
 * followed by zero rows up to a multiple of 64, keypoints (x, y) float2 [n], the keypoints after the .feat text round
 * trip (6 significant digits, AKAZEOpenCV.cpp:80-81,106-111) float2 [n], and the six-float records [n x 6].  Only the
 * counts come back: ONE host synchronisation per call, nothing else crosses PCIe.  _resident_arrays: the device addresses
 * (valid until the extractor's next call) -- sfmloc_query_create_view(map, desc, kpt, kpt6, bow, n, w, h) makes the query;
 * queue its localisation on the extractor's stream (sfmloc_akaze_share_stream) or synchronise in between. */
int sfmloc_akaze_detect_resident(sfmloc_akaze *ak, const uint8_t *gray, uint32_t *n_out);
int sfmloc_akaze_detect_resident_batch(sfmloc_akaze *const *aks, const uint8_t *const *grays, uint32_t n, uint32_t *n_out);
int sfmloc_akaze_resident_arrays(sfmloc_akaze *ak, const void **desc_dev, const void **kpt_dev, const void **kpt6_dev,
                                 const void **kp6_dev);
int sfmloc_akaze_compute(sfmloc_akaze *ak, const uint8_t *gray, const float *kin, uint32_t n, uint8_t *desc64,
                         float *angle_out);
/* scale-space introspection for parity tests: level sizes, and the stacked Ldet / Lt images of the last call */
int sfmloc_akaze_levels(const sfmloc_akaze *ak, int *n_levels, int *wh);
/* From now on the extractor queues its work on the context's stream (ctx = NULL: back on its own).  A server worker that
 * extracts an image and then localises it with that context (localizeImage.cc:149-177 then LocalizeEngine::localize) does
 * the two one after the other anyway; with one stream it occupies one hardware queue instead of two, and concurrent
 * workers stop competing for queues (profiles/r02_image_in_sweep.txt).  The context must outlive the sharing. */
int sfmloc_akaze_share_stream(sfmloc_akaze *ak, sfmloc_context *ctx);
int sfmloc_akaze_read_levels(sfmloc_akaze *ak, float *ldet, float *lt);
/* the duplicate suppression of the last call from the inside (kpts_aux of AKAZEFeatures.cpp, Find_Scale_Space_Extrema): out9 =
 * extrema candidates, keypoints, rounds of the first pass, three in-kernel clocks (10 ns ticks: first pass, second pass,
 * compaction), candidates whose A / P neighbour lists spilled (decided by band scans), whose N list spilled, levels decided
 * out of the global arrays instead of LDS.  SFMLOC_AKAZE_SUPPRESS=spill / global (read by sfmloc_akaze_create) forces
 * those forms; results are the same bits. */
int sfmloc_akaze_suppress_stats(sfmloc_akaze *ak, uint32_t *out9);

/* cv::imread of the query image, host side (AKAZEOpenCV.cpp:60 `imread(filename, IMREAD_GRAYSCALE)` for extraction;
 * DenseLocalFeatureWrapper.cpp:85 and localizeImage.cc:463 `IMREAD_COLOR` for the dense-BoW front end).  Decodes
 * JPEG (baseline / extended / progressive Huffman, 8 bit, 4:4:4 / 4:2:2 / 4:2:0 / gray -- libjpeg's default path
 * restated: islow IDCT, fancy upsampling, fixed-point YCbCr->RGB; gray = the Y plane as JCS_GRAYSCALE gives it), PNG
 * (8 bit, non-interlaced; colour -> gray as png_set_rgb_to_gray(1, .299, .587)) and binary PGM / PPM.
 * color == 0: out is h x w gray; color != 0: out is h x w x 3 in B G R order.  With out == NULL only *width and
 * *height are returned (size query).  SFMLOC_EIO for a file that cannot be read or decoded, SFMLOC_ECAP when `cap`
 * bytes do not hold the image.  No device is needed. */
int sfmloc_image_decode(const uint8_t *bytes, uint64_t n_bytes, int32_t color, uint8_t *out, uint64_t cap,
                        int32_t *width, int32_t *height);
int sfmloc_image_read(const char *path, int32_t color, uint8_t *out, uint64_t cap, int32_t *width, int32_t *height);

/* The server's per-user image undistortion in front of LocalizeEngine::localize (localizeImage.cc:149-177, SURVEY
 * 8f-4).  _create = the once-per-camera part, host arithmetic only (no device needed):
 *   newCameraMat = getOptimalNewCameraMatrix(K, dist, size, 1.0, size, &validRoi)  +  cv::undistort's CV_16SC2 maps;
 * _apply = the per-image part on the GPU: undistort(image, K, dist, newCameraMat) then the crop to validRoi --
 * fixed-point bilinear remap, zero border.  K row-major 3x3; dist = k1 k2 p1 p2 [k3 [k4 k5 k6]] (n_dist 0, 4, 5, 8).
 * src: height x width x channels (1 or 3) u8; dst: roi_h x roi_w x channels.  _info: new camera matrix [9] and
 * validRoi [4] = x y w h; _maps: the full-size maps (xy [h*w*2] i16, frac [h*w] u16 = fy<<5|fx) for parity tests. */
typedef struct sfmloc_undistorter sfmloc_undistorter;
int sfmloc_undistorter_create(int device, const double *K, const double *dist, uint32_t n_dist, uint32_t width,
                              uint32_t height, sfmloc_undistorter **out);
void sfmloc_undistorter_destroy(sfmloc_undistorter *u);
int sfmloc_undistorter_info(const sfmloc_undistorter *u, double *new_camera, int32_t *roi);
int sfmloc_undistorter_maps(const sfmloc_undistorter *u, int16_t *xy, uint16_t *frac);
int sfmloc_undistorter_apply(sfmloc_undistorter *u, const uint8_t *src, uint32_t channels, uint8_t *dst, uint64_t cap);

/* Dense-BoW front end (DenseLocalFeatureWrapper.cpp:89-99): colour image (BGR, 8-bit,
 * row-major h x w x 3) -> cv::resize(size x size, INTER_CUBIC) -> BGR2GRAY -> normalize(0, 255, NORM_MINMAX);
 * gray_out [size*size].  The grid keypoints (DenseFeatureDetector.cpp:44-69) and the chaining into
 * sfmloc_akaze_compute / sfmloc_bof_compute are host glue (sfmlocalization_amd/engine.py::dense_bow). */
int sfmloc_dense_gray(int device, const uint8_t *bgr, uint32_t width, uint32_t height, uint32_t size,
                      uint8_t *gray_out);

/* The query-side BoW vector from the IMAGE as one resident, asynchronous chain: replaces, per
 * query, DenseLocalFeatureWrapper::calcDenseLocalFeature (BoWCommon/src/DenseLocalFeatureWrapper.cpp:83-183) ->
 * PcaWrapper::calcPcaProject (VisionLocalizeCommon/src/PcaWrapper.cpp:67-89) -> BoFSpatialPyramids::calcBoF
 * (BoWCommon/src/BoFSpatialPyramids.cpp:108-302) as localization.cpp:346-361 / LocalizeEngine.cc:205-232 call them.
 * _create: everything an image of this size needs is allocated once (model = the BOWfile.yml / PCAfile.yml contents as
 * for sfmloc_bof_create; in_dim must be 61; channels 3 = BGR as imread(IMREAD_COLOR) returns it, 1 = a gray image,
 * whose colour read has three equal channels).  _compute: image [height x width x channels] in host memory -> the
 * vector; nothing is allocated, nothing crosses PCIe but the image (and the result when out_bow is given).
 *   query   non-NULL: the float32 vector (what BoFUtils.cpp:51-54 hands the matcher) is written into the query's
 *           resident BoW slot, as sfmloc_query_set_bow would, ASYNCHRONOUSLY on the extractor's stream: share that
 *           stream with the context that will localise the query (sfmloc_imgbow_share_stream) so that
 *           sfmloc_localize_bow_begin(ctx, query, NULL, knn) is ordered after it, or pass out_bow (which synchronises).
 *   out_bow non-NULL: the reference's float64 vector [sfmloc_imgbow_dim], after a stream synchronisation.
 * Same kernels as sfmloc_dense_gray + sfmloc_akaze_compute + sfmloc_bof_compute: the same bits. */
typedef struct sfmloc_imgbow sfmloc_imgbow;
int sfmloc_imgbow_create(const sfmloc_bof_desc *model, int device, uint32_t width, uint32_t height, uint32_t channels,
                         sfmloc_imgbow **out);
void sfmloc_imgbow_destroy(sfmloc_imgbow *ib);
int sfmloc_imgbow_dim(const sfmloc_imgbow *ib);
int sfmloc_imgbow_share_stream(sfmloc_imgbow *ib, sfmloc_context *ctx); /* NULL: back on its own stream */
int sfmloc_imgbow_compute(sfmloc_imgbow *ib, const uint8_t *image, sfmloc_query *query, double *out_bow);
/* The vectors of n frames (1..32) in ONE gang session on the first extractor's stream: one launch per kernel for all of
 * them; the float32 vectors stay on the device (sfmloc_imgbow_vector_dev of each extractor).  The reference computes the
 * vector per query image (localization.cpp:346-361, LocalizeEngine.cc:205-232); same bits as n single calls. */
int sfmloc_imgbow_compute_batch(sfmloc_imgbow *const *ibs, const uint8_t *const *images, uint32_t n);
/* the reference's float64 vector of the extractor's last call (single or batch), after waiting for it */
int sfmloc_imgbow_vector_read(sfmloc_imgbow *ib, double *out_bow);
/* where the float32 vector of a call WITHOUT a query lands (device memory, [sfmloc_imgbow_dim] floats): the bow_dev of a
 * query view (sfmloc_query_create_view) over an extractor's resident outputs */
const void *sfmloc_imgbow_vector_dev(const sfmloc_imgbow *ib);
/* an extractor on a stream of its own (not shared) works beside the feature extraction of the same frame; this makes the
 * context's stream wait for everything the extractor has queued so far (call it before sfmloc_localize_bow_begin) */
int sfmloc_imgbow_order_before(sfmloc_imgbow *ib, sfmloc_context *ctx);

/* ------------------------------------------------------------------------- */
/* Map-side matching: the reference's matchAKAZE /         */
/* trackAKAZE on the same kernels.  Views are addressed by their index in the  */
/* map's view table (ascending view id).                                       */
/* ------------------------------------------------------------------------- */
typedef struct sfmloc_matches sfmloc_matches; /* PairWiseMatches: std::map<Pair, vector<IndMatch>> flattened */

/* The descriptors (and keypoints) of one map image as a query object, rebuilt on the device from the bank: what
 * the reference obtains by re-reading <base>.desc for the second image of a pair (MatchUtils.cpp:85-96). */
int sfmloc_query_from_view(sfmloc_map *map, uint32_t view_index, sfmloc_query **out);

/* matchAKAZE's per-pair body (MatchUtils.cpp:99-150) for first = each selected view, second = the query:
 * 2-NN + ratio of every row of `first` among the query's rows (sfmloc_match_putative), then the one-to-one filter
 * (a query row hit twice or more loses all its hits, :125-143) and the emit loop that never emits the last row of
 * `first` (:146).  Results are read with sfmloc_putative_read. */
int sfmloc_match_one_to_one(sfmloc_map *map, sfmloc_query *query, const uint32_t *view_sel, uint32_t n_sel);

/* hulo::matchAKAZE (MatchUtils.cpp:73-152): pairs[2k], pairs[2k+1] = (first, second) view indices; images with
 * fewer than 2 descriptors are skipped (:101-103); pairs without matches get no entry. */
int sfmloc_match_pairs(sfmloc_map *map, const uint32_t *pairs, uint32_t n_pairs, sfmloc_matches **out);

/* hulo::trackAKAZE (MatchUtils.cpp:156-277) over the map's views in table order: consecutive-frame matches as
 * above, then chained up to max_frame_dist frames apart (:240-276).  Every consecutive pair has an entry (possibly
 * empty), longer-range pairs only when non-empty, as the reference's std::map ends up. */
int sfmloc_track(sfmloc_map *map, uint32_t max_frame_dist, sfmloc_matches **out);

/* hulo::geometricMatch (MatchUtils.cpp:372-420) for map image pairs: F-matrix AC-RANSAC (params.ransac_round,
 * params.geom_precision) on the given putative lists -- pair k = (pairs[2k], pairs[2k+1]) owns
 * match_i/j[offsets[k] .. offsets[k+1]) -- keeping a pair iff it has more than 2.5*7 inliers; the surviving matches
 * come in AC-RANSAC's inlier order.  No minimum list length is applied here (the caller applies ExtFeatAndMatch's
 * minMatch, computeFeaturesAndMatches.cpp:211-221).  With params.guided_matching a surviving pair's matches are
 * replaced by the guided ones (all features of both images under the estimated F, one per feature of I, ascending;
 * possibly none -- the pair keeps its entry, as in Robust_model_estimation). */
int sfmloc_geometric_pairs(sfmloc_map *map, const uint32_t *pairs, uint32_t n_pairs, const uint64_t *offsets,
                           const uint32_t *match_i, const uint32_t *match_j, sfmloc_matches **out);

uint32_t sfmloc_matches_pairs(const sfmloc_matches *m); /* number of (I, J) entries, ascending (I, J) */
int sfmloc_matches_pair(const sfmloc_matches *m, uint32_t k, uint32_t *view_i, uint32_t *view_j, uint32_t *n);
int sfmloc_matches_read(const sfmloc_matches *m, uint32_t k, uint32_t *i, uint32_t *j, uint32_t cap);
void sfmloc_matches_destroy(sfmloc_matches *m);

/* Parity probe: runs one of the f64 device building blocks over n items (tests compare with the oracle).
 * op: 0 log10, 1 sqrt+div, 2 cubic, 3 quartic, 4 seven-point, 5 P3P, 6 KRt_From_P, 7 sample,
 *     8 seven-point, wave-parallel form (K3's fast kernel; layout of op 4) */
int sfmloc_debug_math(int device, int op, const double *in, int n, int in_stride, double *out, int out_stride);
/* Test hook: allocation k (0..11) of the NEXT regrowth of a context's P3P workspace fails with SFMLOC_ENOMEM (one shot;
 * -1 disarms).  The regrowth has no counterpart in the reference (localization.cpp:479-509 has no size limit). */
void sfmloc_debug_fail_p3p_alloc(int k);

/* ------------------------------------------------------------------------- */
/* Measurement (params.profile = 1): accumulated HIP-event time per kernel on  */
/* the handle's stream.                                                        */
/* ------------------------------------------------------------------------- */
enum {
  SFMLOC_K_HAMMING = 0,
  SFMLOC_K_COMPACT = 1,
  SFMLOC_K_FMATRIX = 2,
  SFMLOC_K_MATCHSET = 3,
  SFMLOC_K_P3P = 4,
  SFMLOC_K_BOW = 5,
  SFMLOC_K_COUNT = 8
};
typedef struct sfmloc_kernel_stats {
  double total_ms[SFMLOC_K_COUNT];
  uint64_t launches[SFMLOC_K_COUNT];
  uint64_t hamming_pairs;      /* bank rows x query rows compared since reset */
  uint64_t hamming_alg_bytes;  /* 64*rows + 64*Nq + 12*matches is finalised by the caller */
  uint64_t hamming_lane_ops;       /* VALU lane-instructions K1 issued for those pairs (35 per exact pair; fewer when
                                      the screening kernel rejects a pair on its prefix distance) */
  uint64_t hamming_pairs_finished; /* screened pairs whose full distance had to be computed */
  uint64_t hamming_rows_flagged;   /* bank rows the screening kernel handed to the exact kernel */
} sfmloc_kernel_stats;
int sfmloc_stats_read(sfmloc_map *map, sfmloc_kernel_stats *out); /* synchronises */
int sfmloc_stats_reset(sfmloc_map *map);
/* Changes params.profile of a live map: 0 = no events, 1 = every stage bracketed, 2 = the Hamming scan only.  Each
 * bracketed stage costs two marker packets on the stream (about 10 us of idle GPU between dependent kernels), so a
 * latency measurement wants 0 or 2. */
int sfmloc_set_profile(sfmloc_map *map, int level);

#ifdef __cplusplus
}
#endif
#endif /* SFMLOC_H */
