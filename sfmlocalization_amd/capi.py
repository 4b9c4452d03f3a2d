"""ctypes mirror of include/sfmloc.h (one Python method per C entry point, same names and meaning)."""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib

NOMATCH = 0xFFFFFFFF
K_HAMMING, K_COMPACT, K_FMATRIX, K_MATCHSET, K_P3P, K_BOW, K_COUNT = 0, 1, 2, 3, 4, 5, 8

OK, EINVAL, ENODEV, EHIP, EIO, ECAP, ENOMEM = 0, -1, -2, -3, -4, -5, -6


class SfmlocError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sfmloc error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("dist_ratio", C.c_float),
        ("ransac_round", C.c_int),
        ("geom_precision", C.c_double),
        ("bow_knn", C.c_int),
        ("min_putative", C.c_int),
        ("min_resection_points", C.c_int),
        ("min_inliers", C.c_int),
        ("p3p_max_iteration", C.c_int),
        ("seed", C.c_uint64),
        ("refine_pose", C.c_int),
        ("device", C.c_int),
        ("profile", C.c_int),
        ("exact_rows", C.c_int),
        ("guided_matching", C.c_int),
    ]


class MapDesc(C.Structure):
    _fields_ = [
        ("n_views", C.c_uint32),
        ("view_id", C.POINTER(C.c_uint32)),
        ("view_off", C.POINTER(C.c_uint32)),
        ("view_wh", C.POINTER(C.c_uint32)),
        ("n_rows", C.c_uint64),
        ("desc", C.POINTER(C.c_uint8)),
        ("kpt_xy", C.POINTER(C.c_float)),
        ("row_landmark", C.POINTER(C.c_int32)),
        ("n_landmarks", C.c_uint32),
        ("landmark_id", C.POINTER(C.c_uint32)),
        ("landmark_X", C.POINTER(C.c_double)),
        ("focal", C.c_double), ("ppx", C.c_double), ("ppy", C.c_double),
        ("k1", C.c_double), ("k2", C.c_double), ("k3", C.c_double),
        ("bow_dim", C.c_uint32),
        ("bow", C.POINTER(C.c_float)),
        ("intrinsic_type", C.c_uint32),
    ]


class MapInfo(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_views", C.c_uint32), ("n_landmarks", C.c_uint32),
                ("hbm_bytes", C.c_uint64), ("device", C.c_int)]


class Pose(C.Structure):
    _fields_ = [("ok", C.c_int32), ("n_inliers", C.c_int32), ("n_matches_2d3d", C.c_int32),
                ("iterations", C.c_int32), ("status", C.c_int32), ("n_putative_views", C.c_int32),
                ("n_geometric_views", C.c_int32), ("reserved", C.c_int32),
                ("nfa", C.c_double), ("error_max", C.c_double),
                ("P", C.c_double * 12), ("K", C.c_double * 9), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("center", C.c_double * 3), ("stage_seconds", C.c_double * 7)]


class BofDesc(C.Structure):
    _fields_ = [("K", C.c_int), ("in_dim", C.c_int), ("centers", C.POINTER(C.c_float)),
                ("resized_image_size", C.c_int), ("use_spatial_pyramid", C.c_int), ("pyramid_level", C.c_int),
                ("norm_type", C.c_int), ("n_pca", C.c_int), ("pca_mean", C.POINTER(C.c_float)),
                ("pca_eigvec", C.POINTER(C.c_float)), ("pca_eigval", C.POINTER(C.c_float))]


class ScanInfo(C.Structure):
    _fields_ = [("n_views_total", C.c_uint32), ("n_views_posed", C.c_uint32), ("n_rows", C.c_uint64),
                ("n_landmarks", C.c_uint32), ("n_observations", C.c_uint32), ("bow_dim", C.c_uint32),
                ("focal", C.c_double), ("ppx", C.c_double), ("ppy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("k3", C.c_double),
                ("desc_fnv1a", C.c_uint64), ("kpt_sum", C.c_double), ("row_landmark_sum", C.c_int64)]


class KernelStats(C.Structure):
    _fields_ = [("total_ms", C.c_double * K_COUNT), ("launches", C.c_uint64 * K_COUNT),
                ("hamming_pairs", C.c_uint64), ("hamming_alg_bytes", C.c_uint64),
                ("hamming_lane_ops", C.c_uint64), ("hamming_pairs_finished", C.c_uint64),
                ("hamming_rows_flagged", C.c_uint64)]


# every symbol include/sfmloc.h declares (tests check the library exports each one)
SYMBOLS = [
    "sfmloc_last_error", "sfmloc_abi_version", "sfmloc_device_count", "sfmloc_default_params",
    "sfmloc_map_create", "sfmloc_map_destroy", "sfmloc_map_get_info", "sfmloc_open", "sfmloc_scan",
    "sfmloc_map_views", "sfmloc_map_view_sizes",
    "sfmloc_query_create", "sfmloc_query_destroy",
    "sfmloc_match_putative", "sfmloc_putative_read", "sfmloc_putative_read_rows", "sfmloc_sync",
    "sfmloc_geometric_filter", "sfmloc_geometric_read", "sfmloc_match_set", "sfmloc_match_set_read",
    "sfmloc_resection", "sfmloc_pose_read", "sfmloc_localize", "sfmloc_debug_math", "sfmloc_debug_fail_p3p_alloc",
    "sfmloc_context_create", "sfmloc_context_destroy", "sfmloc_localize_begin", "sfmloc_localize_end",
    "sfmloc_dense_gray", "sfmloc_bow_distances", "sfmloc_query_from_view", "sfmloc_match_one_to_one", "sfmloc_match_pairs", "sfmloc_track", "sfmloc_geometric_pairs",
    "sfmloc_matches_pairs", "sfmloc_matches_pair", "sfmloc_matches_read", "sfmloc_matches_destroy",
    "sfmloc_localize_batch", "sfmloc_part_bytes", "sfmloc_shard_begin", "sfmloc_shard_export",
    "sfmloc_context_sync", "sfmloc_merge_begin", "sfmloc_bow_select", "sfmloc_bof_create", "sfmloc_bof_destroy",
    "sfmloc_bof_dim", "sfmloc_bof_compute", "sfmloc_akaze_create", "sfmloc_akaze_destroy",
    "sfmloc_akaze_detect_and_compute", "sfmloc_akaze_compute", "sfmloc_akaze_levels", "sfmloc_akaze_read_levels", "sfmloc_akaze_suppress_stats",
    "sfmloc_akaze_share_stream", "sfmloc_akaze_detect_and_compute_batch",
    "sfmloc_stats_read", "sfmloc_stats_reset", "sfmloc_set_profile", "sfmloc_image_decode", "sfmloc_image_read",
    "sfmloc_view_list_open", "sfmloc_view_list_get", "sfmloc_view_list_close", "sfmloc_localize_bow_begin", "sfmloc_localize_bow",
    "sfmloc_pack", "sfmloc_scan_packed", "sfmloc_open_packed",
    "sfmloc_undistorter_create", "sfmloc_undistorter_destroy", "sfmloc_undistorter_info", "sfmloc_undistorter_maps",
    "sfmloc_undistorter_apply",
    "sfmloc_query_set_bow", "sfmloc_context_signal", "sfmloc_context_wait", "sfmloc_shard_bow_keys",
    "sfmloc_geometric_read_pairs", "sfmloc_shard_begin_bow", "sfmloc_packed_bytes", "sfmloc_shard_export_packed", "sfmloc_merge_begin_packed",
    "sfmloc_gang_begin", "sfmloc_gang_end", "sfmloc_gang_counters", "sfmloc_context_create_sharing", "sfmloc_context_create_merge",
    "sfmloc_query_create_view", "sfmloc_feat_round_trip",
    "sfmloc_imgbow_create", "sfmloc_imgbow_destroy", "sfmloc_imgbow_dim", "sfmloc_imgbow_share_stream", "sfmloc_imgbow_compute", "sfmloc_imgbow_compute_batch", "sfmloc_imgbow_vector_read",
    "sfmloc_shard_batch_bow_keys", "sfmloc_shard_batch_begin_bow", "sfmloc_shard_batch_begin", "sfmloc_merge_batch_begin",
    "sfmloc_imgbow_vector_dev", "sfmloc_imgbow_order_before", "sfmloc_akaze_detect_resident", "sfmloc_akaze_detect_resident_batch", "sfmloc_akaze_resident_arrays",
]

_bound = False

GANG_MAX = 32


class gang:
    """`with gang(ctxs):` -- sfmloc_gang_begin / _end: the asynchronous calls made on these contexts (<= GANG_MAX, one
    map) inside the block record their launches, and leaving it issues them as ONE launch per kernel for all members."""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)
        self._arr = (C.c_void_p * len(self.ctxs))(*[c._h for c in self.ctxs])

    def __enter__(self):
        _check(_L().sfmloc_gang_begin(self._arr, len(self.ctxs)))
        return self

    def __exit__(self, *exc):
        _check(_L().sfmloc_gang_end(self._arr, len(self.ctxs)))
        return False


def result_fingerprint(pose, pair_qfeat, pair_landmark, view_counts=True):
    """Every bit of a query's result -- ok, status, counts, iterations, NFA, error_max, P, K, R, t, centre and the
    inlier pairs (localization.cpp:504-547: what the result file is written from) -- except the stage timings, as 8
    bytes.  Two runs of a query agree bit for bit iff their fingerprints do (bench.py `identical_to_single_flight`,
    tests/test_gpu_load.py).  view_counts=False leaves out n_putative_views / n_geometric_views, which a shard of a
    sharded map counts for its own views only."""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    raw = C.string_at(C.addressof(pose), Pose.stage_seconds.offset)
    if not view_counts:
        o = Pose.n_putative_views.offset
        raw = raw[:o] + raw[o + 8:]
    h.update(raw)
    h.update(np.ascontiguousarray(pair_qfeat).tobytes())
    h.update(np.ascontiguousarray(pair_landmark).tobytes())
    return h.digest()


def _handles(objs):
    return (C.c_void_p * len(objs))(*[o._h for o in objs])


def shard_batch_bow_keys(ctxs, gang, queries, knn, keys_dev_ptr):
    """sfmloc_shard_batch_bow_keys: every query's knn best views of this shard -> keys [len(queries)][knn] on the device"""
    _check(_L().sfmloc_shard_batch_bow_keys(_handles(ctxs), len(ctxs), gang, _handles(queries), len(queries), knn,
                                            C.c_void_p(keys_dev_ptr)))


def shard_batch_begin_bow(ctxs, gang, queries, keys_all_dev_ptr, n_parts, knn, packed_dev_ptr, budget):
    """sfmloc_shard_batch_begin_bow: stage 1 of a batch on the global shortlist + the packed export, one call"""
    _check(_L().sfmloc_shard_batch_begin_bow(_handles(ctxs), len(ctxs), gang, _handles(queries), len(queries),
                                             C.c_void_p(keys_all_dev_ptr), n_parts, knn, C.c_void_p(packed_dev_ptr), budget))


def shard_batch_begin(ctxs, gang, queries, packed_dev_ptr, budget):
    _check(_L().sfmloc_shard_batch_begin(_handles(ctxs), len(ctxs), gang, _handles(queries), len(queries),
                                         C.c_void_p(packed_dev_ptr), budget))


def merge_batch_begin(ctxs, queries, query_index, packed_all_dev_ptr, n_parts, part_stride, n_queries, budget):
    """sfmloc_merge_batch_begin: stage 2 of len(ctxs) queries in one session (finish each with Context.end)"""
    qi = np.ascontiguousarray(query_index, np.uint32)
    _check(_L().sfmloc_merge_batch_begin(_handles(ctxs), len(ctxs), _handles(queries), _ptr(qi, C.c_uint32),
                                         C.c_void_p(packed_all_dev_ptr), n_parts, part_stride, n_queries, budget))


def feat_round_trip(kpt_xy):
    """sfmloc_feat_round_trip: the keypoints after the reference's `.feat` text round trip (6 significant digits)."""
    k = np.ascontiguousarray(kpt_xy, dtype=np.float32)
    out = np.empty_like(k)
    L = _L()
    L.sfmloc_feat_round_trip.restype = None
    L.sfmloc_feat_round_trip.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_float)]
    L.sfmloc_feat_round_trip(_ptr(k, C.c_float), k.size, _ptr(out, C.c_float))
    return out


def gang_counters(lead_ctx):
    """(launches issued by this leader's sessions, of which gang launches with more than one member)"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    _check(_L().sfmloc_gang_counters(lead_ctx._h, C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def _L():
    global _bound
    L = _lib.load()
    if not _bound:
        L.sfmloc_last_error.restype = C.c_char_p
        L.sfmloc_abi_version.restype = C.c_int
        L.sfmloc_device_count.restype = C.c_int
        L.sfmloc_default_params.restype = None
        L.sfmloc_default_params.argtypes = [C.POINTER(Params)]
        L.sfmloc_map_create.argtypes = [C.POINTER(MapDesc), C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.sfmloc_map_destroy.restype = None
        L.sfmloc_map_destroy.argtypes = [C.c_void_p]
        L.sfmloc_map_get_info.argtypes = [C.c_void_p, C.POINTER(MapInfo)]
        L.sfmloc_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.sfmloc_scan.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(ScanInfo)]
        L.sfmloc_map_views.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_double)]
        L.sfmloc_query_create.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.sfmloc_query_destroy.restype = None
        L.sfmloc_query_destroy.argtypes = [C.c_void_p]
        L.sfmloc_match_putative.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        L.sfmloc_putative_read.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4 + [C.c_uint64]
        L.sfmloc_putative_read_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.sfmloc_bow_distances.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.sfmloc_dense_gray.argtypes = [C.c_int, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_uint8)]
        L.sfmloc_query_from_view.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.sfmloc_match_one_to_one.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        L.sfmloc_match_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_void_p)]
        L.sfmloc_track.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.sfmloc_geometric_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]
        L.sfmloc_matches_pairs.argtypes = [C.c_void_p]
        L.sfmloc_matches_pairs.restype = C.c_uint32
        L.sfmloc_matches_pair.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32)]
        L.sfmloc_matches_read.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
        L.sfmloc_matches_destroy.argtypes = [C.c_void_p]
        L.sfmloc_matches_destroy.restype = None
        L.sfmloc_sync.argtypes = [C.c_void_p]
        U32P, F64P = C.POINTER(C.c_uint32), C.POINTER(C.c_double)
        L.sfmloc_geometric_filter.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_geometric_read.argtypes = [C.c_void_p, U32P, U32P, C.c_uint64]
        L.sfmloc_geometric_read_pairs.argtypes = [C.c_void_p, U32P, U32P, U32P, C.c_uint64]
        L.sfmloc_match_set.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_match_set_read.argtypes = [C.c_void_p, U32P, U32P, U32P, F64P, F64P, C.c_uint32]
        L.sfmloc_resection.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_pose_read.argtypes = [C.c_void_p, C.POINTER(Pose), U32P, U32P, U32P, C.c_uint32]
        L.sfmloc_localize.argtypes = [C.c_void_p, C.c_void_p, U32P, C.c_uint32, C.POINTER(Pose), U32P, U32P,
                                      C.c_uint32]
        L.sfmloc_debug_math.argtypes = [C.c_int, C.c_int, F64P, C.c_int, C.c_int, F64P, C.c_int]
        L.sfmloc_debug_fail_p3p_alloc.argtypes = [C.c_int]
        L.sfmloc_debug_fail_p3p_alloc.restype = None
        L.sfmloc_context_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.sfmloc_context_destroy.restype = None
        L.sfmloc_context_destroy.argtypes = [C.c_void_p]
        L.sfmloc_localize_begin.argtypes = [C.c_void_p, C.c_void_p, U32P, C.c_uint32]
        L.sfmloc_localize_end.argtypes = [C.c_void_p, C.POINTER(Pose), U32P, U32P, C.c_uint32]
        L.sfmloc_part_bytes.restype = C.c_uint64
        L.sfmloc_part_bytes.argtypes = [C.c_uint32]
        L.sfmloc_shard_begin.argtypes = [C.c_void_p, C.c_void_p, U32P, C.c_uint32]
        L.sfmloc_shard_export.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.sfmloc_context_sync.argtypes = [C.c_void_p]
        L.sfmloc_merge_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
        L.sfmloc_bow_select.argtypes = [C.c_void_p, C.POINTER(C.c_float), U32P, C.c_uint32, C.c_uint32, U32P, U32P]
        L.sfmloc_bof_create.argtypes = [C.POINTER(BofDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.sfmloc_bof_destroy.restype = None
        L.sfmloc_bof_destroy.argtypes = [C.c_void_p]
        L.sfmloc_bof_dim.argtypes = [C.c_void_p]
        L.sfmloc_bof_compute.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, F64P]
        L.sfmloc_akaze_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_void_p)]
        L.sfmloc_akaze_destroy.restype = None
        L.sfmloc_akaze_destroy.argtypes = [C.c_void_p]
        L.sfmloc_akaze_detect_and_compute.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float),
                                                      C.POINTER(C.c_uint8), C.c_uint32, U32P]
        L.sfmloc_akaze_compute.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_uint32,
                                           C.POINTER(C.c_uint8), C.POINTER(C.c_float)]
        L.sfmloc_akaze_levels.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sfmloc_akaze_share_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_akaze_read_levels.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.sfmloc_akaze_suppress_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.sfmloc_localize_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32,
                                            C.POINTER(Pose), U32P, U32P, C.c_uint32]
        L.sfmloc_stats_read.argtypes = [C.c_void_p, C.POINTER(KernelStats)]
        L.sfmloc_stats_reset.argtypes = [C.c_void_p]
        L.sfmloc_set_profile.argtypes = [C.c_void_p, C.c_int]
        L.sfmloc_query_set_bow.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.sfmloc_context_signal.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_context_wait.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_shard_bow_keys.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_void_p]
        L.sfmloc_shard_begin_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32]
        L.sfmloc_packed_bytes.restype = C.c_uint64
        L.sfmloc_packed_bytes.argtypes = [C.c_uint32, C.c_uint32]
        L.sfmloc_shard_export_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.sfmloc_merge_begin_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32,
                                                C.c_uint32, C.c_uint32]
        L.sfmloc_imgbow_create.argtypes = [C.POINTER(BofDesc), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.sfmloc_imgbow_destroy.restype = None
        L.sfmloc_imgbow_destroy.argtypes = [C.c_void_p]
        L.sfmloc_imgbow_dim.argtypes = [C.c_void_p]
        L.sfmloc_imgbow_share_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_imgbow_compute.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_void_p, F64P]
        L.sfmloc_imgbow_compute_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32]
        L.sfmloc_imgbow_vector_read.argtypes = [C.c_void_p, F64P]
        VPP = C.POINTER(C.c_void_p)
        L.sfmloc_shard_batch_bow_keys.argtypes = [VPP, C.c_uint32, C.c_uint32, VPP, C.c_uint32, C.c_uint32, C.c_void_p]
        L.sfmloc_shard_batch_begin_bow.argtypes = [VPP, C.c_uint32, C.c_uint32, VPP, C.c_uint32, C.c_void_p, C.c_uint32,
                                                   C.c_uint32, C.c_void_p, C.c_uint32]
        L.sfmloc_shard_batch_begin.argtypes = [VPP, C.c_uint32, C.c_uint32, VPP, C.c_uint32, C.c_void_p, C.c_uint32]
        L.sfmloc_merge_batch_begin.argtypes = [VPP, C.c_uint32, VPP, U32P, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32,
                                               C.c_uint32]
        L.sfmloc_imgbow_order_before.argtypes = [C.c_void_p, C.c_void_p]
        L.sfmloc_imgbow_vector_dev.restype = C.c_void_p
        L.sfmloc_imgbow_vector_dev.argtypes = [C.c_void_p]
        L.sfmloc_akaze_detect_resident.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), U32P]
        L.sfmloc_akaze_detect_resident_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_uint8)), C.c_uint32, U32P]
        L.sfmloc_akaze_resident_arrays.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4
        _bound = True
    return L


def _check(rc):
    if rc != 0:
        raise SfmlocError(rc, _L().sfmloc_last_error().decode("utf-8", "replace"))


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def _sel(view_sel):
    """view selection -> (pointer, count, keep-alive).  An EMPTY selection must stay a non-null pointer with count 0
    (NULL means "all views" in the C ABI), so it is backed by a one-element dummy."""
    sel = np.ascontiguousarray(view_sel, dtype=np.uint32).ravel()
    back = sel if sel.shape[0] else np.zeros(1, np.uint32)
    return _ptr(back, C.c_uint32), int(sel.shape[0]), back


def device_count():
    return int(_L().sfmloc_device_count())


def packed_bytes(n_queries, budget):
    return int(_L().sfmloc_packed_bytes(n_queries, budget))


def part_bytes(cap):
    return int(_L().sfmloc_part_bytes(cap))


def scan(sfm_dir, match_dir):
    """sfmloc_scan: parse <sfm_dir>/sfm_data.json + <match_dir>/*.desc|feat|bow on the host, no GPU."""
    info = ScanInfo()
    _check(_L().sfmloc_scan(os.fsencode(sfm_dir), os.fsencode(match_dir), C.byref(info)))
    return {f: getattr(info, f) for f, _ in ScanInfo._fields_}


def pack(sfm_dir, match_dir, out_path):
    """sfmloc_pack: the map's files -> one packed binary (host only)."""
    _check(_L().sfmloc_pack(os.fsencode(sfm_dir), os.fsencode(match_dir), os.fsencode(out_path)))


def scan_packed(path):
    """sfmloc_scan_packed: the fields of scan() from a packed map file, no GPU."""
    info = ScanInfo()
    L = _L()
    L.sfmloc_scan_packed.argtypes = [C.c_char_p, C.POINTER(ScanInfo)]
    _check(L.sfmloc_scan_packed(os.fsencode(path), C.byref(info)))
    return {f: getattr(info, f) for f, _ in ScanInfo._fields_}


def default_params(**overrides):
    p = Params()
    _L().sfmloc_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def dense_gray(bgr, size=300, device=0):
    """sfmloc_dense_gray: BGR u8 [h, w, 3] -> the size x size gray image the dense AKAZE describes
    (DenseLocalFeatureWrapper.cpp:89-99)."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, c = bgr.shape
    if c != 3:
        raise ValueError("dense_gray: expected an h x w x 3 BGR image")
    out = np.zeros((size, size), np.uint8)
    _check(_L().sfmloc_dense_gray(device, _ptr(bgr, C.c_uint8), w, h, size, _ptr(out, C.c_uint8)))
    return out


def image_decode(data, color=False):
    """sfmloc_image_decode: the bytes of a JPEG / PNG / PGM / PPM file -> what cv::imread returns for it: u8 [h, w]
    (IMREAD_GRAYSCALE, AKAZEOpenCV.cpp:60) or u8 [h, w, 3] in B G R order (IMREAD_COLOR,
    DenseLocalFeatureWrapper.cpp:85).  Host code; raises SfmlocError (SFMLOC_EIO) for an undecodable file."""
    raw = np.frombuffer(bytes(data), dtype=np.uint8)
    w, h = C.c_int32(0), C.c_int32(0)
    L = _L()
    L.sfmloc_image_decode.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_int32, C.POINTER(C.c_uint8), C.c_uint64,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    _check(L.sfmloc_image_decode(_ptr(raw, C.c_uint8), raw.size, int(bool(color)), None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value, 3) if color else (h.value, w.value), np.uint8)
    _check(L.sfmloc_image_decode(_ptr(raw, C.c_uint8), raw.size, int(bool(color)), _ptr(out, C.c_uint8), out.size,
                                 C.byref(w), C.byref(h)))
    return out


def image_size(path):
    """(width, height) of an image file from its header (sfmloc_image_read with a NULL buffer)."""
    w, h = C.c_int32(0), C.c_int32(0)
    L = _L()
    L.sfmloc_image_read.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32)]
    _check(L.sfmloc_image_read(os.fsencode(path), 0, None, 0, C.byref(w), C.byref(h)))
    return w.value, h.value


def image_read(path, color=False):
    """sfmloc_image_read: cv::imread(path, IMREAD_GRAYSCALE | IMREAD_COLOR) as the reference's tools call it."""
    w, h = C.c_int32(0), C.c_int32(0)
    L = _L()
    L.sfmloc_image_read.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32)]
    bpath = os.fsencode(path)
    _check(L.sfmloc_image_read(bpath, int(bool(color)), None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value, 3) if color else (h.value, w.value), np.uint8)
    _check(L.sfmloc_image_read(bpath, int(bool(color)), _ptr(out, C.c_uint8), out.size, C.byref(w), C.byref(h)))
    return out


class Undistorter:
    """sfmloc_undistorter: the server's per-user undistortion (localizeImage.cc:149-177).  new_camera (3x3) and roi
    (x, y, w, h) are what getOptimalNewCameraMatrix(K, dist, size, 1.0, size, &validRoi) returns; apply() is
    cv::undistort + the crop to validRoi, on the GPU."""

    def __init__(self, K, dist, width, height, device=0):
        L = _L()
        L.sfmloc_undistorter_create.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.sfmloc_undistorter_destroy.restype = None
        L.sfmloc_undistorter_destroy.argtypes = [C.c_void_p]
        L.sfmloc_undistorter_info.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.sfmloc_undistorter_maps.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.POINTER(C.c_uint16)]
        L.sfmloc_undistorter_apply.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint8),
                                               C.c_uint64]
        Kc = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        d = np.ascontiguousarray(dist, dtype=np.float64).ravel()
        self._h = C.c_void_p()
        self.width, self.height = int(width), int(height)
        _check(L.sfmloc_undistorter_create(device, _ptr(Kc, C.c_double), _ptr(d, C.c_double) if d.size else None,
                                           d.size, self.width, self.height, C.byref(self._h)))
        P = np.zeros(9, np.float64)
        roi = np.zeros(4, np.int32)
        _check(L.sfmloc_undistorter_info(self._h, _ptr(P, C.c_double), _ptr(roi, C.c_int32)))
        self.new_camera = P.reshape(3, 3)
        self.roi = tuple(int(v) for v in roi)

    def maps(self):
        xy = np.zeros((self.height, self.width, 2), np.int16)
        fr = np.zeros((self.height, self.width), np.uint16)
        _check(_L().sfmloc_undistorter_maps(self._h, _ptr(xy, C.c_int16), _ptr(fr, C.c_uint16)))
        return xy, fr

    def apply(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        ch = 1 if img.ndim == 2 else img.shape[2]
        if img.shape[0] != self.height or img.shape[1] != self.width:
            raise ValueError("Undistorter.apply: image size differs from the plan's")
        out = np.zeros((self.roi[3], self.roi[2]) if img.ndim == 2 else (self.roi[3], self.roi[2], ch), np.uint8)
        _check(_L().sfmloc_undistorter_apply(self._h, _ptr(img, C.c_uint8), ch, _ptr(out, C.c_uint8), out.size))
        return out

    def close(self):
        if self._h:
            _L().sfmloc_undistorter_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def debug_fail_p3p_alloc(k):
    """sfmloc_debug_fail_p3p_alloc: allocation k of the next P3P workspace regrowth fails (one shot; -1 disarms)."""
    _L().sfmloc_debug_fail_p3p_alloc(int(k))


def debug_math(op, x, out_stride, device=0):
    """sfmloc_debug_math: run an f64 device building block over the rows of x."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    out = np.zeros((x.shape[0], out_stride), np.float64)
    _check(_L().sfmloc_debug_math(device, op, _ptr(x, C.c_double), x.shape[0], x.shape[1], _ptr(out, C.c_double),
                                  out_stride))
    return out


class Map:
    """Owns a sfmloc_map handle: the reconstruction's descriptor bank (and, later, its keypoints,
    landmarks and BoW vectors) resident in HBM -- what localization.cpp:238-280 loads once."""

    def __init__(self, view_id, view_off, desc, params=None, view_wh=None, kpt_xy=None, row_landmark=None,
                 landmark_id=None, landmark_X=None, intrinsic=None, bow=None):
        self._h = None
        self._children = weakref.WeakSet()
        self.view_id = np.ascontiguousarray(view_id, dtype=np.uint32)
        self.view_off = np.ascontiguousarray(view_off, dtype=np.uint32)
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 64)
        self.n_rows = desc.shape[0]
        self.n_views = self.view_id.shape[0]
        keep = [self.view_id, self.view_off, desc]
        d = MapDesc()
        d.n_views = self.n_views
        d.view_id = _ptr(self.view_id, C.c_uint32)
        d.view_off = _ptr(self.view_off, C.c_uint32)
        d.n_rows = self.n_rows
        d.desc = _ptr(desc, C.c_uint8)
        if view_wh is not None:
            view_wh = np.ascontiguousarray(view_wh, dtype=np.uint32).reshape(-1, 2)
            keep.append(view_wh)
            d.view_wh = _ptr(view_wh, C.c_uint32)
        if kpt_xy is not None:
            kpt_xy = np.ascontiguousarray(kpt_xy, dtype=np.float32).reshape(-1, 2)
            keep.append(kpt_xy)
            d.kpt_xy = _ptr(kpt_xy, C.c_float)
        if row_landmark is not None:
            row_landmark = np.ascontiguousarray(row_landmark, dtype=np.int32)
            landmark_id = np.ascontiguousarray(landmark_id, dtype=np.uint32)
            landmark_X = np.ascontiguousarray(landmark_X, dtype=np.float64).reshape(-1, 3)
            keep += [row_landmark, landmark_id, landmark_X]
            d.row_landmark = _ptr(row_landmark, C.c_int32)
            d.n_landmarks = landmark_id.shape[0]
            d.landmark_id = _ptr(landmark_id, C.c_uint32)
            d.landmark_X = _ptr(landmark_X, C.c_double)
        if intrinsic is not None:
            d.focal, d.ppx, d.ppy = intrinsic[:3]
            if len(intrinsic) >= 6:           # (f, ppx, ppy, k1, k2, k3) = pinhole_radial_k3
                d.k1, d.k2, d.k3 = intrinsic[3:6]
                d.intrinsic_type = 3
        if bow is not None:
            bow = np.ascontiguousarray(bow, dtype=np.float32).reshape(self.n_views, -1)
            keep.append(bow)
            d.bow_dim = bow.shape[1]
            d.bow = _ptr(bow, C.c_float)
        self.params = params if params is not None else default_params()
        h = C.c_void_p()
        _check(_L().sfmloc_map_create(C.byref(d), C.byref(self.params), C.byref(h)))
        self._h = h
        del keep

    @classmethod
    def open(cls, sfm_dir, match_dir, params=None):
        """sfmloc_open: load the reference's on-disk map (sfm_data.json + .desc/.feat[/.bow])."""
        self = cls.__new__(cls)
        self._h = None
        self._children = weakref.WeakSet()
        self.params = params if params is not None else default_params()
        h = C.c_void_p()
        _check(_L().sfmloc_open(os.fsencode(sfm_dir), os.fsencode(match_dir), C.byref(self.params), C.byref(h)))
        self._h = h
        i = self.info()
        self.n_rows, self.n_views = i["n_rows"], i["n_views"]
        self.view_id = np.zeros(self.n_views, np.uint32)
        self.view_off = np.zeros(self.n_views + 1, np.uint32)
        self.view_center = np.zeros((self.n_views, 3), np.float64)
        _check(_L().sfmloc_map_views(self._h, _ptr(self.view_id, C.c_uint32), _ptr(self.view_off, C.c_uint32),
                                     _ptr(self.view_center, C.c_double)))
        return self

    @classmethod
    def open_packed(cls, path, params=None):
        """sfmloc_open_packed: the map written by pack(); equal to open() on the files it was packed from."""
        self = cls.__new__(cls)
        self._h = None
        self._children = weakref.WeakSet()
        self.params = params if params is not None else default_params()
        h = C.c_void_p()
        L = _L()
        L.sfmloc_open_packed.argtypes = [C.c_char_p, C.POINTER(Params), C.POINTER(C.c_void_p)]
        _check(L.sfmloc_open_packed(os.fsencode(path), C.byref(self.params), C.byref(h)))
        self._h = h
        i = self.info()
        self.n_rows, self.n_views = i["n_rows"], i["n_views"]
        self.view_id = np.zeros(self.n_views, np.uint32)
        self.view_off = np.zeros(self.n_views + 1, np.uint32)
        self.view_center = np.zeros((self.n_views, 3), np.float64)
        _check(L.sfmloc_map_views(self._h, _ptr(self.view_id, C.c_uint32), _ptr(self.view_off, C.c_uint32),
                                  _ptr(self.view_center, C.c_double)))
        return self

    def close(self):
        if self._h is not None:
            # queries and contexts hold pointers into the map: release them first, whatever order the
            # garbage collector would have chosen
            for ch in list(self._children):
                try:
                    ch.close()
                except Exception:
                    pass
            _L().sfmloc_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        i = MapInfo()
        _check(_L().sfmloc_map_get_info(self._h, C.byref(i)))
        return {"n_rows": i.n_rows, "n_views": i.n_views, "n_landmarks": i.n_landmarks,
                "hbm_bytes": i.hbm_bytes, "device": i.device}

    def query(self, desc, kpt_xy=None, width=0, height=0):
        return Query(self, desc, kpt_xy, width, height)

    def match_putative(self, q, view_sel=None):
        """sfmloc_match_putative: asynchronous; results stay on the device."""
        if view_sel is None:
            _check(_L().sfmloc_match_putative(self._h, q._h, None, 0))
        else:
            p, n, keep = _sel(view_sel)
            _check(_L().sfmloc_match_putative(self._h, q._h, p, n))

    def putative_read(self):
        """-> view_count[V], match_i, match_j, match_d (each [n_rows]; view v's list at view_off[v])."""
        cnt = np.zeros(self.n_views, np.uint32)
        mi = np.full(self.n_rows, NOMATCH, np.uint32)
        mj = np.full(self.n_rows, NOMATCH, np.uint32)
        md = np.full(self.n_rows, NOMATCH, np.uint32)
        _check(_L().sfmloc_putative_read(self._h, _ptr(cnt, C.c_uint32), _ptr(mi, C.c_uint32),
                                         _ptr(mj, C.c_uint32), _ptr(md, C.c_uint32), self.n_rows))
        return cnt, mi, mj, md

    def putative_read_rows(self):
        b0 = np.empty(self.n_rows, np.uint32)
        b1 = np.empty(self.n_rows, np.uint32)
        _check(_L().sfmloc_putative_read_rows(self._h, _ptr(b0, C.c_uint32), _ptr(b1, C.c_uint32)))
        return b0, b1

    def sync(self):
        _check(_L().sfmloc_sync(self._h))

    # ----- map-side matching (matchAKAZE / trackAKAZE, MatchUtils.cpp:73-277) -----
    def query_view(self, desc_ptr, kpt_ptr, kpt6_ptr, bow_ptr, n, width, height):
        """sfmloc_query_create_view: a query over device arrays that stay the caller's (integers = device pointers)."""
        return Query._over_device_arrays(self, desc_ptr, kpt_ptr, kpt6_ptr, bow_ptr, n, width, height)

    def query_from_view(self, view_index):
        """sfmloc_query_from_view: one map image's descriptors as a Query (rebuilt from the bank on the device)."""
        return Query._from_view(self, int(view_index))

    def match_one_to_one(self, q, view_sel=None):
        """sfmloc_match_one_to_one: putative matching + matchAKAZE's one-to-one filter; read with putative_read()."""
        if view_sel is None:
            _check(_L().sfmloc_match_one_to_one(self._h, q._h, None, 0))
        else:
            p, n, keep = _sel(view_sel)
            _check(_L().sfmloc_match_one_to_one(self._h, q._h, p, n))

    @staticmethod
    def _take_matches(h):
        out = {}
        try:
            for k in range(_L().sfmloc_matches_pairs(h)):
                vi, vj, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
                _check(_L().sfmloc_matches_pair(h, k, C.byref(vi), C.byref(vj), C.byref(n)))
                mi = np.zeros(n.value, np.uint32)
                mj = np.zeros(n.value, np.uint32)
                _check(_L().sfmloc_matches_read(h, k, _ptr(mi, C.c_uint32), _ptr(mj, C.c_uint32), n.value))
                out[(vi.value, vj.value)] = (mi, mj)
        finally:
            _L().sfmloc_matches_destroy(h)
        return out

    def match_pairs(self, pairs):
        """sfmloc_match_pairs (hulo::matchAKAZE): pairs = [(first, second) view indices] ->
        {(first, second): (i[], j[])} in std::map order."""
        arr = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        h = C.c_void_p()
        _check(_L().sfmloc_match_pairs(self._h, _ptr(arr, C.c_uint32), arr.shape[0], C.byref(h)))
        return self._take_matches(h)

    def track(self, max_frame_dist):
        """sfmloc_track (hulo::trackAKAZE) over the views in table order."""
        h = C.c_void_p()
        _check(_L().sfmloc_track(self._h, int(max_frame_dist), C.byref(h)))
        return self._take_matches(h)

    def geometric_pairs(self, matches):
        """sfmloc_geometric_pairs (hulo::geometricMatch): {(I, J): (i[], j[])} -> the same for the pairs that pass the
        F-matrix AC-RANSAC, matches in its inlier order."""
        keys = sorted(matches)
        pairs = np.array(keys, np.uint32).reshape(-1, 2)
        off = np.zeros(len(keys) + 1, np.uint64)
        off[1:] = np.cumsum([len(matches[k][0]) for k in keys])
        mi = np.concatenate([np.asarray(matches[k][0], np.uint32) for k in keys]) if keys else np.zeros(0, np.uint32)
        mj = np.concatenate([np.asarray(matches[k][1], np.uint32) for k in keys]) if keys else np.zeros(0, np.uint32)
        mi, mj = np.ascontiguousarray(mi, np.uint32), np.ascontiguousarray(mj, np.uint32)
        h = C.c_void_p()
        _check(_L().sfmloc_geometric_pairs(self._h, _ptr(pairs, C.c_uint32), len(keys), _ptr(off, C.c_uint64),
                                           _ptr(mi, C.c_uint32), _ptr(mj, C.c_uint32), C.byref(h)))
        return self._take_matches(h)

    def geometric_filter(self, q):
        _check(_L().sfmloc_geometric_filter(self._h, q._h))

    def geometric_read(self):
        """-> geo_count[V], geo_idx[n_rows] (indices into each view's putative list, AC-RANSAC inlier order)."""
        cnt = np.zeros(self.n_views, np.uint32)
        idx = np.full(self.n_rows, NOMATCH, np.uint32)
        _check(_L().sfmloc_geometric_read(self._h, _ptr(cnt, C.c_uint32), _ptr(idx, C.c_uint32), self.n_rows))
        return cnt, idx

    def geometric_read_pairs(self):
        """-> geo_count[V], geo_i[n_rows], geo_j[n_rows]: each view's geometric matches as (map feature, query feature),
        view v's list at view_off[v] -- the guided matches with params.guided_matching, else the putative matches the
        inlier indices name."""
        cnt = np.zeros(self.n_views, np.uint32)
        gi = np.full(self.n_rows, NOMATCH, np.uint32)
        gj = np.full(self.n_rows, NOMATCH, np.uint32)
        _check(_L().sfmloc_geometric_read_pairs(self._h, _ptr(cnt, C.c_uint32), _ptr(gi, C.c_uint32),
                                                _ptr(gj, C.c_uint32), self.n_rows))
        return cnt, gi, gj

    def match_set(self, q):
        _check(_L().sfmloc_match_set(self._h, q._h))

    def match_set_read(self, cap=65536):
        n = C.c_uint32()
        qf = np.zeros(cap, np.uint32)
        lm = np.zeros(cap, np.uint32)
        p2 = np.zeros((cap, 2), np.float64)
        p3 = np.zeros((cap, 3), np.float64)
        _check(_L().sfmloc_match_set_read(self._h, C.byref(n), _ptr(qf, C.c_uint32), _ptr(lm, C.c_uint32),
                                          _ptr(p2, C.c_double), _ptr(p3, C.c_double), cap))
        k = n.value
        return qf[:k].copy(), lm[:k].copy(), p2[:k].copy(), p3[:k].copy()

    def resection(self, q):
        _check(_L().sfmloc_resection(self._h, q._h))

    def pose_read(self, cap=65536):
        pose = Pose()
        pq = np.zeros(cap, np.uint32)
        pl = np.zeros(cap, np.uint32)
        ii = np.zeros(cap, np.uint32)
        _check(_L().sfmloc_pose_read(self._h, C.byref(pose), _ptr(pq, C.c_uint32), _ptr(pl, C.c_uint32),
                                     _ptr(ii, C.c_uint32), cap))
        k = pose.n_inliers if pose.ok else 0
        return pose, pq[:k].copy(), pl[:k].copy(), ii[:k].copy()

    def localize(self, q, view_sel=None, cap=65536):
        """sfmloc_localize: the whole per-query path; -> (Pose, pair_qfeat, pair_landmark)."""
        pose = Pose()
        pq = np.zeros(cap, np.uint32)
        pl = np.zeros(cap, np.uint32)
        if view_sel is None:
            sel_p, n_sel = None, 0
        else:
            sel_p, n_sel, keep = _sel(view_sel)
        _check(_L().sfmloc_localize(self._h, q._h, sel_p, n_sel, C.byref(pose), _ptr(pq, C.c_uint32),
                                    _ptr(pl, C.c_uint32), cap))
        k = pose.n_inliers if pose.ok else 0
        return pose, pq[:k].copy(), pl[:k].copy()

    def view_sizes(self):
        """sfmloc_map_view_sizes: (width, height) of every view, [n_views, 2]."""
        wh = np.zeros((self.n_views, 2), np.uint32)
        _check(_L().sfmloc_map_view_sizes(self._h, _ptr(wh, C.c_uint32)))
        return wh

    def localize_bow(self, q, query_bow, knn, cand_views=None, cap=65536):
        """sfmloc_localize_bow: BoW shortlist (when more than knn candidates remain) + the whole path in one call on
        the map's own context; -> (Pose, pair_qfeat, pair_landmark)."""
        pose = Pose()
        pq = np.zeros(cap, np.uint32)
        pl = np.zeros(cap, np.uint32)
        b = np.ascontiguousarray(query_bow, dtype=np.float32).ravel()
        if cand_views is None:
            sel_p, n_sel = None, 0
        else:
            sel_p, n_sel, keep = _sel(cand_views)
        L = _L()
        L.sfmloc_localize_bow.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32),
                                          C.c_uint32, C.POINTER(Pose), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                          C.c_uint32]
        _check(L.sfmloc_localize_bow(self._h, q._h, _ptr(b, C.c_float), int(knn), sel_p, n_sel, C.byref(pose),
                                     _ptr(pq, C.c_uint32), _ptr(pl, C.c_uint32), cap))
        k = pose.n_inliers if pose.ok else 0
        return pose, pq[:k].copy(), pl[:k].copy()

    def context(self, share=None, merge_only=False):
        return Context(self, share, merge_only)

    def bow_select(self, query_bow, k, cand_views=None):
        """sfmloc_bow_select (selectViewByBoF): -> ascending view-table indices of the k nearest .bow vectors."""
        qb = np.ascontiguousarray(query_bow, dtype=np.float32).ravel()
        out = np.zeros(max(k, 1), np.uint32)
        n = C.c_uint32()
        if cand_views is None:
            cp, nc = None, 0
        else:
            cv = np.ascontiguousarray(cand_views, dtype=np.uint32)
            cp, nc = _ptr(cv, C.c_uint32), cv.shape[0]
        _check(_L().sfmloc_bow_select(self._h, _ptr(qb, C.c_float), cp, nc, k, _ptr(out, C.c_uint32), C.byref(n)))
        return out[:n.value].copy()

    def bow_distances(self, query_bow):
        """sfmloc_bow_distances: float32 L2 distance of every view's .bow vector to the query's (what bow_select ranks)."""
        qb = np.ascontiguousarray(query_bow, dtype=np.float32).ravel()
        out = np.zeros(self.n_views, np.float32)
        _check(_L().sfmloc_bow_distances(self._h, _ptr(qb, C.c_float), _ptr(out, C.c_float)))
        return out

    def localize_batch(self, queries, n_contexts=4, cap=0):
        """sfmloc_localize_batch: queries pipelined over n_contexts streams; -> list of Pose (+ pairs if cap)."""
        n = len(queries)
        arr = (C.c_void_p * n)(*[q._h for q in queries])
        poses = (Pose * n)()
        pq = np.zeros((n, cap), np.uint32) if cap else None
        pl = np.zeros((n, cap), np.uint32) if cap else None
        _check(_L().sfmloc_localize_batch(self._h, arr, n, n_contexts, poses, _ptr(pq, C.c_uint32),
                                          _ptr(pl, C.c_uint32), cap))
        if cap:
            return list(poses), pq, pl
        return list(poses)

    def stats(self):
        s = KernelStats()
        _check(_L().sfmloc_stats_read(self._h, C.byref(s)))
        return s

    def stats_reset(self):
        _check(_L().sfmloc_stats_reset(self._h))

    def set_profile(self, level):
        """sfmloc_set_profile: 0 off, 1 every stage, 2 the Hamming scan only."""
        _check(_L().sfmloc_set_profile(self._h, int(level)))


NORM_TYPES = {"NONE": 0, "L2": 1, "L1": 2}


def _bof_desc(centers, in_dim, resized=300, use_pyramid=True, pyramid_level=2, norm="L1", pca_mean=None,
              pca_eigvec=None, pca_eigval=None, n_pca=0):
    """-> (BofDesc, the arrays it points into)"""
    centers = np.ascontiguousarray(centers, np.float32)
    d = BofDesc()
    d.K, d.in_dim = centers.shape[0], int(in_dim)
    d.centers = _ptr(centers, C.c_float)
    d.resized_image_size, d.use_spatial_pyramid, d.pyramid_level = int(resized), int(bool(use_pyramid)), int(pyramid_level)
    d.norm_type = NORM_TYPES[norm] if isinstance(norm, str) else int(norm)
    keep = [centers]
    if n_pca:
        pm = np.ascontiguousarray(pca_mean, np.float32).ravel()
        pe = np.ascontiguousarray(np.asarray(pca_eigvec, np.float32)[:n_pca])
        pv = np.ascontiguousarray(np.asarray(pca_eigval, np.float32).ravel()[:n_pca])
        keep += [pm, pe, pv]
        d.n_pca, d.pca_mean, d.pca_eigvec, d.pca_eigval = int(n_pca), _ptr(pm, C.c_float), _ptr(pe, C.c_float), _ptr(pv, C.c_float)
    return d, keep


def _bof_desc_from_files(bow_file, pca_file=None, in_dim=61):
    from . import fileio
    b = fileio.read_cv_yaml(bow_file)
    kw = {}
    if pca_file:
        p = fileio.read_cv_yaml(pca_file)
        kw = dict(pca_mean=p["MeanPCA"], pca_eigvec=p["EigenVectorsPCA"], pca_eigval=p["EigenValuesPCA"],
                  n_pca=int(p["DimPCA"]))
    return dict(centers=b["Centers"], in_dim=in_dim, resized=int(b["ResizedImageSize"]),
                use_pyramid=bool(b["UseSpatialPyramid"]), pyramid_level=int(b["PyramidLevel"]),
                norm=b["NormBofFeatureType"], **kw)


class ImgBow:
    """sfmloc_imgbow: the query-side BoW vector from the image as one resident, asynchronous chain
    (DenseLocalFeatureWrapper::calcDenseLocalFeature -> PcaWrapper::calcPcaProject -> BoFSpatialPyramids::calcBoF)."""

    def __init__(self, width, height, channels=3, device=0, **model):
        self._h = None
        d, keep = _bof_desc(**model)
        h = C.c_void_p()
        _check(_L().sfmloc_imgbow_create(C.byref(d), device, int(width), int(height), int(channels), C.byref(h)))
        self._h = h
        self.width, self.height, self.channels = int(width), int(height), int(channels)
        self.dim = int(_L().sfmloc_imgbow_dim(h))

    @classmethod
    def from_files(cls, bow_file, pca_file, width, height, channels=3, device=0):
        return cls(width, height, channels, device, **_bof_desc_from_files(bow_file, pca_file))

    def share_stream(self, ctx):
        _check(_L().sfmloc_imgbow_share_stream(self._h, None if ctx is None else ctx._h))

    def order_before(self, ctx):
        """the context's stream waits for what this extractor has queued (an extractor on a stream of its own)"""
        _check(_L().sfmloc_imgbow_order_before(self._h, ctx._h))

    def vector_dev(self):
        """device address of the float32 vector a compute() without a query writes (a query view's bow_dev)"""
        return int(_L().sfmloc_imgbow_vector_dev(self._h) or 0)

    def compute(self, image, query=None, want_vector=None):
        """image [h, w, channels] (or [h, w] for channels = 1) u8.  query: its resident BoW slot is filled
        (asynchronously: share the stream with the context that localises it).  want_vector (default: when no query is
        given): also return the float64 vector, which synchronises."""
        img = np.ascontiguousarray(image, np.uint8)
        assert img.size == self.width * self.height * self.channels, img.shape
        want = (query is None) if want_vector is None else want_vector
        out = np.zeros(self.dim, np.float64) if want else None
        _check(_L().sfmloc_imgbow_compute(self._h, _ptr(img, C.c_uint8), None if query is None else query._h,
                                          _ptr(out, C.c_double)))
        return out

    def vector_read(self):
        """the float64 vector of the last compute() / compute_batch() this extractor took part in (synchronises)"""
        out = np.zeros(self.dim, np.float64)
        _check(_L().sfmloc_imgbow_vector_read(self._h, _ptr(out, C.c_double)))
        return out

    @staticmethod
    def compute_batch(extractors, images):
        """sfmloc_imgbow_compute_batch: the vectors of len(images) frames (one extractor each, same image size) in one gang
        session on the first extractor's stream -- one launch per kernel for all of them.  The float32 vectors stay on the
        device (vector_dev of each extractor); asynchronous."""
        n = len(images)
        assert 1 <= n <= len(extractors) and n <= GANG_MAX
        imgs = [np.ascontiguousarray(g, np.uint8) for g in images]
        for e, g in zip(extractors, imgs):
            assert g.size == e.width * e.height * e.channels, g.shape
        hs = (C.c_void_p * n)(*[e._h for e in extractors[:n]])
        ps = (C.c_void_p * n)(*[g.ctypes.data for g in imgs])
        _check(_L().sfmloc_imgbow_compute_batch(hs, ps, n))

    def close(self):
        if self._h is not None:
            _L().sfmloc_imgbow_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BofModel:
    """sfmloc_bof: BOWfile.yml (+ PCAfile.yml) on the GPU; compute() = calcPcaProject + calcBoF."""

    def __init__(self, centers, in_dim, resized=300, use_pyramid=True, pyramid_level=2, norm="L1", pca_mean=None,
                 pca_eigvec=None, pca_eigval=None, n_pca=0, device=0):
        self._h = None
        d, keep = _bof_desc(centers, in_dim, resized, use_pyramid, pyramid_level, norm, pca_mean, pca_eigvec, pca_eigval, n_pca)
        centers = keep[0]
        h = C.c_void_p()
        _check(_L().sfmloc_bof_create(C.byref(d), device, C.byref(h)))
        self._h = h
        self.dim = int(_L().sfmloc_bof_dim(h))
        self.resized = int(resized)
        self.K = int(centers.shape[0])

    @classmethod
    def from_files(cls, bow_file, pca_file=None, in_dim=61, device=0):
        return cls(device=device, **_bof_desc_from_files(bow_file, pca_file, in_dim))

    def compute(self, desc, kpt_xy):
        desc = np.ascontiguousarray(desc, np.float32)
        kpt_xy = np.ascontiguousarray(kpt_xy, np.float32).reshape(-1, 2)
        out = np.zeros(self.dim, np.float64)
        _check(_L().sfmloc_bof_compute(self._h, _ptr(desc, C.c_float), _ptr(kpt_xy, C.c_float), desc.shape[0],
                                       _ptr(out, C.c_double)))
        return out

    def close(self):
        if self._h is not None:
            _L().sfmloc_bof_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Akaze:
    """sfmloc_akaze: AKAZE + M-LDB extractor for one image size (extractAKAZESingleImg's compute part,
    AKAZEOpenCV.cpp:44-46,67; defaults AKAZEOption.h:31-34)."""

    def __init__(self, width, height, n_octaves=4, n_sublevels=4, threshold=0.001, device=0):
        self._h = None
        self.width, self.height = int(width), int(height)
        h = C.c_void_p()
        _check(_L().sfmloc_akaze_create(device, self.width, self.height, n_octaves, n_sublevels, threshold, C.byref(h)))
        self._h = h
        n = C.c_int()
        wh = (C.c_int * 64)()
        _check(_L().sfmloc_akaze_levels(h, C.byref(n), wh))
        self.levels = [(wh[2 * i], wh[2 * i + 1]) for i in range(n.value)]

    def share_stream(self, ctx):
        """sfmloc_akaze_share_stream: queue the extraction on the context's stream (None: back on its own)."""
        _check(_L().sfmloc_akaze_share_stream(self._h, None if ctx is None else ctx._h))

    def detect_and_compute(self, gray, cap=65536):
        """-> kpts [n x 6] (x, y, size, angle, response, class_id), desc [n x 64] (.desc rows)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (self.height, self.width)
        if getattr(self, "_out_cap", 0) != cap:      # output staging kept between calls (5.8 MB at the default cap)
            self._out_kp = np.zeros((cap, 6), np.float32)
            self._out_desc = np.zeros((cap, 64), np.uint8)
            self._out_cap = cap
        kp, desc = self._out_kp, self._out_desc
        n = C.c_uint32()
        _check(_L().sfmloc_akaze_detect_and_compute(self._h, _ptr(gray, C.c_uint8), _ptr(kp, C.c_float),
                                                    _ptr(desc, C.c_uint8), cap, C.byref(n)))
        return kp[:n.value].copy(), desc[:n.value].copy()

    @staticmethod
    def detect_and_compute_batch(extractors, grays, cap=65536):
        """sfmloc_akaze_detect_and_compute_batch: extractors[i] takes grays[i] (all of one size), the device side of
        detection as ONE launch per kernel for all of them -> [(kpts, desc)] as detect_and_compute gives."""
        n = len(extractors)
        assert n == len(grays) and n >= 1
        imgs = [np.ascontiguousarray(g, np.uint8) for g in grays]
        for e, g in zip(extractors, imgs):
            assert g.shape == (e.height, e.width)
            if getattr(e, "_out_cap", 0) != cap:
                e._out_kp = np.zeros((cap, 6), np.float32)
                e._out_desc = np.zeros((cap, 64), np.uint8)
                e._out_cap = cap
        aks = (C.c_void_p * n)(*[e._h for e in extractors])
        gp = (C.POINTER(C.c_uint8) * n)(*[_ptr(g, C.c_uint8) for g in imgs])
        kp = (C.POINTER(C.c_float) * n)(*[_ptr(e._out_kp, C.c_float) for e in extractors])
        dp = (C.POINTER(C.c_uint8) * n)(*[_ptr(e._out_desc, C.c_uint8) for e in extractors])
        n_out = (C.c_uint32 * n)()
        _check(_L().sfmloc_akaze_detect_and_compute_batch(aks, gp, n, kp, dp, cap, n_out))
        return [(e._out_kp[:n_out[i]].copy(), e._out_desc[:n_out[i]].copy()) for i, e in enumerate(extractors)]

    def resident_arrays(self):
        """sfmloc_akaze_resident_arrays -> device addresses (desc, kpt, kpt6, kp6) of the last resident detection"""
        p = [C.c_void_p() for _ in range(4)]
        _check(_L().sfmloc_akaze_resident_arrays(self._h, *[C.byref(x) for x in p]))
        return tuple(int(x.value or 0) for x in p)

    def detect_resident(self, gray):
        """sfmloc_akaze_detect_resident: outputs stay on the device as a query block -> number of keypoints"""
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (self.height, self.width)
        n = C.c_uint32()
        _check(_L().sfmloc_akaze_detect_resident(self._h, _ptr(gray, C.c_uint8), C.byref(n)))
        return int(n.value)

    @staticmethod
    def detect_resident_batch(extractors, grays):
        n = len(extractors)
        imgs = [np.ascontiguousarray(g, np.uint8) for g in grays]
        aks = (C.c_void_p * n)(*[e._h for e in extractors])
        gp = (C.POINTER(C.c_uint8) * n)(*[_ptr(g, C.c_uint8) for g in imgs])
        n_out = (C.c_uint32 * n)()
        _check(_L().sfmloc_akaze_detect_resident_batch(aks, gp, n, n_out))
        return [int(x) for x in n_out]

    def query_view(self, m, n, bow_dev=0):
        """the last resident detection as a query of map m (sfmloc_query_create_view over the extractor's arrays)"""
        desc, kpt, kpt6, _ = self.resident_arrays()
        return m.query_view(desc, kpt, kpt6, bow_dev, n, self.width, self.height)

    def compute(self, gray, kin):
        gray = np.ascontiguousarray(gray, np.uint8)
        kin = np.ascontiguousarray(kin, np.float32).reshape(-1, 4)
        desc = np.zeros((kin.shape[0], 64), np.uint8)
        ang = np.zeros(kin.shape[0], np.float32)
        _check(_L().sfmloc_akaze_compute(self._h, _ptr(gray, C.c_uint8), _ptr(kin, C.c_float), kin.shape[0],
                                         _ptr(desc, C.c_uint8), _ptr(ang, C.c_float)))
        return desc, ang

    def read_levels(self):
        tot = sum(w * h for w, h in self.levels)
        ldet = np.zeros(tot, np.float32)
        lt = np.zeros(tot, np.float32)
        _check(_L().sfmloc_akaze_read_levels(self._h, _ptr(ldet, C.c_float), _ptr(lt, C.c_float)))
        return ldet, lt

    def suppress_stats(self):
        """The duplicate suppression of the last call (sfmloc_akaze_suppress_stats): counts and in-kernel clocks."""
        out = np.zeros(9, np.uint32)
        _check(_L().sfmloc_akaze_suppress_stats(self._h, _ptr(out, C.c_uint32)))
        names = ("candidates", "keypoints", "rounds", "first_pass_ticks", "second_pass_ticks", "compaction_ticks",
                 "spilled_ap", "spilled_n", "global_levels")
        return dict(zip(names, (int(v) for v in out)))

    def close(self):
        if self._h is not None:
            _L().sfmloc_akaze_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """sfmloc_context: one in-flight query (stream + workspace) on a map; begin() is asynchronous."""

    def __init__(self, m, share=None, merge_only=False):
        self._h = None
        self.map = m
        h = C.c_void_p()
        if merge_only:  # sfmloc_context_create_merge: for merge_begin[_packed] + end only, no matching workspace
            _check(_L().sfmloc_context_create_merge(m._h, None if share is None else share._h, C.byref(h)))
            self._lender = share
        elif share is None:
            _check(_L().sfmloc_context_create(m._h, C.byref(h)))
        else:       # sfmloc_context_create_sharing: no stream of its own, work goes to `share`'s
            _check(_L().sfmloc_context_create_sharing(m._h, share._h, C.byref(h)))
            self._lender = share
        self._h = h
        m._children.add(self)

    def begin(self, q, view_sel=None):
        if view_sel is None:
            _check(_L().sfmloc_localize_begin(self._h, q._h, None, 0))
        else:
            p, n, keep = _sel(view_sel)
            _check(_L().sfmloc_localize_begin(self._h, q._h, p, n))

    def begin_bow(self, q, bow, knn, cand_views=None):
        """sfmloc_localize_bow_begin: BoW shortlist (knn of the candidate views, all views when None) + the whole
        path on it, asynchronous, the shortlist never leaving the device.  Finish with end()."""
        b = None if bow is None else np.ascontiguousarray(bow, dtype=np.float32).ravel()   # None: q.set_bow()'s
        L = _L()
        L.sfmloc_localize_bow_begin.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint32,
                                                C.POINTER(C.c_uint32), C.c_uint32]
        if cand_views is None:
            _check(L.sfmloc_localize_bow_begin(self._h, q._h, _ptr(b, C.c_float), int(knn), None, 0))
        else:
            p, n, keep = _sel(cand_views)
            _check(L.sfmloc_localize_bow_begin(self._h, q._h, _ptr(b, C.c_float), int(knn), p, n))

    def shard_bow_keys(self, q, knn, keys_dev_ptr, bow=None):
        """sfmloc_shard_bow_keys: this shard's knn best (distance, view id) keys -> device buffer [knn] u64."""
        b = None if bow is None else np.ascontiguousarray(bow, dtype=np.float32).ravel()
        _check(_L().sfmloc_shard_bow_keys(self._h, q._h, _ptr(b, C.c_float), int(knn), C.c_void_p(keys_dev_ptr)))

    def shard_begin_bow(self, q, keys_dev_ptr, n_parts, knn, part_stride_keys=0):
        """sfmloc_shard_begin_bow: global knn best among the gathered key lists -> this shard's views -> K1..K3 +
        candidate emission, all on the device."""
        _check(_L().sfmloc_shard_begin_bow(self._h, q._h, C.c_void_p(keys_dev_ptr), int(n_parts),
                                           int(part_stride_keys), int(knn)))

    def shard_export_packed(self, packed_dev_ptr, n_queries, budget, query_index):
        _check(_L().sfmloc_shard_export_packed(self._h, C.c_void_p(packed_dev_ptr), n_queries, budget, query_index))

    def merge_begin_packed(self, q, packed_dev_ptr, n_parts, n_queries, budget, query_index, part_stride=0):
        _check(_L().sfmloc_merge_begin_packed(self._h, q._h, C.c_void_p(packed_dev_ptr), n_parts, part_stride, n_queries,
                                              budget, query_index))

    def signal(self, hip_stream=0):
        """sfmloc_context_signal: `hip_stream` (a hipStream_t as an integer) waits for this context's queued work."""
        _check(_L().sfmloc_context_signal(self._h, C.c_void_p(hip_stream)))

    def wait(self, hip_stream=0):
        """sfmloc_context_wait: this context waits for the work queued on `hip_stream` so far."""
        _check(_L().sfmloc_context_wait(self._h, C.c_void_p(hip_stream)))

    def shard_begin(self, q, view_sel=None):
        """K1..K3 + candidate emission on this shard (asynchronous)."""
        if view_sel is None:
            _check(_L().sfmloc_shard_begin(self._h, q._h, None, 0))
        else:
            p, n, keep = _sel(view_sel)
            _check(_L().sfmloc_shard_begin(self._h, q._h, p, n))

    def shard_export(self, dst_dev_ptr, cap):
        _check(_L().sfmloc_shard_export(self._h, C.c_void_p(dst_dev_ptr), cap))

    def sync(self):
        _check(_L().sfmloc_context_sync(self._h))

    def merge_begin(self, q, parts_dev_ptr, n_parts, cap, part_stride=0):
        _check(_L().sfmloc_merge_begin(self._h, q._h, C.c_void_p(parts_dev_ptr), n_parts, cap, part_stride))

    def end(self, cap=65536):
        pose = Pose()
        pq = np.zeros(cap, np.uint32)
        pl = np.zeros(cap, np.uint32)
        _check(_L().sfmloc_localize_end(self._h, C.byref(pose), _ptr(pq, C.c_uint32), _ptr(pl, C.c_uint32), cap))
        k = pose.n_inliers if pose.ok else 0
        return pose, pq[:k].copy(), pl[:k].copy()

    def close(self):
        if self._h is not None and self.map._h is not None:
            _L().sfmloc_context_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Query:
    """A query image's descriptors/keypoints resident in HBM (output of extractAKAZESingleImg,
    AKAZEOpenCV.cpp:37-113)."""

    def __init__(self, m, desc, kpt_xy=None, width=0, height=0):
        self._h = None
        self.map = m
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 64)
        self.n = desc.shape[0]
        if kpt_xy is not None:
            kpt_xy = np.ascontiguousarray(kpt_xy, dtype=np.float32).reshape(-1, 2)
            assert kpt_xy.shape[0] == self.n
        h = C.c_void_p()
        _check(_L().sfmloc_query_create(m._h, _ptr(desc, C.c_uint8), _ptr(kpt_xy, C.c_float), self.n,
                                        int(width), int(height), C.byref(h)))
        self._h = h
        m._children.add(self)

    def set_bow(self, bow):
        """sfmloc_query_set_bow: the query's BoW vector becomes resident with it."""
        b = np.ascontiguousarray(bow, dtype=np.float32).ravel()
        _check(_L().sfmloc_query_set_bow(self._h, _ptr(b, C.c_float)))

    @classmethod
    def _over_device_arrays(cls, m, desc_ptr, kpt_ptr, kpt6_ptr, bow_ptr, n, width, height):
        self = cls.__new__(cls)
        self._h = None
        self.map = m
        self.n = int(n)
        h = C.c_void_p()
        _check(_L().sfmloc_query_create_view(m._h, C.c_void_p(desc_ptr), C.c_void_p(kpt_ptr), C.c_void_p(kpt6_ptr),
                                             C.c_void_p(bow_ptr) if bow_ptr else None, int(n), int(width), int(height),
                                             C.byref(h)))
        self._h = h
        m._children.add(self)
        return self

    @classmethod
    def _from_view(cls, m, view_index):
        self = cls.__new__(cls)
        self._h = None
        self.map = m
        self.n = int(m.view_off[view_index + 1] - m.view_off[view_index])
        h = C.c_void_p()
        _check(_L().sfmloc_query_from_view(m._h, view_index, C.byref(h)))
        self._h = h
        m._children.add(self)
        return self

    def close(self):
        if self._h is not None and self.map._h is not None:
            _L().sfmloc_query_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
