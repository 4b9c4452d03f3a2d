"""Host-side mirror of the reference's two callers of the hot path, on top of the C ABI:

* `LocalizeEngine`  -- VisionLocalizeServer/src/LocalizeEngine.{h,cc} (same constructor arguments, same return
  convention: 12 doubles [t(3), R(9 row-major)], empty on failure, 6 stage times);
* `main()`          -- the OpenMVGLocalization_AKAZE command line (localization.cpp:64-82,155-587): same positional
  arguments and keys, writes the same <outDir>/<basename>.json files the Python orchestration consumes
  (mergeSfM.py:50-66: success <=> "t" in json).

Feature extraction (A2, extractAKAZESingleImg) is not on the GPU yet: a query image `foo.jpg` is localised from
precomputed `foo.desc` / `foo.feat` next to it (or in --featdir), which is what the reference's own extractor
writes (AKAZEOpenCV.cpp:80-81, FileUtils.cpp:77-92).
"""
import os
import sys
import time

import numpy as np

from . import capi, fileio

IMAGE_EXT = ("jpg", "JPG", "jpeg", "JPEG", "png", "PNG")  # localization.cpp:204-206


def _image_size(path, default):
    try:
        return capi.image_size(path)
    except capi.SfmlocError:
        return default


def _load_gray(path):
    """imread(filename, IMREAD_GRAYSCALE) (AKAZEOpenCV.cpp:60) = sfmloc_image_read; None when not decodable."""
    try:
        return capi.image_read(path, color=False)
    except capi.SfmlocError:
        return None


def _load_bgr(path):
    """imread(filename, IMREAD_COLOR) (DenseLocalFeatureWrapper.cpp:85): h x w x 3, B G R order."""
    try:
        return capi.image_read(path, color=True)
    except capi.SfmlocError:
        return None


def synth_round6(a):
    a = np.asarray(a, dtype=np.float32)
    return np.array([np.float32(float("%.6g" % float(v))) for v in a.ravel()], dtype=np.float32).reshape(a.shape)


class LocalizeEngine:
    """LocalizeEngine(sfmDataDir, matchDir, AmatFile, secondTestRatio, ransacRound, ransacPrecision,
    guidedMatching, beaconKnnNum=0, bowKnnNum=0)   -- LocalizeEngine.h:75-77."""

    def __init__(self, sfm_data_dir, match_dir, amat_file=None, second_test_ratio=0.6, ransac_round=25,
                 ransac_precision=4.0, guided_matching=False, beacon_knn_num=0, bow_knn_num=0, device=0, profile=0):
        if beacon_knn_num:
            raise NotImplementedError("iBeacon view pre-selection is out of scope (SURVEY.md 2.1)")
        self.sfm_data_dir, self.match_dir = sfm_data_dir, match_dir
        self.params = capi.default_params(dist_ratio=second_test_ratio, ransac_round=ransac_round,
                                          geom_precision=ransac_precision, bow_knn=bow_knn_num, device=device,
                                          profile=profile, guided_matching=int(bool(guided_matching)))
        # sfm_data_dir may also name a packed map file written by capi.pack / sfmloc_pack
        self.map = (capi.Map.open_packed(sfm_data_dir, self.params) if os.path.isfile(sfm_data_dir)
                    else capi.Map.open(sfm_data_dir, match_dir, self.params))
        self.A = None
        if amat_file:
            y = fileio.read_cv_yaml(amat_file)           # LocalizeEngine.cc:116-118
            self.A = np.asarray(y["A"], dtype=np.float64).reshape(3, 4)

    def close(self):
        for ak in getattr(self, "_akaze", {}).values():
            ak.close()
        self.map.close()

    def extract(self, gray):
        """extractAKAZESingleImg's compute part (AKAZEOpenCV.cpp:44-46,67) on the GPU, with the map's
        image_describer.txt options (AKAZEOption.cpp:44-55): -> desc [n x 64], kpts [n x 4] (x, y, size, angle)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        if not hasattr(self, "_akaze"):
            self._akaze = {}
            self._akaze_opt = fileio.read_image_describer(os.path.join(self.match_dir, "image_describer.txt"))
            if int(self._akaze_opt["desc_ch"]) != 3:
                raise NotImplementedError("only the 3-channel M-LDB descriptor the reference uses is implemented")
        if (w, h) not in self._akaze:
            o = self._akaze_opt
            self._akaze[(w, h)] = capi.Akaze(w, h, int(o["nOct"]), int(o["nOctLay"]), float(o["thres"]),
                                             device=int(self.params.device))
        kp, desc = self._akaze[(w, h)].detect_and_compute(gray)
        return desc, kp[:, :4]

    def localize_image(self, gray, **kw):
        """LocalizeEngine::localize on an image (LocalizeEngine.cc:288-661): extract, then localize().  A colour
        image (h x w x 3, B G R -- what the server decodes, localizeImage.cc:393/463) goes through cvtColor BGR2GRAY
        first, as cv::AKAZE::detectAndCompute does with a multi-channel input."""
        gray = np.asarray(gray)
        if gray.ndim == 3:
            b, g, r = (gray[:, :, i].astype(np.int32) for i in range(3))
            gray = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
        t_feat = time.perf_counter()
        desc, kp = self.extract(gray)
        t_feat = time.perf_counter() - t_feat
        h, w = gray.shape
        res, ex = self.localize(desc, kp[:, :2], w, h, **kw)
        ex["n_features"] = len(desc)
        if "times" in ex:
            ex["times"][2] = t_feat
        return res, ex

    def localize_file(self, path, **kw):
        """The server's entry point on a stored image (localizeImage.cc:463: imread(path, IMREAD_COLOR) -> localize)."""
        return self.localize_image(capi.image_read(path, color=True), **kw)

    # getLocalViews (SfMDataUtils.cpp:210-227 / LocalizeEngine.cc:200-): NOTE the reference compares the SQUARED
    # distance with the un-squared radius; reproduced.
    def local_views(self, center, radius):
        c = self.map.view_center
        if self.A is not None:  # the engine works in the A-transformed frame (TRANSFORM_SFM_DATA_BEFORE_LOCALIZE)
            c = c @ self.A[:, :3].T + self.A[:, 3]
        d2 = ((c - np.asarray(center, np.float64)[None, :]) ** 2).sum(1)
        return np.nonzero(d2 <= radius)[0].astype(np.uint32)

    def localize(self, desc, kpt_xy, width, height, return_keypoints=False, return_time=False, center=None,
                 radius=-1.0, bow=None):
        """-> result (list of 12 doubles or []), extras dict.  LocalizeEngine.cc:288-661 from `endFeat` on.
        bow: the query's BoW vector (calcBoF output, 500 doubles); with bow_knn_num > 0 it shortlists the views
        (LocalizeEngine.cc:333-361) -- only when more views than knn remain."""
        t0 = time.perf_counter()
        view_sel = None
        if center is not None and len(center) == 3 and radius > 0:
            view_sel = self.local_views(center, radius)
            if len(view_sel) == 0:
                return [], {}
        knn = int(self.params.bow_knn)
        q = self.map.query(desc, kpt_xy, width, height)
        prof = int(self.map.params.profile)
        if return_time and prof != 1:
            self.map.set_profile(1)          # the reference's `times` come from this query's per-stage events
        try:
            if bow is not None and knn > 0:
                # the shortlist applies when more than knn views remain (localization.cpp:346 / LocalizeEngine.cc:342);
                # shortlist and path are one call, the selected views never leave the device
                pose, pq, pl = self.map.localize_bow(q, np.asarray(bow, np.float64).astype(np.float32), knn, view_sel)
            else:
                pose, pq, pl = self.map.localize(q, view_sel)
        finally:
            q.close()
            if return_time and prof != 1:
                self.map.set_profile(prof)
        extras = {"pose": pose, "pairs": list(zip(pq.tolist(), pl.tolist()))}
        if return_time:
            # LocalizeEngine.cc:651-657: selectBeacon (no iBeacon stage here), selectBow, extFeat (localize_image fills
            # it), putMatch, geoMatch, PnP
            ss = list(pose.stage_seconds)
            extras["times"] = [0.0, ss[1], 0.0, ss[3], ss[4], ss[5]]
            extras["time_total"] = time.perf_counter() - t0
        if not pose.ok:
            return [], extras
        R = np.array(pose.R).reshape(3, 3)
        c = np.array(pose.center)
        if self.A is not None:
            # localising in the A-transformed map (LocalizeEngine.cc:121-144) = transforming the result
            A3 = self.A[:, :3]
            s = np.cbrt(np.linalg.det(A3))
            c = A3 @ c + self.A[:, 3]
            R = R @ (A3 / s).T
        return list(c) + list(R.ravel()), extras


class UserCamera:
    """What execLocalizeImage keeps per user (localizeImage.cc:133-168): the user's K and dist (OpenCV YAML files with
    the keys "K" and "dist"), and from them the new camera matrix, the valid region and the undistortion maps."""

    def __init__(self, k_mat_file, dist_mat_file, width, height, device=0):
        self.K = np.asarray(fileio.read_cv_yaml(k_mat_file)["K"], np.float64).reshape(3, 3)
        self.dist = np.asarray(fileio.read_cv_yaml(dist_mat_file)["dist"], np.float64).ravel()
        self.undistorter = capi.Undistorter(self.K, self.dist, width, height, device=device)

    def close(self):
        self.undistorter.close()


def exec_localize_image(engine, camera, bgr, **kw):
    """execLocalizeImage (localizeImage.cc:61-200) from the decoded image on: undistort with the user's camera, crop to
    the valid region, LocalizeEngine::localize.  -> (result, extras) as LocalizeEngine.localize_image."""
    und = camera.undistorter.apply(bgr)          # cv::undistort + undistortImage(validRoi)
    return engine.localize_image(und, **kw)


def dense_grid_keypoints(size=300, step=6, levels=4, init_scale=4.0, scale_mul=1.5, bound=0):
    """DenseFeatureDetector::detectImpl (BoWCommon/src/DenseFeatureDetector.cpp:44-69) with the constants of
    DenseLocalFeatureWrapper.h:32-38: per scale level s a regular grid (x fastest), size = init_scale * mul^s,
    class_id = s (which cv::AKAZE::compute then reads as the evolution level).  -> [n, 4] f32 (x, y, size, class_id)"""
    out = []
    fs = np.float32(init_scale)
    for s in range(levels):
        for y in range(bound, size - bound, step):
            for x in range(bound, size - bound, step):
                out.append((x, y, fs, s))
        fs = np.float32(fs * np.float32(scale_mul))
    return np.array(out, np.float32).reshape(-1, 4)


class DenseBow:
    """The query-side BoW vector of the reference (DenseLocalFeatureWrapper::calcDenseLocalFeature ->
    PcaWrapper::calcPcaProject -> BoFSpatialPyramids::calcBoF; LocalizeEngine.cc:205-232): colour image ->
    300x300 gray (sfmloc_dense_gray) -> dense-grid AKAZE descriptors (sfmloc_akaze_compute, cv::AKAZE defaults) as
    float32 -> PCA + BoF (sfmloc_bof_compute)."""

    def __init__(self, bow_file, pca_file=None, device=0):
        self.bof = capi.BofModel.from_files(bow_file, pca_file, in_dim=61, device=device)
        self.size = int(self.bof.resized) if hasattr(self.bof, "resized") else 300
        self.device = device
        self.akaze = capi.Akaze(self.size, self.size, 4, 4, 0.001, device=device)   # cv::AKAZE::create() defaults
        self.grid = dense_grid_keypoints(self.size)
        self._files = (bow_file, pca_file)
        self._resident = {}     # (w, h): capi.ImgBow -- the same chain without the host round trips (sfmloc_imgbow)

    def local_features(self, bgr):
        gray = capi.dense_gray(bgr, self.size, device=self.device)
        desc, _ = self.akaze.compute(gray, self.grid)
        return desc[:, :61].astype(np.float32), self.grid[:, :2].copy(), gray      # convertTo(CV_32FC1)

    def compute(self, bgr, staged=False):
        """-> the float64 BoF vector.  staged=True: the three stage-level calls (sfmloc_dense_gray, sfmloc_akaze_compute,
        sfmloc_bof_compute); default: sfmloc_imgbow, one resident chain per image size -- the same bits."""
        if staged:
            feats, kxy, _ = self.local_features(bgr)
            return self.bof.compute(feats, kxy)
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w = bgr.shape[:2]
        ib = self._resident.get((w, h))
        if ib is None:
            ib = self._resident[(w, h)] = capi.ImgBow.from_files(self._files[0], self._files[1], w, h, 3, device=self.device)
        return ib.compute(bgr)

    def close(self):
        for ib in self._resident.values():
            ib.close()
        self._resident = {}
        self.akaze.close()
        self.bof.close()


def parse_cv_args(argv, spec):
    """cv::CommandLineParser syntax: positional arguments and -k=v / --key=v / bare flags."""
    pos, opts = [], {}
    for a in argv:
        if a.startswith("-") and not _is_number(a):
            k, eq, v = a.lstrip("-").partition("=")
            opts[k] = v if eq else "true"
        else:
            pos.append(a)
    out = {}
    for names, default, conv in spec:
        val = default
        for n in names:
            if n in opts:
                val = opts[n]
        out[names[-1]] = conv(val)
    return pos, out


def _is_number(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def _b(v):
    return str(v).lower() in ("1", "true", "yes")


KEYS = [  # localization.cpp:64-82
    (("f", "fDistRatio"), "0.6", float), (("r", "ransacRound"), "200", int), (("w", "writematch"), "false", _b),
    (("k", "knnbow"), "0", int), (("x", "cenLocX"), "0.0", float), (("y", "cenLocY"), "0.0", float),
    (("z", "cenLocZ"), "0.0", float), (("d", "cenRadius"), "-1.0", float), (("a", "bowModelFile"), "", str),
    (("p", "pcaModelFile"), "", str), (("i", "locEvryNFrame"), "1", int), (("g", "geomLimit"), "4.0", float),
    (("gm", "guidedMatch"), "false", _b), (("featdir",), "", str), (("device",), "0", int),
]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    pos, o = parse_cv_args(argv, KEYS)
    if len(pos) < 4 or "h" in argv or "--help" in argv:
        print("usage: localize <queryImage|dir> <sfmDataDir> <matchDir> <outputFolder> [-f=0.6] [-r=200] [-k=0] "
              "[-x= -y= -z= -d=-1] [-i=1] [-g=4.0] [--featdir=DIR]")
        return 1
    query, sfm_dir, match_dir, out_dir = pos[:4]
    print("Start localizing input image.")
    if os.path.isfile(query):
        if query.rsplit(".", 1)[-1] not in IMAGE_EXT:
            print("Input image is not JPEG or PNG file")
            return 1
        images = [query]
    elif os.path.isdir(query):
        images = [os.path.join(query, f) for f in sorted(os.listdir(query)) if f.rsplit(".", 1)[-1] in IMAGE_EXT]
        if not images:
            print("JPEG or PNG file is not found in input image directory")
            return 1
    else:
        # the reference would fail in imread; with precomputed features the image itself may be absent
        images = [query]
    packed = os.path.isfile(sfm_dir)            # a packed map file (capi.pack) in place of the sfm_data directory
    sfm_json = sfm_dir if packed else os.path.join(sfm_dir, "sfm_data.json")
    try:
        eng = LocalizeEngine(sfm_dir, match_dir, None, o["fDistRatio"], o["ransacRound"], o["geomLimit"],
                             o["guidedMatch"], 0, o["knnbow"], device=o["device"])
    except capi.SfmlocError as e:
        print(str(e), file=sys.stderr)
        return 1
    rc_all = 0
    # the query's view index in the reference's match files: id of the LAST view of sfm_data + 1 (localization.cpp:371)
    ind_query_file = int(eng.map.view_id[-1]) + 1 if eng.map.n_views else 0
    if packed:
        wh0 = eng.map.view_sizes()[0]
        default_wh = (int(wh0[0]), int(wh0[1]))
    else:
        sd = fileio.read_sfm_data(sfm_json)
        v0 = sd["views"][0]["value"]["ptr_wrapper"]["data"]
        default_wh = (int(v0["width"]), int(v0["height"]))
        ind_query_file = max(int(v["value"]["ptr_wrapper"]["data"]["id_view"]) for v in sd["views"]) + 1
    every = o["locEvryNFrame"] if o["locEvryNFrame"] > 0 else 1
    os.makedirs(out_dir, exist_ok=True)
    n_img, match_next = 0, 0
    dense = None
    for img in images:
        n_img += 1
        if n_img % every == 0:      # localization.cpp:289-298
            pass
        elif match_next <= 0:
            continue
        else:
            match_next -= 1
        base = os.path.splitext(os.path.basename(img))[0]
        fdir = o["featdir"] or os.path.dirname(img)
        w, h = _image_size(img, default_wh)
        try:
            desc = fileio.read_desc(os.path.join(fdir, base + ".desc"))
            feat = fileio.read_feat(os.path.join(fdir, base + ".feat"))
        except (IOError, OSError) as e:
            gray = _load_gray(img)
            if gray is None:
                print(f"cannot read {img} nor precomputed features for it: {e}", file=sys.stderr)
                fileio.write_result_json(out_dir, img, sfm_json, match_dir)
                continue
            print("Extract features from query image")          # localization.cpp:313
            desc, feat = eng.extract(gray)
            feat = synth_round6(feat)                            # what the reference reads back from its .feat
            h, w = gray.shape
        center = (o["cenLocX"], o["cenLocY"], o["cenLocZ"]) if o["cenRadius"] > 0 else None
        bow = None
        if o["knnbow"] > 0 and o["bowModelFile"]:
            # the query's BoW vector (localization.cpp:346-361): a precomputed <base>.bow next to its features if
            # there is one (what TrainBoW's calcBoF writes per view, TrainBoW.cpp:256-271), else computed from the
            # colour image as the reference does (DenseLocalFeatureWrapper -> PcaWrapper -> BoFSpatialPyramids)
            bpath = os.path.join(fdir, base + ".bow")
            if os.path.exists(bpath):
                bow = fileio.read_mat_bin(bpath).ravel()
            else:
                bgr = _load_bgr(img)
                if bgr is not None:
                    if dense is None:
                        dense = DenseBow(o["bowModelFile"], o["pcaModelFile"] or None, device=o["device"])
                    bow = dense.compute(bgr)
        try:
            res, ex = eng.localize(desc, feat[:, :2], w, h, center=center, radius=o["cenRadius"], bow=bow)
        except capi.SfmlocError as e:
            # an error on ONE image does not end the run: the reference writes a result file for every image
            # (localization.cpp:441,530); this one gets the failure form and the exit status remembers it
            print(f"{img}: {e}", file=sys.stderr)
            rc_all = 1
            fileio.write_result_json(out_dir, img, sfm_json, match_dir)
            continue
        pose = ex.get("pose")
        if pose is None:                                # no view near the given position: nothing was matched
            print("Not enough putative matches")
            fileio.write_result_json(out_dir, img, sfm_json, match_dir)
            continue
        print(f"number of putative matches : {pose.n_putative_views}")      # localization.cpp:416
        if pose.n_putative_views == 0:
            print("Not enough putative matches")                             # :420
            fileio.write_result_json(out_dir, img, sfm_json, match_dir)
            continue
        if o["writematch"]:
            # -w: exportPairWiseMatches(map_geometricMatches, <matchDir>/matches.fQ.txt) (localization.cpp:452-455);
            # the putative list goes to a per-query folder the reference deletes again (:399-403, :585).  Pair =
            # (view id, id of the last view of sfm_data + 1), matches in AC-RANSAC's inlier order.
            # (with -gm the matches are the guided ones, in ascending map-feature order)
            gc, gi, gj = eng.map.geometric_read_pairs()
            geo = {}
            for v in np.nonzero(gc)[0]:
                o0 = int(eng.map.view_off[v])
                geo[(int(eng.map.view_id[v]), ind_query_file)] = (gi[o0:o0 + int(gc[v])], gj[o0:o0 + int(gc[v])])
            fileio.write_matches_txt(os.path.join(match_dir, "matches.fQ.txt"), geo)
        print(f"number of geometric matches : {pose.n_geometric_views}")     # :458
        print(f"mapFeatTo3DFeat size = {pose.n_matches_2d3d}")               # :476
        print(f"cpt = {pose.n_matches_2d3d}")                                # :502
        if not res:
            print("Fail to estimate camera matrix")                          # :512
            print(f"#inliers = {pose.n_inliers}")
            fileio.write_result_json(out_dir, img, sfm_json, match_dir)
            continue
        print(f"#inliers = {pose.n_inliers}")
        fileio.write_result_json(out_dir, img, sfm_json, match_dir, K=np.array(pose.K), R=np.array(pose.R),
                                 center=np.array(pose.center), pairs=ex["pairs"])
        match_next = every - 1
        print("complete")
    eng.close()
    if dense is not None:
        dense.close()
    return rc_all


if __name__ == "__main__":
    sys.exit(main())
