"""Host-side mirror of the reference's `ExtFeatAndMatch` tool (ExtFeatAndMatch/src/computeFeaturesAndMatches.cpp:49-247),
the map-building client of the same C ABI:

    ExtFeatAndMatch <matchDir> [-c=3 -t=0.001 -o=4 -l=4 -f=0.6 -r=4096 -v=0 -p= -mf=0 -mm=60 -g=4.0 -gm -sm]

    <matchDir>/sfm_data.json  (views + intrinsics)      ->  image_describer.txt, <base>.feat / <base>.desc per view,
                                                            matches.putative.txt, matches.f.txt

extractAKAZE (AKAZEOpenCV.cpp:116-187) = sfmloc_akaze_detect_and_compute per image; matchAKAZE / trackAKAZE
(MatchUtils.cpp:73-277) = sfmloc_match_pairs / sfmloc_track; geometricMatch (MatchUtils.cpp:372-420) =
sfmloc_geometric_pairs, with -gm under guided matching (the map builder's default, reconstructGraph.py:155-163).
Files that already exist are kept, as in the reference.
"""
import os
import sys

import numpy as np

from . import capi, fileio
from .engine import _b, _load_gray, parse_cv_args

KEYS = [  # computeFeaturesAndMatches.cpp:49-64
    (("c", "akazeChannel"), "3", int), (("t", "akazeThreshold"), "0.001", float), (("o", "akazeNOctave"), "4", int),
    (("l", "akazeOctaveLayer"), "4", int), (("f", "fdistratio"), "0.6", float), (("r", "ransacround"), "4096", int),
    (("v", "videoMatchFrame"), "0", int), (("p", "pairfile"), "", str), (("mf", "maxFrameDist"), "0", int),
    (("mm", "minMatch"), "60", int), (("g", "geomError"), "4.0", lambda v: int(float(v))),   # parsed as int (:92)
    (("gm", "guidedMatch"), "false", _b), (("sm", "skipMathing"), "false", _b), (("device",), "0", int),
]


def generate_all_pairs(view_ids):
    """hulo::generateAllPairs (SfMDataUtils.cpp:128-141)"""
    return [(view_ids[i], view_ids[j]) for i in range(len(view_ids)) for j in range(i + 1, len(view_ids))]


def generate_video_match_pairs(view_ids, frame):
    """hulo::generateVideoMatchPairs (SfMDataUtils.cpp:144-157)"""
    n = len(view_ids)
    return [(view_ids[i], view_ids[j]) for i in range(n) for j in range(i + 1, min(n, i + frame + 1))]


def remove_dup_pairs(pairs):
    """hulo::removeDupPairs (SfMDataUtils.cpp:168-187), literally: scanning from the back, earlier entries get
    ordered (first <= second) as a side effect, a later duplicate of an earlier pair is dropped."""
    pairs = [tuple(p) for p in pairs]
    dup = []
    for i in range(len(pairs) - 1, 0, -1):
        for j in range(i - 1, -1, -1):
            if pairs[j][0] > pairs[j][1]:
                pairs[j] = (pairs[j][1], pairs[j][0])
            if pairs[i] == pairs[j] or pairs[i] == (pairs[j][1], pairs[j][0]):
                dup.append(i)
                break
    for i in dup:          # descending indices: erasing one does not shift the ones still to erase
        del pairs[i]
    return pairs


def read_pair_file(path):
    """hulo::readPairFile (FileUtils.cpp:180-194)"""
    out = []
    try:
        with open(path) as fh:
            for line in fh:
                t = line.split()
                if len(t) >= 2:
                    out.append((int(t[0]), int(t[1])))
    except OSError:
        pass
    return out


def _views(sd, match_dir):
    root = sd.get("root_path", "")
    out = []
    for v in sorted(sd["views"], key=lambda e: e["key"]):
        d = v["value"]["ptr_wrapper"]["data"]
        base, ext = os.path.splitext(os.path.basename(d["filename"]))
        out.append({"id": int(d["id_view"]), "image": os.path.join(root, base + ext), "w": int(d["width"]),
                    "h": int(d["height"]), "feat": os.path.join(match_dir, base + ".feat"),
                    "desc": os.path.join(match_dir, base + ".desc")})
    return out


def extract_akaze(views, opt, device=0):
    """hulo::extractAKAZE (AKAZEOpenCV.cpp:116-187): features of every view whose .feat or .desc is missing."""
    extractors = {}
    for v in views:
        if os.path.exists(v["feat"]) and os.path.exists(v["desc"]):
            continue
        gray = _load_gray(v["image"])
        if gray is None:
            print(f"cannot open file to write features for {v['image']}", file=sys.stderr)
            continue
        h, w = gray.shape
        if (w, h) not in extractors:
            extractors[(w, h)] = capi.Akaze(w, h, opt["nOct"], opt["nOctLay"], opt["thres"], device=device)
        kp, desc = extractors[(w, h)].detect_and_compute(gray)
        fileio.write_feat(v["feat"], kp[:, :4])          # x y size angle, ostream << float
        fileio.write_desc(v["desc"], desc)
    for a in extractors.values():
        a.close()


def _load_bank(views):
    descs, kps = [], []
    for v in views:
        descs.append(fileio.read_desc(v["desc"]))
        kp = fileio.read_feat(v["feat"])
        kps.append(kp[:, :2] if len(kp) else np.zeros((0, 2), np.float32))
    off = np.zeros(len(views) + 1, np.uint32)
    off[1:] = np.cumsum([len(d) for d in descs])
    desc = np.concatenate(descs) if descs else np.zeros((0, 64), np.uint8)
    kpt = np.concatenate(kps).astype(np.float32) if kps else np.zeros((0, 2), np.float32)
    return off, desc, kpt


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    pos, o = parse_cv_args(argv, KEYS)
    if len(pos) < 1 or "h" in argv or "--help" in argv:
        print("usage: ExtFeatAndMatch <matchdir> [-c=3 -t=0.001 -o=4 -l=4 -f=0.6 -r=4096 -v=0 -p= -mf=0 -mm=60 -g=4.0 "
              "-gm -sm]")
        return 1
    match_dir = pos[0]
    have = (bool(o["pairfile"]), o["videoMatchFrame"] > 0, o["maxFrameDist"] > 0)
    if sum(have) > 1:      # CV_Assert at :103-106
        print("pair file, video match frames and track length are mutually exclusive", file=sys.stderr)
        return 1
    print(f"Matches directory : {match_dir}")
    sd_path = os.path.join(match_dir, "sfm_data.json")
    try:
        sd = fileio.read_sfm_data(sd_path)
    except (OSError, ValueError):
        print(f"Cannot load {sd_path}", file=sys.stderr)
        return 1
    views = _views(sd, match_dir)
    ids = [v["id"] for v in views]
    if o["akazeChannel"] != 3:
        print("only 3-channel M-LDB (486 bits) is implemented", file=sys.stderr)
        return 1
    fileio.write_image_describer(os.path.join(match_dir, "image_describer.txt"), o["akazeChannel"], o["akazeThreshold"],
                                 o["akazeNOctave"], o["akazeOctaveLayer"])
    extract_akaze(views, {"nOct": o["akazeNOctave"], "nOctLay": o["akazeOctaveLayer"], "thres": o["akazeThreshold"]},
                  device=o["device"])
    if o["skipMathing"]:
        print("Skip matching option is set. Exit without feature matching.")
        return 1          # the reference returns 1 here (:147)

    off, desc, kpt = _load_bank(views)
    wh = np.array([[v["w"], v["h"]] for v in views], np.uint32)
    idx_of = {vid: k for k, vid in enumerate(ids)}
    params = capi.default_params(dist_ratio=o["fdistratio"], ransac_round=o["ransacround"],
                                 geom_precision=float(o["geomError"]), device=o["device"],
                                 guided_matching=int(bool(o["guidedMatch"])))       # -gm: MatchUtils.cpp:413-415
    put_path = os.path.join(match_dir, "matches.putative.txt")
    f_path = os.path.join(match_dir, "matches.f.txt")
    with capi.Map(np.array(ids, np.uint32), off, desc, params=params, view_wh=wh, kpt_xy=kpt) as dm:
        print("Start Putative Matching...")
        if not os.path.exists(put_path):
            if o["maxFrameDist"] != 0:
                m = dm.track(o["maxFrameDist"])
            else:
                print("Generating pairs")
                if o["pairfile"]:
                    pairs = read_pair_file(o["pairfile"])
                elif o["videoMatchFrame"] > 0:
                    pairs = remove_dup_pairs(generate_video_match_pairs(ids, o["videoMatchFrame"]))
                else:
                    pairs = generate_all_pairs(ids)
                print(" ".join(f"({a} {b})" for a, b in pairs))
                print(f"Total number of pairs : {len(pairs)}")
                m = dm.match_pairs([(idx_of[a], idx_of[b]) for a, b in pairs if a in idx_of and b in idx_of])
            fileio.write_matches_txt(put_path, {(ids[a], ids[b]): v for (a, b), v in m.items()})
        print("Start Geometric Matching...")
        if not os.path.exists(f_path):
            put = fileio.read_matches_txt(put_path)
            kept = {}
            for (a, b), v in put.items():
                if len(v[0]) < o["minMatch"]:          # :211-221
                    print(f"{a},{b},{len(v[0])} ", end="")
                else:
                    kept[(idx_of[a], idx_of[b])] = v
            print()
            g = dm.geometric_pairs(kept)
            print(f"number of putative matches : {len(kept)}")
            print(f"number of geometric matches : {len(g)}")
            fileio.write_matches_txt(f_path, {(ids[a], ids[b]): v for (a, b), v in g.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
