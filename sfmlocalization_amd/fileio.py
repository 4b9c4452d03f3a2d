"""The on-disk contract of the localisation path, byte for byte.

Readers/writers for what ExtFeatAndMatch / TrainBoW / OpenMVG leave in <matchDir> and <sfmDir>, and for what
OpenMVGLocalization_AKAZE writes to <outDir>.  Pure Python + NumPy; used by the CLI / engine mirror and by the
tests that build toy maps on disk.  Citations are relative to /root/reference.
"""
import json
import os
import struct

import numpy as np

CV_TYPES = {0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}
CV_CODES = {np.dtype(v): k for k, v in CV_TYPES.items()}


# ---- <base>.desc : u64 N (native size_t) + N x 64 B  (FileUtils.cpp:77-103, OpenMVG saveDescsToBinFile) ----
def write_desc(path, desc):
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    if desc.ndim != 2 or desc.shape[1] not in (61, 64):
        raise ValueError("descriptors must be N x 61 (M-LDB) or N x 64 (padded)")
    if desc.shape[1] == 61:  # saveAKAZEBin pads 61 -> 64 with zeros
        desc = np.concatenate([desc, np.zeros((desc.shape[0], 3), np.uint8)], axis=1)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", desc.shape[0]))
        f.write(desc.tobytes())


def read_desc(path):
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise IOError(f"{path}: truncated .desc header")
        (n,) = struct.unpack("<Q", head)
        data = np.fromfile(f, dtype=np.uint8, count=n * 64)
    if data.size != n * 64:
        raise IOError(f"{path}: expected {n} descriptors, file holds {data.size // 64}")
    return data.reshape(n, 64)


# ---- <base>.feat : text "x y size angle" per line, default ostream precision (AKAZEOpenCV.cpp:80-81) ----
def _g6(v):
    return "%.6g" % float(v)


def write_feat(path, kpts):
    """kpts: N x 4 (x, y, size, angle) or N x 2 (size/angle written as 0)."""
    kpts = np.asarray(kpts, dtype=np.float32)
    if kpts.ndim != 2 or kpts.shape[1] not in (2, 4):
        raise ValueError("keypoints must be N x 2 or N x 4")
    with open(path, "w") as f:
        for r in kpts:
            if kpts.shape[1] == 4:
                f.write(f"{_g6(r[0])} {_g6(r[1])} {_g6(r[2])} {_g6(r[3])}\n")
            else:
                f.write(f"{_g6(r[0])} {_g6(r[1])} 0 0\n")


def read_feat(path):
    """-> N x 4 float32 (FileUtils.cpp:151-177 reads x y and skips the other two)."""
    rows = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 4:
                break  # `fileRead >> a >> b >> c >> d` stops at the first short line
            rows.append([np.float32(p[0]), np.float32(p[1]), np.float32(p[2]), np.float32(p[3])])
    return np.array(rows, dtype=np.float32).reshape(-1, 4)


# ---- <base>.bow and friends : int32 rows, cols, cvType + raw data (FileUtils.cpp:43-75) ----
def write_mat_bin(path, mat):
    with open(path, "wb") as f:
        if mat is None or np.size(mat) == 0:
            f.write(struct.pack("<i", 0))
            return
        mat = np.ascontiguousarray(mat)
        if mat.ndim == 1:
            mat = mat.reshape(-1, 1)
        code = CV_CODES.get(mat.dtype)
        if code is None:
            raise ValueError(f"unsupported dtype {mat.dtype}")
        f.write(struct.pack("<iii", mat.shape[0], mat.shape[1], code))
        f.write(mat.tobytes())


def read_mat_bin(path):
    with open(path, "rb") as f:
        (rows,) = struct.unpack("<i", f.read(4))
        if rows == 0:
            return np.zeros((0, 0))
        cols, code = struct.unpack("<ii", f.read(8))
        if code not in CV_TYPES:
            raise IOError(f"{path}: unknown cv type {code}")
        data = np.fromfile(f, dtype=CV_TYPES[code], count=rows * cols)
    if data.size != rows * cols:
        raise IOError(f"{path}: truncated matrix")
    return data.reshape(rows, cols)


# ---- OpenCV FileStorage YAML (image_describer.txt, BOWfile.yml, PCAfile.yml, Amat.yml) ----
def write_cv_yaml(path, items):
    """items: ordered dict name -> int | float | str | 2-D ndarray (written as !!opencv-matrix)."""
    with open(path, "w") as f:
        f.write("%YAML:1.0\n")
        for k, v in items.items():
            if isinstance(v, np.ndarray):
                v2 = np.atleast_2d(v)
                dt = {np.dtype(np.float32): "f", np.dtype(np.float64): "d", np.dtype(np.int32): "i",
                      np.dtype(np.uint8): "u"}[v2.dtype]
                f.write(f"{k}: !!opencv-matrix\n   rows: {v2.shape[0]}\n   cols: {v2.shape[1]}\n   dt: {dt}\n")
                flat = ", ".join(repr(float(x)) if dt in "fd" else str(int(x)) for x in v2.ravel())
                f.write(f"   data: [ {flat} ]\n")
            elif isinstance(v, bool):
                f.write(f"{k}: {int(v)}\n")
            elif isinstance(v, (int, np.integer)):
                f.write(f"{k}: {int(v)}\n")
            elif isinstance(v, (float, np.floating)):
                f.write(f"{k}: {float(v)!r}\n")
            else:
                f.write(f'{k}: "{v}"\n')


def read_cv_yaml(path):
    """Minimal reader for the subset OpenCV writes for these files (scalars, strings, opencv-matrix)."""
    out = {}
    with open(path) as f:
        lines = [ln.rstrip("\n") for ln in f]
    i = 0
    while i < len(lines):
        ln = lines[i]
        i += 1
        if not ln.strip() or ln.startswith("%") or ln.startswith("---"):
            continue
        if ":" not in ln or ln.startswith(" "):
            continue
        key, _, rest = ln.partition(":")
        key, rest = key.strip(), rest.strip()
        if rest.startswith("!!opencv-matrix"):
            rows = cols = None
            dt = "d"
            buf = ""
            while i < len(lines) and (lines[i].startswith(" ") or lines[i].startswith("\t")):
                s = lines[i].strip()
                i += 1
                if s.startswith("rows:"):
                    rows = int(s.split(":")[1])
                elif s.startswith("cols:"):
                    cols = int(s.split(":")[1])
                elif s.startswith("dt:"):
                    dt = s.split(":")[1].strip()
                elif s.startswith("data:"):
                    buf = s.split(":", 1)[1]
                else:
                    buf += " " + s
            vals = [v for v in buf.replace("[", " ").replace("]", " ").replace(",", " ").split()]
            dtype = {"f": np.float32, "d": np.float64, "i": np.int32, "u": np.uint8}[dt[-1]]
            out[key] = np.array([float(v) for v in vals]).astype(dtype).reshape(rows, cols)
        elif rest.startswith('"'):
            out[key] = rest.strip('"')
        else:
            try:
                out[key] = int(rest)
            except ValueError:
                try:
                    out[key] = float(rest)
                except ValueError:
                    out[key] = rest
    return out


def write_image_describer(path, desc_ch=3, thres=0.001, n_oct=4, n_oct_lay=4):
    """AKAZEOption::write (AKAZEOption.cpp:31-41)."""
    write_cv_yaml(path, {"desc_ch": int(desc_ch), "thres": float(thres), "nOct": int(n_oct), "nOctLay": int(n_oct_lay)})


def read_image_describer(path):
    """AKAZEOption::read (AKAZEOption.cpp:44-55); defaults AKAZEOption.h:31-34."""
    d = {"desc_ch": 3, "thres": 0.001, "nOct": 4, "nOctLay": 4}
    if os.path.exists(path):
        d.update({k: v for k, v in read_cv_yaml(path).items() if k in d})
    return d


# ---- sfm_data.json (cereal layout; reconstructGraph.py:121-148, mergeSfM.py:89-108) ----
def make_sfm_data(view_ids, filenames, width, height, focal, ppx, ppy, poses=None, structure=None,
                  root_path="", intrinsic_type="pinhole", disto_k3=None):
    """poses: dict view_id -> (R 3x3, C 3); structure: list of (landmark_id, X, [(view_id, id_feat, (u, v))])."""
    views = []
    for k, (vid, fn) in enumerate(zip(view_ids, filenames)):
        views.append({"key": int(vid), "value": {
            "polymorphic_id": 1073741824 if k else 2147483649,
            "ptr_wrapper": {"id": 2147483649 + k, "data": {
                "local_path": "/", "filename": fn, "width": int(width), "height": int(height),
                "id_view": int(vid), "id_intrinsic": 0, "id_pose": int(vid)}}}})
        if k == 0:
            views[0]["value"]["polymorphic_name"] = "view"
    data = {"width": int(width), "height": int(height), "focal_length": float(focal),
            "principal_point": [float(ppx), float(ppy)]}
    if intrinsic_type == "pinhole_radial_k3":
        data = {"value0": data, "disto_k3": [float(x) for x in (disto_k3 or (0, 0, 0))]}
    intr = [{"key": 0, "value": {"polymorphic_id": 2147483650, "polymorphic_name": intrinsic_type,
                                 "ptr_wrapper": {"id": 2147483660, "data": data}}}]
    ext = []
    for vid in sorted(poses or {}):
        R, C = poses[vid]
        ext.append({"key": int(vid), "value": {"rotation": np.asarray(R, float).tolist(),
                                               "center": np.asarray(C, float).tolist()}})
    st = []
    for lid, X, obs in (structure or []):
        st.append({"key": int(lid), "value": {"X": [float(x) for x in X], "observations": [
            {"key": int(v), "value": {"id_feat": int(i), "x": [float(x[0]), float(x[1])]}} for v, i, x in obs]}})
    return {"sfm_data_version": "0.2", "root_path": root_path, "views": views, "intrinsics": intr,
            "extrinsics": ext, "structure": st, "control_points": []}


def write_sfm_data(path, sfm_data):
    with open(path, "w") as f:
        json.dump(sfm_data, f)


def read_sfm_data(path):
    with open(path) as f:
        return json.load(f)


# ---- <outDir>/<base>.json (localization.cpp:84-153) ----
def _eigen_format(mat, row_prefix, row_suffix):
    """Eigen IOFormat(6, 0, ",", ",\\n", row_prefix, row_suffix, "[", "]"): 6 significant digits, aligned columns."""
    m = np.atleast_2d(np.asarray(mat, dtype=np.float64))
    cells = [["%.6g" % v for v in row] for row in m]
    width = max(len(c) for row in cells for c in row)
    rows = [row_prefix + ",".join(c.rjust(width) for c in row) + row_suffix for row in cells]
    return "[" + ",\n".join(rows) + "]"


def format_result_json(query_path, sfm_data_path, matches_dir, K=None, R=None, center=None, pairs=None):
    s = "{\n"
    s += f'\t"filename": "{query_path}",\n'
    s += f'\t"sfm_data": "{sfm_data_path}",\n'
    if K is None:  # failure: first three keys only (localization.cpp:84-109)
        s += f'\t"matches_dir": "{matches_dir}"\n'
        s += "}\n"
        return s
    s += f'\t"matches_dir": "{matches_dir}",\n'
    s += '\t"K": ' + _eigen_format(np.asarray(K).reshape(3, 3), "[", "]") + ",\n"
    s += '\t"R": ' + _eigen_format(np.asarray(R).reshape(3, 3), "[", "]") + ",\n"
    s += '\t"t": ' + _eigen_format(np.asarray(center).reshape(3, 1), "", "") + ",\n"
    s += '\t"pair": [' + ",".join(f"[{int(a)},{int(b)}]" for a, b in ([] if pairs is None else pairs)) + "]\n"
    s += "}\n"
    return s


def write_result_json(out_dir, query_path, sfm_data_path, matches_dir, **kw):
    base = os.path.splitext(os.path.basename(query_path))[0]
    path = os.path.join(out_dir, base + ".json")
    with open(path, "w") as f:
        f.write(format_result_json(query_path, sfm_data_path, matches_dir, **kw))
    return path


# ---------------------------------------------------------------------------------------------------------
# PairWiseMatches text files (matches.putative.txt / matches.f.txt): "I J\nN\ni j\n..." per pair, pairs in
# std::map order -- hulo::exportMatch (FileUtils.cpp:123-148) and OpenMVG's matching::Save(".txt") share the layout
# ---------------------------------------------------------------------------------------------------------
def write_matches_txt(path, matches):
    """matches: {(I, J): (i[], j[])} as Map.match_pairs / Map.track return (view ids as keys)."""
    with open(path, "w") as fh:
        for (a, b) in sorted(matches):
            mi, mj = matches[(a, b)]
            fh.write(f"{int(a)} {int(b)}\n{len(mi)}\n")
            for x, y in zip(mi, mj):
                fh.write(f"{int(x)} {int(y)}\n")


def read_matches_txt(path):
    out = {}
    with open(path) as fh:
        tok = fh.read().split()
    p = 0
    while p < len(tok):
        a, b, n = int(tok[p]), int(tok[p + 1]), int(tok[p + 2])
        p += 3
        arr = np.array(tok[p:p + 2 * n], dtype=np.int64).reshape(n, 2) if n else np.zeros((0, 2), np.int64)
        p += 2 * n
        out[(a, b)] = (arr[:, 0].astype(np.uint32), arr[:, 1].astype(np.uint32))
    return out
