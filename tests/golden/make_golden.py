"""Mints the golden fixtures of tests/golden/ from the NumPy twin of the oracle (oracle/oracle_np.py).

The reference holds no golden vectors for this path -- these are the build's own
known-answer vectors: planted near-duplicates at known Hamming distances, exact ties, d1 = 0.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_np as onp  # noqa: E402
import synthdata as synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    q, bank = synth.planted_bank(seed=1)
    j0, d0, j1, d1 = onp.hamming_2nn(q, bank)
    acc = onp.ratio_accept(d0, d1, 0.6)
    view_off = np.array([0, 100, 100, 163, 300, 512], dtype=np.uint32)  # 5 views, one empty, ragged
    cnt, mi, mj, md = onp.match_to_query(q, bank, view_off, None, 0.6)
    np.savez_compressed(os.path.join(HERE, "hamming_planted.npz"), query=q, bank=bank, j0=j0, d0=d0, j1=j1, d1=d1,
                        accept06=acc, view_off=view_off, view_count=cnt, match_i=mi, match_j=mj, match_d=md)
    print("hamming_planted.npz: rows", bank.shape[0], "query", q.shape[0], "accepted", int(acc.sum()),
          "d1==0 rows", int((d1 == 0).sum()), "ties", int((d0 == d1).sum()))


def geometry():
    """(ii) planted pinhole scenes with the analytically known answer plus the oracle's exact output (regression pin
    of the C restatement: sampling, solvers, NFA bookkeeping, budget rules)."""
    from oracle import oracle_c
    oracle_c.build()
    rng = np.random.Generator(np.random.PCG64(77))
    f, ppx, ppy, w, h = 800.0, 320.0, 240.0, 640, 480
    # --- resection: 2D-3D correspondences, 35 % outliers
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    C = np.array([0.5, -0.2, -9.0])
    X = rng.uniform(-3, 3, (260, 3))
    pc = (X - C) @ R.T
    px = np.stack([f * pc[:, 0] / pc[:, 2] + ppx, f * pc[:, 1] / pc[:, 2] + ppy], 1) + rng.normal(0, 0.7, (260, 2))
    is_out = rng.uniform(size=260) < 0.35
    px[is_out] = np.stack([rng.uniform(0, w, is_out.sum()), rng.uniform(0, h, is_out.sum())], 1)
    px = px.astype(np.float32).astype(np.float64)       # pt2D comes from float32 keypoints
    r = oracle_c.p3p_localize(px, X, f, ppx, ppy, 4096, 0x5f3759df12345678, stream=0)
    K, Rr, t, c = oracle_c.krt_from_p(r["P"])
    # --- fundamental: two views of the same points, 40 % outliers
    ang2 = -0.25
    R2 = np.array([[np.cos(ang2), 0, np.sin(ang2)], [0, 1, 0], [-np.sin(ang2), 0, np.cos(ang2)]])
    C2 = np.array([2.0, 0.3, -9.5])
    pc2 = (X - C2) @ R2.T
    px2 = np.stack([f * pc2[:, 0] / pc2[:, 2] + ppx, f * pc2[:, 1] / pc2[:, 2] + ppy], 1) + rng.normal(0, 0.5, (260, 2))
    x1 = synth.round6((px[:180] if False else np.stack([f * pc[:180, 0] / pc[:180, 2] + ppx, f * pc[:180, 1] / pc[:180, 2] + ppy], 1)
                       + rng.normal(0, 0.5, (180, 2))).astype(np.float32)).astype(np.float64)
    x2 = synth.round6(px2[:180].astype(np.float32)).astype(np.float64)
    f_out = rng.uniform(size=180) < 0.4
    x2[f_out] = np.stack([rng.uniform(0, w, f_out.sum()), rng.uniform(0, h, f_out.sum())], 1)
    x2 = synth.round6(x2.astype(np.float32)).astype(np.float64)
    fr = oracle_c.fmatrix_filter(x1, (w, h), x2, (w, h), 4.0, 200, 0x5f3759df12345678, stream=7)
    np.savez_compressed(os.path.join(HERE, "geometry_scenes.npz"), intrinsic=np.array([f, ppx, ppy]), wh=np.array([w, h]),
                        pt2d=px, pt3d=X, R_true=R, C_true=C, is_outlier=is_out,
                        p3p_n=r["n"], p3p_inliers=r["inliers"], p3p_P=r["P"], p3p_nfa=r["nfa"], p3p_errmax=r["errmax"],
                        p3p_iters=r["iters"], p3p_center=c, p3p_R=Rr,
                        f_x1=x1, f_x2=x2, f_is_outlier=f_out, f_n=fr["n"], f_inliers=fr["inliers"], f_F=fr["F"],
                        f_nfa=fr["nfa"], f_iters=fr["iters"])
    print("geometry_scenes.npz: P3P inliers", r["n"], "of", int((~is_out).sum()), "true; centre error",
          float(np.abs(c - C).max()), "| F inliers", fr["n"], "of", int((~f_out).sum()), "true")


def files():
    """(iii) byte-level samples of the on-disk contract, built by hand from the layouts in FileUtils.cpp:43-103,
    AKAZEOption.cpp:31-41, AKAZEOpenCV.cpp:80-81, reconstructGraph.py:135-148 -- NOT through sfmlocalization_amd.fileio."""
    import json
    import struct
    d = os.path.join(HERE, "files")
    os.makedirs(d, exist_ok=True)
    rows = np.arange(3 * 64, dtype=np.uint8).reshape(3, 64).copy()
    rows[:, 61:] = 0
    rows[:, 60] &= 0x3F
    with open(os.path.join(d, "img000.desc"), "wb") as fh:          # [u64 N][N x 64 B], bytes 61..63 zero
        fh.write(struct.pack("<Q", 3) + rows.tobytes())
    with open(os.path.join(d, "img000.feat"), "w") as fh:           # x y size angle, `ostream << float`
        fh.write("12.5 7.25 4.8 90\n123.457 0.000123457 9.6 359.5\n639 479 19.2 0\n")
    vec = np.array([0.0, 0.25, 0.5, 1.0], np.float64)
    with open(os.path.join(d, "img000.bow"), "wb") as fh:           # [i32 rows][i32 cols][i32 type=CV_64F][data]
        fh.write(struct.pack("<iii", 4, 1, 6) + vec.tobytes())
    with open(os.path.join(d, "image_describer.txt"), "w") as fh:   # cv::FileStorage YAML
        fh.write("%YAML:1.0\ndesc_ch: 3\nthres: 1.0000000474974513e-03\nnOct: 4\nnOctLay: 4\n")
    view = lambda k, first: {"key": k, "value": dict(({"polymorphic_name": "view"} if first else {}), **{
        "polymorphic_id": 2147483649 if first else 1073741824,
        "ptr_wrapper": {"id": 2147483649 + k, "data": {"local_path": "/", "filename": f"img{k:03d}.jpg", "width": 640,
                                                       "height": 480, "id_view": k, "id_intrinsic": 0, "id_pose": k}}})}
    sd = {"sfm_data_version": "0.2", "root_path": "/data/images", "views": [view(0, True), view(1, False)],
          "intrinsics": [{"key": 0, "value": {"polymorphic_id": 2147483650, "polymorphic_name": "pinhole",
                                              "ptr_wrapper": {"id": 2147483660, "data": {
                                                  "width": 640, "height": 480, "focal_length": 800.0,
                                                  "principal_point": [320.0, 240.0]}}}}],
          "extrinsics": [{"key": 0, "value": {"rotation": [[1, 0, 0], [0, 1, 0], [0, 0, 1]], "center": [0.0, 0.0, 0.0]}}],
          "structure": [{"key": 5, "value": {"X": [1.0, 2.0, 3.0], "observations": [
              {"key": 0, "value": {"id_feat": 1, "x": [123.457, 0.000123457]}}]}}],
          "control_points": []}
    with open(os.path.join(d, "sfm_data.json"), "w") as fh:
        json.dump(sd, fh, indent=1)
    print("files/: img000.desc img000.feat img000.bow image_describer.txt sfm_data.json")


if __name__ == "__main__":
    main()
    geometry()
    files()
