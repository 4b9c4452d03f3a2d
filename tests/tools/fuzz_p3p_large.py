#!/usr/bin/env python3
"""Randomised parity campaign for K5 on LARGE match sets (round 3: the NFA filter from 257 correspondences on, one
workgroup per model from 513 on, the LDS forms up to 4 096 and the global-memory form beyond): few views with thousands
of rows, queries that nearly duplicate them (600 ... 5 000 correspondences, 5 ... 50 % outliers), several queries per
context so that both launch shapes occur (the first large query of a map runs the plain shape, the following ones the
wide one).  Pose bits and inlier pairs against the oracle.  usage: fuzz_p3p_large.py [n_scenes] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402
from oracle import oracle_c, pipeline as opipe  # noqa: E402


def bits(a):
    a = np.ascontiguousarray(a, np.float64)
    u = a.view(np.uint64).copy()
    u[np.isnan(a)] = 0x7FF8000000000000
    return u


def one(seed, stats):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_views = int(rng.integers(3, 7))
    dpv = int(rng.integers(1500, 5200))
    m = synth.make_map(seed, n_views=n_views, desc_per_view=dpv, views_per_place=n_views,
                       landmarks_per_place=int(dpv * 1.2), obs_per_view=int(dpv * rng.uniform(0.6, 0.97)), map_flips=8)
    p3p_it = int(rng.integers(150, 500))
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=p3p_it),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    ctx = dm.context()
    try:
        for k in range(3):
            nf = int(rng.integers(700, min(5300, dpv + 300)))
            nc = int(rng.integers(300, max(301, min(nf - 50, int(dpv * 0.9)))))
            q = synth.make_query(m, seed * 10 + k, n_feat=nf, n_copies=nc, outlier_frac=float(rng.uniform(0.05, 0.5)),
                                 query_flips=int(rng.integers(5, 30)))
            exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), p3p_max_iteration=p3p_it, threads=8)
            dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            ctx.begin(dq)
            pose, pq, pl = ctx.end()
            dq.close()
            assert bool(pose.ok) == exp["ok"], "localised or not"
            n23 = len(exp["ms_qfeat"])
            assert pose.n_matches_2d3d == n23, "2D-3D correspondences"
            stats.append(n23)
            if exp["ok"]:
                assert np.array_equal(pq, exp["pair_qfeat"]) and np.array_equal(pl, exp["pair_landmark"]), "inlier pairs"
                assert np.array_equal(bits(np.array(pose.P)), bits(exp["P"].ravel())), "P"
    finally:
        ctx.close()
        dm.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 93000
    oracle_c.build()
    t0 = time.time()
    stats = []
    for i in range(n):
        try:
            one(s0 + i, stats)
        except AssertionError as e:
            print(f"MISMATCH seed {s0 + i}: {e}")
            return 1
        if (i + 1) % 10 == 0:
            a = np.array(stats)
            print(f"{i + 1} scenes, {len(a)} queries, correspondences {a.min()} .. {a.max()} (median {int(np.median(a))}), "
                  f"{int((a > 512).sum())} above 512, {int((a > 4096).sum())} above 4096, {time.time() - t0:.0f} s", flush=True)
    a = np.array(stats)
    print(f"OK: {n} scenes, {len(a)} queries, every pose and inlier set bit-exact; correspondences {a.min()} .. {a.max()}, "
          f"{int((a > 256).sum())} above 256 (filter), {int((a > 512).sum())} above 512 (one workgroup per model), "
          f"{int((a > 4096).sum())} above 4096 ({time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
