#!/usr/bin/env python3
"""Randomised parity campaign: whole path through the C ABI against the oracle on many random scenes (sizes, ragged
views, view selections, ratios, RANSAC budgets, outlier rates, radial intrinsics).  Every stage bit-exact.
usage: fuzz_parity.py [n_scenes] [first_seed]"""
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


def one(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_views = int(rng.integers(8, 50))
    dpv = int(rng.integers(120, 700))
    vpp = int(rng.integers(4, 12))
    m = synth.make_map(seed, n_views=n_views, desc_per_view=dpv, views_per_place=vpp,
                       landmarks_per_place=int(rng.integers(150, 400)), obs_per_view=int(rng.integers(60, min(dpv, 260))),
                       ragged=bool(rng.integers(0, 2)))
    if rng.uniform() < 0.25:
        m.intrinsic = tuple(m.intrinsic[:3]) + (float(rng.normal(0, 0.1)), float(rng.normal(0, 0.03)), 0.0)
    ratio = float(rng.choice([0.5, 0.6, 0.8]))
    rounds = int(rng.choice([5, 25, 60, 200]))
    nq = int(rng.choice([300, 700, 900, 1500, 2300]))
    guided = bool(rng.uniform() < 0.3)                     # -gm: guided matching in the geometric stage
    p = S.default_params(dist_ratio=ratio, ransac_round=rounds, guided_matching=int(guided))
    n_cmp = 0
    bow = np.sqrt(rng.random((n_views, 64))).astype(np.float32)       # .bow vectors for the shortlist chain
    with S.Map(m.view_id, m.view_off, m.desc, params=p, view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark,
               landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic, bow=bow) as dm:
        for k in range(3):
            q = synth.make_query(m, seed * 10 + k, n_feat=nq, n_copies=int(rng.integers(0, min(nq, 400))),
                                 outlier_frac=float(rng.uniform(0.0, 0.7)))
            sel = None
            if rng.uniform() < 0.3:
                sel = np.sort(rng.choice(n_views, int(rng.integers(1, n_views)), replace=False)).astype(np.uint32)
            exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=sel, ratio=ratio, ransac_round=rounds,
                                 guided=guided)
            dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            dm.match_putative(dq, sel)
            cnt, mi, mj, md = dm.putative_read()
            assert np.array_equal(cnt, exp["put_count"]) and np.array_equal(mi, exp["put_i"]) and np.array_equal(mj, exp["put_j"])
            dm.geometric_filter(dq)
            if guided:
                gc, gi, gj = dm.geometric_read_pairs()
                assert np.array_equal(gc, exp["geo_count"]), "guided matching: counts"
                for v in np.nonzero(gc)[0]:
                    o0, n = int(m.view_off[v]), int(gc[v])
                    assert np.array_equal(gi[o0:o0 + n], exp["geo_idx"][o0:o0 + n]), "guided matching: map features"
                    assert np.array_equal(gj[o0:o0 + n], exp["geo_j"][o0:o0 + n]), "guided matching: query features"
            else:
                gc, gi = dm.geometric_read()
                assert np.array_equal(gc, exp["geo_count"]) and np.array_equal(gi, exp["geo_idx"]), "F-matrix filter"
            dm.match_set(dq)
            qf, lm, p2, p3 = dm.match_set_read()
            assert np.array_equal(qf, exp["ms_qfeat"]) and np.array_equal(bits(p2), bits(exp["pt2d"])), "match set"
            dm.resection(dq)
            pose, pq, pl, ii = dm.pose_read()
            assert bool(pose.ok) == exp["ok"], "ok flag"
            if exp["ok"]:
                assert np.array_equal(ii, exp["inlier_idx"]) and np.array_equal(bits(np.array(pose.P)), bits(exp["P"].ravel()))
            # the one-call path on a context must agree with the staged one
            c = dm.context()
            c.begin(dq, sel)
            pose2, pq2, _ = c.end()
            assert bool(pose2.ok) == exp["ok"] and (not exp["ok"] or np.array_equal(pq2, exp["pair_qfeat"]))
            # shortlist + path in one call (the shortlist never leaves the device) against the oracle's shortlist
            # followed by the oracle's path on it
            if rng.uniform() < 0.5:
                n_cand = n_views if sel is None else len(sel)
                knn = int(rng.integers(1, n_cand + 3))
                qb = (bow[rng.integers(0, n_views)] + rng.normal(0, 0.05, 64)).astype(np.float32)
                osel = oracle_c.bow_select(bow, qb, knn, sel) if knn < n_cand else sel
                exp_b = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=osel, ratio=ratio,
                                       ransac_round=rounds, guided=guided)
                c.begin_bow(dq, qb, knn, sel)
                pose3, pq3, _ = c.end()
                assert bool(pose3.ok) == exp_b["ok"], "shortlist chain: ok flag"
                if exp_b["ok"]:
                    assert np.array_equal(pq3, exp_b["pair_qfeat"]), "shortlist chain: pairs"
                    assert np.array_equal(bits(np.array(pose3.P)), bits(exp_b["P"].ravel())), "shortlist chain: P"
            c.close()
            dq.close()
            n_cmp += 1
    return n_cmp


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    oracle_c.build()
    t0 = time.time()
    total = 0
    for s in range(first, first + n):
        try:
            total += one(s)
        except AssertionError as e:
            print(f"seed {s}: PARITY FAILURE: {e}", flush=True)
            raise
        if (s - first) % 10 == 9:
            print(f"{s - first + 1} scenes, {total} queries compared, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {n} scenes, {total} queries, every stage bit-exact ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
