#!/usr/bin/env python3
"""Randomised campaign for gang sessions: random scenes, 2..7 queries of different sizes taken through the whole path
(sfmloc_localize_bow_begin, or sfmloc_localize_begin on a random view selection) TOGETHER in one session -- members with
and without a stream of their own, random parameters -- against the same calls one context at a time, bit for bit
(those are pinned to the oracle by fuzz_parity.py).
usage: fuzz_gang.py [n_scenes] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sfmlocalization_amd as S  # noqa: E402
from sfmlocalization_amd import capi  # noqa: E402
import synthdata as synth  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def one(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_views = int(rng.integers(12, 70))
    dpv = int(rng.integers(150, 600))
    m = synth.make_map(seed, n_views=n_views, desc_per_view=dpv, views_per_place=int(rng.integers(4, 12)),
                       landmarks_per_place=int(rng.integers(150, 400)), obs_per_view=int(rng.integers(60, min(dpv, 260))),
                       ragged=bool(rng.integers(0, 2)))
    dim = int(rng.choice([16, 64, 500]))
    place_bow = rng.uniform(0, 1, (len(m.place_center), dim)).astype(np.float32)
    bow = (place_bow[m.view_place] + rng.normal(0, 0.05, (m.n_views, dim))).astype(np.float32)
    p = S.default_params(dist_ratio=float(rng.choice([0.5, 0.6, 0.8])), ransac_round=int(rng.choice([5, 25, 60])),
                         guided_matching=int(rng.random() < 0.25))
    n_q = int(rng.integers(2, 8))
    knn = int(rng.integers(3, max(4, n_views - 1)))
    n = 0
    with S.Map(m.view_id, m.view_off, m.desc, params=p, view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark,
               landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic, bow=bow) as dm:
        qs, dqs, sels = [], [], []
        for k in range(n_q):
            nq = int(rng.choice([120, 300, 800, 1600]))
            q = synth.make_query(m, seed * 10 + k, n_feat=nq, n_copies=int(rng.integers(0, min(nq, 400))),
                                 outlier_frac=float(rng.uniform(0.0, 0.6)))
            dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            dq.set_bow((place_bow[q.place] + rng.normal(0, 0.05, dim)).astype(np.float32))
            qs.append(q)
            dqs.append(dq)
            # a third of the queries skip the shortlist and bring a view selection of their own
            sels.append(np.sort(rng.choice(n_views, int(rng.integers(1, n_views + 1)), replace=False)).astype(np.uint32)
                        if rng.random() < 0.33 else None)
        plain = dm.context()
        ref = []
        for dq, sel in zip(dqs, sels):
            if sel is None:
                plain.begin_bow(dq, None, knn)
            else:
                plain.begin(dq, sel)
            ref.append(plain.end())
        plain.close()
        lead = dm.context()
        ctxs = [lead] + [dm.context(share=lead if rng.random() < 0.7 else None) for _ in range(n_q - 1)]
        with capi.gang(ctxs):
            for c, dq, sel in zip(ctxs, dqs, sels):
                if sel is None:
                    c.begin_bow(dq, None, knn)
                else:
                    c.begin(dq, sel)
        for c, r in zip(ctxs, ref):
            pose, pq, pl = c.end()
            assert bool(pose.ok) == bool(r[0].ok) and pose.n_inliers == r[0].n_inliers, "ok / inliers"
            assert pose.n_putative_views == r[0].n_putative_views and pose.n_geometric_views == r[0].n_geometric_views, "views"
            if pose.ok:     # (a failed localisation leaves no pose to compare)
                assert np.array_equal(pq, r[1]) and np.array_equal(pl, r[2]), "pairs"
                assert np.array_equal(bits(np.array(pose.P)), bits(np.array(r[0].P))), "P"
                assert np.array_equal(bits(np.array(pose.center)), bits(np.array(r[0].center))), "centre"
            n += 1
        for c in reversed(ctxs):
            c.close()
        for dq in dqs:
            dq.close()
    return n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
    t0 = time.time()
    total = 0
    for s in range(first, first + n):
        try:
            total += one(s)
        except AssertionError as e:
            print(f"seed {s}: PARITY FAILURE: {e}", flush=True)
            raise
        if (s - first) % 20 == 19:
            print(f"{s - first + 1} scenes, {total} queries compared, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {n} scenes, {total} queries, a gang session = the same calls one at a time, bit for bit ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
