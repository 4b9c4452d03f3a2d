#!/usr/bin/env python3
"""Randomised parity campaign for the map-building twins: sfmloc_track / sfmloc_match_pairs /
sfmloc_geometric_pairs against the literal restatements of hulo::trackAKAZE / matchAKAZE / geometricMatch in
oracle/pipeline.py, on random image sets (ragged, duplicated descriptors that trigger the one-to-one filter, random
ratios, track lengths, RANSAC budgets).  usage: fuzz_mapside.py [n_sets] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402
from oracle import oracle_c, pipeline as opipe  # noqa: E402


def same(got, exp, what):
    assert list(got.keys()) == list(exp.keys()), f"{what}: pair set"
    for k in exp:
        assert np.array_equal(got[k][0], exp[k][0]) and np.array_equal(got[k][1], exp[k][1]), f"{what}: pair {k}"


def one(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_views = int(rng.integers(3, 11))
    dpv = int(rng.integers(60, 500))
    m = synth.make_map(seed, n_views=n_views, desc_per_view=dpv, views_per_place=n_views,
                       landmarks_per_place=int(rng.integers(60, 300)), obs_per_view=int(rng.integers(30, max(31, dpv - 10))),
                       map_flips=int(rng.integers(2, 14)), ragged=bool(rng.integers(0, 2)))
    desc = m.desc.copy()
    off = m.view_off.astype(np.int64)
    for v in range(n_views - 1):                      # duplicates: rows the one-to-one filter has to drop
        n = int(off[v + 1] - off[v])
        for _ in range(int(rng.integers(0, 4))):
            if n >= 4:
                a, b = rng.choice(n, 2, replace=False)
                desc[off[v] + b] = desc[off[v] + a]
    descs = [desc[off[v]:off[v + 1]] for v in range(n_views)]
    ratio = float(rng.choice([0.5, 0.6, 0.8]))
    rounds = int(rng.choice([10, 25, 120]))
    kp = synth.round6(m.kpt_xy)
    p = S.default_params(dist_ratio=ratio, ransac_round=rounds, geom_precision=4.0)
    with S.Map(m.view_id, m.view_off, desc, params=p, view_wh=m.view_wh, kpt_xy=kp) as dm:
        dist = int(rng.integers(1, n_views + 2))
        got = dm.track(dist)
        same(got, opipe.track_akaze(descs, dist, ratio), "track")
        pairs = [(int(rng.integers(0, n_views)), int(rng.integers(0, n_views))) for _ in range(int(rng.integers(1, 12)))]
        same(dm.match_pairs(pairs), opipe.match_akaze(descs, sorted(set(pairs)), ratio), "match_pairs")
        put = {k: v for k, v in got.items() if len(v[0]) >= int(rng.integers(8, 30))}
        if put:
            g = dm.geometric_pairs(put)
            e = opipe.geometric_match([kp[off[v]:off[v + 1]] for v in range(n_views)], m.view_wh, m.view_id, put,
                                      ransac_round=rounds)
            same(g, e, "geometric_pairs")
    return len(got)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
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
            print(f"{s - first + 1} image sets, {total} tracked pairs compared, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {n} image sets, {total} tracked pairs, every list bit-exact ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
