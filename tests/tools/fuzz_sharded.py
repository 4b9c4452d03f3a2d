#!/usr/bin/env python3
"""Randomised parity campaign for the multi-GPU path on one GPU: the map's views split into 2..5 shards (separate
sfmloc_map objects), sfmloc_shard_begin / _export per shard, the parts concatenated as the all-gather would,
sfmloc_merge_begin on one of them -- against the oracle's UNSHARDED result, bit for bit.
usage: fuzz_sharded.py [n_scenes] [first_seed]
SFMLOC_FUZZ_GANG=1: a scene's two queries go through every shard, and through the merge, TOGETHER in gang sessions
(sfmloc_gang_begin / _end, the second context without a stream of its own) -- same expectation."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sfmlocalization_amd as S  # noqa: E402
from sfmlocalization_amd import dist as D  # noqa: E402
import synthdata as synth
from oracle import oracle_c, pipeline as opipe  # noqa: E402


def bits(a):
    a = np.ascontiguousarray(a, np.float64)
    u = a.view(np.uint64).copy()
    u[np.isnan(a)] = 0x7FF8000000000000
    return u


def one(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_views = int(rng.integers(6, 40))
    dpv = int(rng.integers(150, 600))
    m = synth.make_map(seed, n_views=n_views, desc_per_view=dpv, views_per_place=int(rng.integers(4, 12)),
                       landmarks_per_place=int(rng.integers(150, 400)), obs_per_view=int(rng.integers(60, min(dpv, 260))),
                       ragged=bool(rng.integers(0, 2)))
    ratio = float(rng.choice([0.5, 0.6, 0.8]))
    rounds = int(rng.choice([5, 25, 60]))
    nq = int(rng.choice([300, 800, 1600]))
    world = int(rng.integers(2, 6))
    cap = 4096
    p = S.default_params(dist_ratio=ratio, ransac_round=rounds)
    ranges = D.shard_views(m.view_off, world)
    shards = []
    for v0, v1 in ranges:
        if v1 == v0:              # more ranks than this split has views for: that rank's part stays empty
            shards.append(None)
            continue
        r0, r1 = int(m.view_off[v0]), int(m.view_off[v1])
        shards.append(S.Map(m.view_id[v0:v1], m.view_off[v0:v1 + 1] - m.view_off[v0], m.desc[r0:r1], params=p,
                            view_wh=m.view_wh[v0:v1], kpt_xy=m.kpt_xy[r0:r1], row_landmark=m.row_landmark[r0:r1],
                            landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic))
    pb = D.part_bytes(cap)
    n = 0
    if os.environ.get("SFMLOC_FUZZ_GANG") == "1":
        try:
            return one_ganged(m, rng, seed, shards, world, cap, pb, nq, ratio, rounds)
        finally:
            for sm in shards:
                if sm is not None:
                    sm.close()
    try:
        for k in range(2):
            q = synth.make_query(m, seed * 10 + k, n_feat=nq, n_copies=int(rng.integers(0, min(nq, 400))),
                                 outlier_frac=float(rng.uniform(0.0, 0.6)))
            exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), ratio=ratio, ransac_round=rounds)
            parts = torch.zeros((world, pb), dtype=torch.uint8, device="cuda")
            qs = []
            for s, sm in enumerate(shards):
                if sm is None:
                    qs.append(None)
                    continue
                sq = sm.query(q.desc, q.kpt_xy, q.width, q.height)
                qs.append(sq)
                c = sm.context()
                c.shard_begin(sq)
                c.shard_export(parts.data_ptr() + s * pb, cap)
                c.sync()
                c.close()
            owner = int(rng.choice([i for i, sm in enumerate(shards) if sm is not None]))
            c = shards[owner].context()
            c.merge_begin(qs[owner], parts.data_ptr(), world, cap)
            pose, pq, pl = c.end()
            c.close()
            assert bool(pose.ok) == exp["ok"], "ok flag"
            if exp["ok"]:
                assert np.array_equal(pq, exp["pair_qfeat"]) and np.array_equal(pl, exp["pair_landmark"]), "pairs"
                assert np.array_equal(bits(np.array(pose.P)), bits(exp["P"].ravel())), "P"
            for sq in qs:
                if sq is not None:
                    sq.close()
            n += 1
    finally:
        for sm in shards:
            if sm is not None:
                sm.close()
    return n


def one_ganged(m, rng, seed, shards, world, cap, pb, nq, ratio, rounds):
    from sfmlocalization_amd import capi
    qs, exps = [], []
    for k in range(2):
        q = synth.make_query(m, seed * 10 + k, n_feat=nq, n_copies=int(rng.integers(0, min(nq, 400))),
                             outlier_frac=float(rng.uniform(0.0, 0.6)))
        qs.append(q)
        exps.append(opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), ratio=ratio, ransac_round=rounds))
    parts = torch.zeros((2, world, pb), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    sq = [[None] * world for _ in range(2)]
    for s, sm in enumerate(shards):
        if sm is None:
            continue
        lead = sm.context()
        cs = [lead, sm.context(share=lead)]
        for k in range(2):
            sq[k][s] = sm.query(qs[k].desc, qs[k].kpt_xy, qs[k].width, qs[k].height)
        with capi.gang(cs):
            for k in range(2):
                cs[k].shard_begin(sq[k][s])
                cs[k].shard_export(parts[k, s].data_ptr(), cap)
        lead.sync()
        cs[1].close()
        lead.close()
    owner = int(rng.choice([i for i, sm in enumerate(shards) if sm is not None]))
    lead = shards[owner].context()
    cs = [lead, shards[owner].context(share=lead)]
    with capi.gang(cs):
        for k in range(2):
            cs[k].merge_begin(sq[k][owner], parts[k].data_ptr(), world, cap)
    for k in range(2):
        pose, pq, pl = cs[k].end()
        exp = exps[k]
        assert bool(pose.ok) == exp["ok"], "ok flag"
        if exp["ok"]:
            assert np.array_equal(pq, exp["pair_qfeat"]) and np.array_equal(pl, exp["pair_landmark"]), "pairs"
            assert np.array_equal(bits(np.array(pose.P)), bits(exp["P"].ravel())), "P"
    cs[1].close()
    lead.close()
    for k in range(2):
        for x in sq[k]:
            if x is not None:
                x.close()
    return 2


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    oracle_c.build()
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
    print(f"OK: {n} scenes, {total} queries, sharded = the oracle's unsharded result bit for bit ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
