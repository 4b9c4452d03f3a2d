"""What a round-by-round schedule of K5 (P3P AC-RANSAC) costs, from the ORACLE's sequential run.

AC-RANSAC is sequential: iteration i samples from the index set the iterations before it left.  The device evaluates a
round of B hypotheses side by side and replays the rule; everything behind the round's first index-changing iteration is
thrown away.  Which iterations improve the model / change the index set does not depend on the schedule, so the oracle's
trace (orc_acransac_trace) is enough to price any schedule: rounds (a lone query's latency) and hypotheses evaluated (the
chip's time under load).  CPU only; uses the oracle, i.e. measurement infrastructure, not the product.

  python tools/k5_policy_sim.py [--queries 24] [--views 400]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def traces(n_queries, n_views, nq, knn):
    import synthdata as synth
    from oracle import oracle_c, pipeline as opipe
    import bench
    m = synth.make_map(2, n_views=n_views, desc_per_view=2000)
    queries = [synth.make_query(m, 1000 + i, n_feat=nq) for i in range(n_queries)]
    bow, qbow = bench.synth_bow(m, queries)
    L = oracle_c.lib()
    L.orc_acransac_trace.argtypes = [C.c_void_p, C.c_int]
    out = []
    for q, qb in zip(queries, qbow):
        d = ((bow - qb[None, :]) ** 2).sum(1)
        sel = np.sort(np.argsort(d, kind="stable")[:knn]).astype(np.uint32)
        r = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=sel, ransac_round=25, threads=8)
        if "p3p" not in r:
            continue
        buf = np.zeros(4096, np.int32)
        L.orc_acransac_trace(buf.ctypes.data, len(buf))
        f, ppx, ppy = m.intrinsic[:3]
        rr = oracle_c.p3p_localize(r["pt2d"], r["pt3d"], f, ppx, ppy, 4096, 0x5f3759df12345678, stream=0)
        n = L.orc_acransac_trace_count()
        L.orc_acransac_trace(None, 0)
        ev = [(int(x) >> 2, bool(x & 2), bool(x & 1)) for x in buf[:n]]
        out.append({"n": len(r["pt2d"]), "iters": rr["iters"], "inliers": rr["n"], "events": ev})
    return out


def simulate(tr, policy):
    """-> (rounds, hypotheses evaluated).  policy(round_index, identity, t_since_switch, n) -> batch size."""
    total = tr["iters"]
    changes = [it for it, changed, _ in tr["events"] if changed]     # iterations that end a round early
    it0, rounds, evaluated = 0, 0, 0
    switch_it = None
    while it0 < total:
        identity = switch_it is None
        t = 0 if identity else it0 - switch_it
        b = max(1, policy(rounds, identity, t, tr["n"]))
        # the budget (n_iter) is known to the device: a round never runs past it.  While sampling is uniform the budget is
        # the main phase's (3 687); afterwards it is `total`
        horizon = total if not identity else 4096 - 409
        b = min(b, horizon - it0)
        nxt = next((c for c in changes if it0 <= c < it0 + b), None)
        evaluated += b
        rounds += 1
        if nxt is None:
            it0 += b
        else:
            it0 = nxt + 1
            if switch_it is None:
                switch_it = it0
    return rounds, evaluated


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=24)
    ap.add_argument("--views", type=int, default=400)
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--knn", type=int, default=100)
    ap.add_argument("--dump", default="")
    a = ap.parse_args()
    trs = traces(a.queries, a.views, a.nq, a.knn)
    if a.dump:
        json.dump(trs, open(a.dump, "w"))
    print(f"{len(trs)} queries; correspondences {np.mean([t['n'] for t in trs]):.0f}, iterations "
          f"{np.mean([t['iters'] for t in trs]):.0f}, improvements {np.mean([sum(1 for e in t['events'] if e[2]) for t in trs]):.1f}, "
          f"index changes {np.mean([sum(1 for e in t['events'] if e[1]) for t in trs]):.1f}")
    gaps = []
    for t in trs:
        ch = [it for it, c, _ in t["events"] if c]
        gaps += [b - a_ for a_, b in zip(ch, ch[1:])]
    print("gaps between index changes: " + " ".join(f"p{p}={np.percentile(gaps, p):.0f}" for p in (10, 25, 50, 75, 90)))

    def adaptive(first, floor, quarters, cap=256):
        def pol(r, identity, t, n):
            if identity:
                return first if r == 0 else cap
            return int(min(cap, max(floor, quarters * t // 4)))
        return pol

    def alone(first, later):
        return lambda r, identity, t, n: first if r == 0 else later

    pols = {"alone: 64 then 256 (round 3)": alone(64, 256),
            "alone: 16 then 256": alone(16, 256),
            "alone: 64 then 128": alone(64, 128),
            "shared: first 64, floor 64, 3t (round 3)": adaptive(64, 64, 12),
            "shared: first 16, floor 64, 3t": adaptive(16, 64, 12),
            "shared: first 16, floor 32, 3t": adaptive(16, 32, 12),
            "shared: first 16, floor 16, 3t": adaptive(16, 16, 12),
            "shared: first 8, floor 16, 3t": adaptive(8, 16, 12),
            "shared: first 8, floor 8, 3t": adaptive(8, 8, 12),
            "shared: first 16, floor 16, 2t": adaptive(16, 16, 8),
            "shared: first 16, floor 16, 4t": adaptive(16, 16, 16),
            "shared: first 16, floor 16, 6t": adaptive(16, 16, 24),
            "shared: first 4, floor 4, 3t": adaptive(4, 4, 12),
            "sequential (1 per round)": lambda r, identity, t, n: 1}
    for name, pol in pols.items():
        res = [simulate(t, pol) for t in trs]
        print(f"{name:45s} rounds {np.mean([r for r, _ in res]):5.1f} (max {max(r for r, _ in res)}), "
              f"hypotheses evaluated {np.mean([e for _, e in res]):6.0f}")


if __name__ == "__main__":
    main()
