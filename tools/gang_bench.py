#!/usr/bin/env python3
"""The headline workload (BASELINE configs[2]: 10 k views, BoW shortlist of 100, 2 000 features per query) with the
queries taken through the path in GANG sessions: T host threads, each alternating between two gangs of G contexts (one
stream per gang), so that one gang's chain is queued while the other's results are awaited.
usage: python tools/gang_bench.py --threads 2 --gang 8 [--queries-per-thread 512]"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--gangs-per-thread", type=int, default=2)
    ap.add_argument("--gang", type=int, default=8)
    ap.add_argument("--sessions", type=int, default=40, help="timed sessions per gang")
    ap.add_argument("--views", type=int, default=10000)
    ap.add_argument("--bow-knn", type=int, default=100)
    a = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import bench
    import sfmlocalization_amd as S
    import synthdata as synth
    from sfmlocalization_amd import capi

    m = synth.make_map(2, n_views=a.views, desc_per_view=2000)
    queries = [synth.make_query(m, 1000 + i, n_feat=2000) for i in range(64)]
    bow, qbow = bench.synth_bow(m, queries)
    params = S.default_params(device=0, profile=0, ransac_round=25)
    dm = S.Map(m.view_id, m.view_off, m.desc, params=params, view_wh=m.view_wh, kpt_xy=m.kpt_xy,
               row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
               bow=bow)
    dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in queries]
    for dq, qb in zip(dqs, qbow):
        dq.set_bow(qb)
    gangs = []
    for _ in range(a.threads * a.gangs_per_thread):
        lead = dm.context()
        gangs.append([lead] + [dm.context(share=lead) for _ in range(a.gang - 1)])
    ok = [0] * a.threads
    lat = [[] for _ in range(a.threads)]

    def begin(g, first):
        with capi.gang(g):
            for k, c in enumerate(g):
                c.begin_bow(dqs[(first + k) % len(dqs)], None, a.bow_knn)
        return time.perf_counter()

    def end(g):
        return sum(int(c.end()[0].ok) for c in g)

    def worker(t, n_sessions):
        mine = gangs[t * a.gangs_per_thread:(t + 1) * a.gangs_per_thread]
        t_begin = [0.0] * len(mine)
        n = 0
        for j, g in enumerate(mine):
            t_begin[j] = begin(g, (t * 977 + n * a.gang) % 64)
            n += 1
        done = 0
        while done < n_sessions * len(mine):
            j = done % len(mine)
            ok[t] += end(mine[j])
            lat[t].append(time.perf_counter() - t_begin[j])
            done += 1
            if n < n_sessions * len(mine):
                t_begin[j] = begin(mine[j], (t * 977 + n * a.gang) % 64)
                n += 1

    def run(n_sessions):
        ts = [threading.Thread(target=worker, args=(t, n_sessions)) for t in range(a.threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dm.sync()
        return time.perf_counter() - t0

    run(4)
    ok[:] = [0] * a.threads
    for x in lat:
        x.clear()
    dt = run(a.sessions)
    n_q = a.sessions * a.gang * len(gangs)
    all_lat = sorted(x for l in lat for x in l)
    print(json.dumps({"threads": a.threads, "gangs": len(gangs), "gang": a.gang, "queries": n_q,
                      "queries_per_s": n_q / dt, "localised": f"{sum(ok)}/{n_q}",
                      "session_ms_p50": all_lat[len(all_lat) // 2] * 1e3,
                      "launches": [capi.gang_counters(g[0]) for g in gangs][:2]}), flush=True)
    for g in gangs:
        for c in reversed(g):
            c.close()
    for dq in dqs:
        dq.close()
    dm.close()


if __name__ == "__main__":
    main()
