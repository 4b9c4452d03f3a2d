#!/usr/bin/env python3
"""BASELINE configs[2] (SURVEY 8d cfg 3): 10 k-image / 20 M-descriptor synthetic map on one MI355X, BoW shortlist
(k = 100 of 10 000 views, exact L2 over the .bow matrix) then the whole path on the shortlisted views; plus the
full-bank scan of the same map for the K1 figure at 1.28 GB.  One JSON line per measurement."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402


def main():
    V = int(os.environ.get("CFG3_VIEWS", "10000"))
    t0 = time.perf_counter()
    m = synth.make_map(3, n_views=V, desc_per_view=2000)
    print(json.dumps({"built_map_s": round(time.perf_counter() - t0, 1), "views": V, "rows": int(m.n_rows)}), flush=True)
    rng = np.random.Generator(np.random.PCG64(33))
    n_places = len(m.place_center)
    place_bow = rng.uniform(0, 1, (n_places, 500)).astype(np.float32)
    bow = (place_bow[m.view_place] + rng.normal(0, 0.05, (V, 500))).astype(np.float32)
    queries = [synth.make_query(m, 2000 + i, n_feat=2000) for i in range(8)]
    qbow = [(place_bow[q.place] + rng.normal(0, 0.05, 500)).astype(np.float32) for q in queries]
    params = S.default_params(profile=1, ransac_round=25)
    dm = S.Map(m.view_id, m.view_off, m.desc, params=params, view_wh=m.view_wh, kpt_xy=m.kpt_xy,
               row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
               bow=bow)
    dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in queries]
    # shortlist quality + time
    t = []
    hits = 0
    sels = []
    for i, q in enumerate(queries):
        t1 = time.perf_counter()
        sel = dm.bow_select(qbow[i], 100)
        t.append(time.perf_counter() - t1)
        sels.append(sel)
        hits += int(np.isin(np.nonzero(m.view_place == q.place)[0], sel).sum())
    print(json.dumps({"stage": "K8 bow_select k=100 of %d" % V, "ms_p50": float(np.percentile(t[1:], 50) * 1e3),
                      "place_views_in_shortlist": hits, "of": int(sum((m.view_place == q.place).sum() for q in queries))}),
          flush=True)
    # shortlisted path: latency (one in flight) and throughput (4 in flight)
    ctxs = [dm.context() for _ in range(4)]
    for mode, nfl, steps in (("shortlist, 1 in flight", 1, 64), ("shortlist, 4 in flight", 4, 256)):
        lat, ok = [], 0
        busy = [None] * nfl
        dm.stats_reset()
        tb = [0.0] * nfl
        t1 = time.perf_counter()
        for i in range(steps):
            k = i % nfl
            if busy[k] is not None:
                pose, _, _ = ctxs[k].end()
                lat.append(time.perf_counter() - tb[k])
                ok += int(pose.ok)
            tb[k] = time.perf_counter()
            sel = dm.bow_select(qbow[i % 8], 100)       # the shortlist is part of the query's work
            ctxs[k].begin(dqs[i % 8], sel)
            busy[k] = True
        for k in range(nfl):
            if busy[k] is not None:
                pose, _, _ = ctxs[k].end()
                lat.append(time.perf_counter() - tb[k])
                ok += int(pose.ok)
        dt = time.perf_counter() - t1
        st = dm.stats()
        print(json.dumps({"mode": mode, "queries_per_s": steps / dt, "p50_ms": float(np.percentile(lat, 50) * 1e3),
                          "localised": f"{ok}/{steps}", "k1_ms": st.total_ms[0] / max(1, st.launches[0]),
                          "rows_scanned": int(sum(m.view_off[v + 1] - m.view_off[v] for v in sels[0]))}), flush=True)
    # full bank (no shortlist): K1 at 1.28 GB
    dm.stats_reset()
    lat = []
    for i in range(6):
        t1 = time.perf_counter()
        ctxs[0].begin(dqs[i % 8])
        pose, _, _ = ctxs[0].end()
        lat.append(time.perf_counter() - t1)
    st = dm.stats()
    k1 = st.total_ms[0] / st.launches[0]
    print(json.dumps({"mode": "full bank, 1 in flight", "p50_ms": float(np.percentile(lat, 50) * 1e3), "k1_ms": k1,
                      "pairs_per_s": m.n_rows * 2000 / (k1 * 1e-3), "bank_GBps": m.n_rows * 64 / (k1 * 1e-3) / 1e9,
                      "lane_ops_per_pair": st.hamming_lane_ops / max(1, st.hamming_pairs)}), flush=True)


if __name__ == "__main__":
    main()
