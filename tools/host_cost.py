#!/usr/bin/env python3
"""Host-side cost of the asynchronous calls on the headline workload: how long sfmloc_localize_bow_begin (which queues
the whole chain of ~45 launches) and sfmloc_localize_end take on the host, with 1, 4 and 12 contexts in flight."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
import bench  # noqa: E402
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402


def main():
    m = synth.make_map(2, n_views=10000, desc_per_view=2000)
    queries = [synth.make_query(m, 1000 + i, n_feat=2000) for i in range(32)]
    bow, qbow = bench.synth_bow(m, queries)
    params = S.default_params(device=0, profile=0, ransac_round=25)
    dm = S.Map(m.view_id, m.view_off, m.desc, params=params, view_wh=m.view_wh, kpt_xy=m.kpt_xy,
               row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
               bow=bow)
    dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in queries]
    for dq, qb in zip(dqs, qbow):
        dq.set_bow(qb)
    for nctx in (1, 4, 12):
        ctxs = [dm.context() for _ in range(nctx)]
        busy = [False] * nctx
        tb, te = [], []
        N = 1500
        t0 = time.perf_counter()
        for i in range(N):
            k = i % nctx
            if busy[k]:
                a = time.perf_counter()
                ctxs[k].end()
                te.append(time.perf_counter() - a)
            a = time.perf_counter()
            ctxs[k].begin_bow(dqs[i % len(dqs)], None, 100)
            tb.append(time.perf_counter() - a)
            busy[k] = True
        for k in range(nctx):
            if busy[k]:
                ctxs[k].end()
        dt = time.perf_counter() - t0
        tb, te = np.array(tb[nctx:]) * 1e6, np.array(te[nctx:]) * 1e6
        print(f"in flight {nctx:2d}: {N / dt:7.1f} q/s | begin call: median {np.median(tb):6.1f} us mean {tb.mean():6.1f} | "
              f"end call: median {np.median(te):6.1f} us mean {te.mean():6.1f} | host busy in calls "
              f"{(tb.sum() + te.sum()) / 1e6 / dt * 100:4.1f} % of wall", flush=True)
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
