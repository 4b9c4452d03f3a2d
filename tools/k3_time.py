#!/usr/bin/env python3
"""K3 (F-matrix AC-RANSAC) and K5 (P3P AC-RANSAC) stage time against the iteration budget, on the bench workload's
geometry (a 100-view slice: the query's place is what matters).  One JSON line per setting."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402


def main():
    m = synth.make_map(1, 100, desc_per_view=2000)
    q = synth.make_query(m, 11, n_feat=2000)
    for rounds in (25, 100, 400):
        params = S.default_params(profile=1, ransac_round=rounds)
        with S.Map(m.view_id, m.view_off, m.desc, params=params, kpt_xy=m.kpt_xy, view_wh=m.view_wh,
                   row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic) as dm:
            dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            dm.match_putative(dq)
            cnt = dm.putative_read()[0]
            dm.geometric_filter(dq)
            dm.sync()
            dm.stats_reset()
            for _ in range(10):
                dm.geometric_filter(dq)
            dm.sync()
            st = dm.stats()
            gcnt = dm.geometric_read()[0]
            print(json.dumps({"ransac_round": rounds, "k3_ms": st.total_ms[2] / st.launches[2],
                              "views_ge16": int((cnt >= 16).sum()), "max_putative": int(cnt.max()),
                              "mean_putative_ge16": float(cnt[cnt >= 16].mean()),
                              "views_geometric": int((gcnt > 0).sum())}), flush=True)
            dq.close()


if __name__ == "__main__":
    main()
