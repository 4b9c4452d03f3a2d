"""Where k_p3p_seq's time goes (diagnostic build: make -C sfmlocalization_amd/csrc EXTRA=-DSFMLOC_SEQ_TIMING OBJDIR=../build_t
OUT=../lib/libsfmloc_hip_t.so; run with SFMLOC_LIB_PATH=.../libsfmloc_hip_t.so SFMLOC_P3P_SEQ=2).  Headline-like queries,
one at a time; the kernel prints its own phase totals."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sfmlocalization_amd as S
import synthdata as synth

m = synth.make_map(2, n_views=200, desc_per_view=2000)
dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25), view_wh=m.view_wh, kpt_xy=m.kpt_xy,
           row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic)
ctx = dm.context()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    q = synth.make_query(m, 1000 + i, n_feat=2000)
    dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
    t = time.perf_counter()
    ctx.begin(dq)
    p, pq, pl = ctx.end()
    print(f"query {i}: {1e3 * (time.perf_counter() - t):.2f} ms, 2D-3D {p.n_matches_2d3d}, inliers {p.n_inliers}", flush=True)
    dq.close()
ctx.close()
dm.close()
