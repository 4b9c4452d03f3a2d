#!/usr/bin/env python3
"""Create / use / destroy every handle type repeatedly and watch the device's free memory (hipMemGetInfo through
torch): a leak shows as a steady drop."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402

m = synth.make_map(3, n_views=30, desc_per_view=400, views_per_place=10, landmarks_per_place=300, obs_per_view=150)
q = synth.make_query(m, 5, n_feat=800, n_copies=250)
img = synth.texture_image(3, 240, 320)
bgr = np.stack([img, img, img], 2)
free = []
for it in range(60):
    with S.Map(m.view_id, m.view_off, m.desc, view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark,
               landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
               bow=np.random.rand(30, 16).astype(np.float32)) as dm:
        dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
        ctxs = [dm.context() for _ in range(3)]
        for c in ctxs:
            c.begin(dq)
        for c in ctxs:
            c.end()
        dm.localize(dq)
        dm.bow_select(np.random.rand(16).astype(np.float32), 5)
        ctxs[0].begin_bow(dq, np.random.rand(16).astype(np.float32), 5)
        ctxs[0].end()
        dm.localize_bow(dq, np.random.rand(16).astype(np.float32), 5)
        qv = dm.query_from_view(2)
        dm.match_one_to_one(qv, np.array([1], np.uint32))
        dm.track(3)
        qv.close()
        # gang sessions: members without a stream, a merge-only context, a session's events and records
        from sfmlocalization_amd import capi
        members = [dm.context(share=ctxs[0]) for _ in range(2)]
        mo = dm.context(merge_only=True)
        dq.set_bow(np.random.rand(16).astype(np.float32))
        with capi.gang([ctxs[0]] + members):
            for c in [ctxs[0]] + members:
                c.begin_bow(dq, None, 5)
        for c in [ctxs[0]] + members:
            c.end()
        with capi.gang([ctxs[1], ctxs[2]]):          # members with streams of their own
            ctxs[1].begin(dq)
            ctxs[2].begin(dq)
        ctxs[1].end()
        ctxs[2].end()
        mo.close()
        for c in members:
            c.close()
        for c in ctxs:
            c.close()
        dq.close()
    ak = S.Akaze(320, 240)
    ak.detect_and_compute(img)
    ak2 = [S.Akaze(320, 240) for _ in range(3)]
    S.Akaze.detect_and_compute_batch(ak2, [img, img, img])
    for e in ak2:
        e.close()
    ak.close()
    S.dense_gray(bgr, 64)
    with S.Undistorter(np.array([[300.0, 0, 160], [0, 300, 120], [0, 0, 1]]), [-0.2, 0.05, 0, 0], 320, 240) as ud:
        ud.apply(bgr)
        ud.apply(img)
    if it % 10 == 9:
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
        print(it + 1, "iterations, free MiB:", free[-1] // (1 << 20), flush=True)
drop = (free[0] - free[-1]) / (1 << 20)
print("drop between iteration 10 and 60: %.1f MiB" % drop)
assert drop < 64, "device memory keeps shrinking"
print("OK")
