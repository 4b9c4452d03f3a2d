"""Does a gang session of image-in frames really issue ONE launch per kernel?  (sfmloc_gang_counters around a session of G
frames taken the way bench.py's image-in leg takes them, and around the same frames as uploaded queries.)"""
import os
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sfmlocalization_amd as S
import imageworld as iw
from sfmlocalization_amd import capi, engine, fileio

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 640, 480
rng = np.random.Generator(np.random.PCG64(77))
import torch
tdev = torch.device("cuda", 0)
atlas0 = iw.make_atlas(901, 1, 1600, tdev)
Rs, Cs = iw.cameras(rng, 6, (0.0, 0.0), 16.0)
train = iw.render(atlas0, 100.0, Rs, Cs, 800.0, W, H)
grid = engine.dense_grid_keypoints(300)
ak300 = S.Akaze(300, 300, 4, 4, 0.001)
feats = [ak300.compute(capi.dense_gray(np.stack([g, g, g], 2), 300), grid)[0][:, :61].astype(np.float32) for g in train]
ak300.close()
pca, bowm = iw.train_bow_model(np.concatenate(feats)[::3], rng)
tmp = tempfile.TemporaryDirectory()
bow_file, pca_file = os.path.join(tmp.name, "BOWfile.yml"), os.path.join(tmp.name, "PCAfile.yml")
fileio.write_cv_yaml(pca_file, pca)
fileio.write_cv_yaml(bow_file, bowm)
dense0 = engine.DenseBow(bow_file, pca_file)
world = iw.build(S, 31, 200, 16, tiles=2, dense_bow=dense0)
m = world.m
dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25), view_wh=m.view_wh, kpt_xy=m.kpt_xy,
           row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic, bow=world.bow)
lead = dm.context()
cs = [lead] + [dm.context(share=lead) for _ in range(G - 1)]
es = [S.Akaze(W, H) for _ in range(G)]
for e in es:
    e.share_stream(lead)
ibs = [S.ImgBow.from_files(bow_file, pca_file, W, H, 1) for _ in range(G)]
bow_ctx = dm.context(merge_only=True)
for ib in ibs:
    ib.share_stream(bow_ctx)
frames = [np.ascontiguousarray(f) for f in world.frames]
busy = dm.context()          # another context with work queued: the GPU counts as shared
for rep in range(3):
    idx = list(range(rep * G, rep * G + G))
    for i, ib in zip(idx, ibs):
        ib.compute(frames[i], None, want_vector=False)
    ns = S.Akaze.detect_resident_batch(es, [frames[i] for i in idx])
    for ib, c in zip(ibs, cs):
        ib.order_before(c)
    qs = [e.query_view(dm, c, ib.vector_dev()) for e, c, ib in zip(es, ns, ibs)]
    a0 = capi.gang_counters(lead)
    with capi.gang(cs):
        for c, dq in zip(cs, qs):
            c.begin_bow(dq, None, 100)
    a1 = capi.gang_counters(lead)
    ends = [c.end() for c in cs]
    a2 = capi.gang_counters(lead)
    print(f"resident views, session of {G}: launches {a1[0] - a0[0]} of which ganged {a1[1] - a0[1]}; after end(): +{a2[0] - a1[0]} / +{a2[1] - a1[1]}; "
          f"2d3d {[e[0].n_matches_2d3d for e in ends]}", flush=True)
    for dq in qs:
        dq.close()
