"""GPU: gang sessions (sfmloc_gang_begin / _end, include/sfmloc.h) -- several contexts taken through the same chain
with ONE launch per kernel.  A member's arithmetic is the kernel's own body, so every result must be what the same calls
give one context at a time, bit for bit."""
import numpy as np
import pytest

import sfmlocalization_amd as S
import synthdata as synth
from sfmlocalization_amd import capi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def scene(seed, n_views=60):
    m = synth.make_map(seed, n_views=n_views, desc_per_view=400, views_per_place=10, landmarks_per_place=300,
                       obs_per_view=140)
    rng = np.random.Generator(np.random.PCG64(seed))
    place_bow = rng.uniform(0, 1, (len(m.place_center), 64)).astype(np.float32)
    bow = (place_bow[m.view_place] + rng.normal(0, 0.05, (m.n_views, 64))).astype(np.float32)
    return m, bow, place_bow


def device_map(m, bow, v0=0, v1=None, **params):
    v1 = m.n_views if v1 is None else v1
    r0, r1 = int(m.view_off[v0]), int(m.view_off[v1])
    return S.Map(m.view_id[v0:v1], m.view_off[v0:v1 + 1] - m.view_off[v0], m.desc[r0:r1],
                 params=S.default_params(ransac_round=25, **params), view_wh=m.view_wh[v0:v1], kpt_xy=m.kpt_xy[r0:r1],
                 row_landmark=m.row_landmark[r0:r1], landmark_id=m.landmark_id, landmark_X=m.landmark_X,
                 intrinsic=m.intrinsic, bow=bow[v0:v1])


def same(a, b):
    assert a[0].ok == b[0].ok and a[0].n_inliers == b[0].n_inliers
    assert a[0].n_putative_views == b[0].n_putative_views and a[0].n_geometric_views == b[0].n_geometric_views
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(bits(np.array(a[0].P)), bits(np.array(b[0].P)))
    np.testing.assert_array_equal(bits(np.array(a[0].center)), bits(np.array(b[0].center)))


@pytest.mark.parametrize("own_streams", [False, True])
def test_whole_path_in_a_gang_equals_one_at_a_time(own_streams):
    """sfmloc_localize_bow_begin on 5 contexts inside one session (members with and without a stream of their own), and
    again with a member left out and different queries: the poses, inlier pairs and view counts of the plain calls."""
    m, bow, place_bow = scene(41)
    rng = np.random.Generator(np.random.PCG64(7))
    with device_map(m, bow) as dm:
        qs = [synth.make_query(m, 500 + k, n_feat=700 + 37 * k, n_copies=200, outlier_frac=0.3) for k in range(9)]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        qbow = [(place_bow[q.place] + rng.normal(0, 0.05, 64)).astype(np.float32) for q in qs]
        for dq, b in zip(dqs, qbow):
            dq.set_bow(b)
        ref = [dm.localize_bow(dq, b, 20) for dq, b in zip(dqs, qbow)]
        assert sum(int(r[0].ok) for r in ref) >= 6
        lead = dm.context()
        ctxs = [lead] + [dm.context(share=None if own_streams else lead) for _ in range(4)]
        for first in (0, 4):
            part = list(zip(ctxs, range(first, first + 5)))
            if first:
                part = part[:2] + part[3:]                      # a session need not use every context it could
            with capi.gang([c for c, _ in part]):
                for c, i in part:
                    c.begin_bow(dqs[i], None, 20)
            for c, i in part:
                same(c.end(), ref[i])
        launches, ganged = capi.gang_counters(lead)
        assert ganged >= 15 and launches < 3 * ganged           # the chain went out as gang launches
        # a plain call on a member right after a session sees the session's work as done (stream order)
        ctxs[1].begin_bow(dqs[0], None, 20)
        same(ctxs[1].end(), ref[0])
        # misuse
        with pytest.raises(S.SfmlocError):
            with capi.gang([lead, lead]):
                pass
        with capi.gang([lead, ctxs[1]]):
            with pytest.raises(S.SfmlocError):
                capi._check(capi._L().sfmloc_gang_begin((capi.C.c_void_p * 1)(ctxs[1]._h), 1))
        # a merge-only context (no matching workspace) refuses everything but the merge
        mc = dm.context(merge_only=True)
        with pytest.raises(S.SfmlocError):
            mc.begin_bow(dqs[0], None, 20)
        with pytest.raises(S.SfmlocError):
            mc.begin(dqs[0])
        mc.close()
        # a lender destroyed before its borrowers: its stream lives on until the last of them goes
        lead.close()
        ctxs[2].begin_bow(dqs[1], None, 20)
        same(ctxs[2].end(), ref[1])
        for c in ctxs[1:]:
            c.close()
        for dq in dqs:
            dq.close()


def test_profiled_sessions_and_single_members_run_plainly():
    """params.profile = 1 (stage brackets are events on the stream): a session records nothing, results unchanged."""
    m, bow, place_bow = scene(42, n_views=40)
    with device_map(m, bow, profile=1) as dm:
        qs = [synth.make_query(m, 600 + k, n_feat=600, n_copies=200, outlier_frac=0.3) for k in range(3)]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        for dq, q in zip(dqs, qs):
            dq.set_bow(place_bow[q.place])
        ref = [dm.localize_bow(dq, place_bow[q.place], 10) for dq, q in zip(dqs, qs)]
        ctxs = [dm.context() for _ in range(3)]
        with capi.gang(ctxs):
            for c, dq in zip(ctxs, dqs):
                c.begin_bow(dq, None, 10)
        for c, r in zip(ctxs, ref):
            got = c.end()
            same(got, r)
            assert sum(got[0].stage_seconds[:6]) > 0.0            # the brackets were taken
        assert capi.gang_counters(ctxs[0]) == (0, 0)
        for c in ctxs:
            c.close()
        for dq in dqs:
            dq.close()


def test_consecutive_sessions_under_different_leaders():
    """Members with streams of their own, no host synchronisation between sessions, the leader changing every time: a
    member's stream carries a dependency on the previous session's work (gang_close), so the next leader's stream must
    order itself after it (gang_open) -- the shard stage of dist.py queues batch after batch like this.  The packed
    parts of every session are the ones the same calls give one context at a time."""
    import torch
    m, bow, place_bow = scene(44)
    dev = torch.device("cuda", 0)
    with device_map(m, bow) as dm:
        qs = [synth.make_query(m, 800 + k, n_feat=500 + 150 * k, n_copies=180, outlier_frac=0.3) for k in range(6)]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        B, budget = 3, 3 * 256
        nbytes = capi.packed_bytes(B, budget)
        ctxs = [dm.context() for _ in range(3)]                 # three streams
        plan = [((0, 1, 2), (0, 1, 2)), ((2, 1, 0), (3, 4, 5)), ((1, 0, 2), (5, 0, 3)), ((2, 0, 1), (1, 2, 4)),
                ((0, 2), (4, 5))]

        def run(session):
            parts = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in plan]
            torch.cuda.synchronize()
            for part, (order, queries) in zip(parts, plan):
                cs = [ctxs[k] for k in order]
                work = list(zip(cs, queries))
                if session:
                    with capi.gang(cs):
                        for slot, (c, qi) in enumerate(work):
                            c.shard_begin(dqs[qi])
                            c.shard_export_packed(part.data_ptr(), B, budget, slot)
                else:
                    for slot, (c, qi) in enumerate(work):
                        c.shard_begin(dqs[qi])
                        c.shard_export_packed(part.data_ptr(), B, budget, slot)
                        c.sync()
            for c in ctxs:
                c.sync()
            return [p.cpu().numpy() for p in parts]

        def canon(buf):
            """each query's candidates (where a query's block sits inside the part depends on the order the exports ran
            in, which differs between the two forms; the candidates themselves do not)"""
            from sfmlocalization_amd import dist as D
            return [np.sort(D.unpack_batch(buf, qi).view(np.dtype((np.void, 40))).ravel()).tobytes() for qi in range(B)]

        one = run(False)
        for _ in range(3):
            many = run(True)
            for a, b in zip(one, many):
                assert canon(a) == canon(b)
        assert sum(len(x) for x in canon(one[0])) > 50 * 40
        for c in ctxs:
            c.close()
        for dq in dqs:
            dq.close()


@pytest.mark.parametrize("gang", [3, 16])
def test_sharded_layer_with_gangs_equals_without(gang):
    """dist.HipShardCompute with stage 1 and stage 2 in gang sessions (stream-less members, stage-2 contexts of its own)
    against the same layer one query per launch, through the sharded BoW shortlist and the packed exchange; world 1
    and an emulated second shard's keys / parts are not needed for this: the compute object is what changes."""
    import torch
    from sfmlocalization_amd import dist as D
    m, bow, place_bow = scene(43)
    dev = torch.device("cuda", 0)
    with device_map(m, bow) as dm:
        qs = [synth.make_query(m, 700 + k, n_feat=650, n_copies=200, outlier_frac=0.3) for k in range(11)]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        for dq, q in zip(dqs, qs):
            dq.set_bow(place_bow[q.place])
        out = {}
        for g in (1, gang):
            comp = D.HipShardCompute(dm, n_contexts=2 * g if g > 1 else 4, device=dev, gang=g)
            loc = D.ShardedLocalizer(comp, rank=0, world=1, n_views_global=m.n_views)
            batches = [dqs[:7], dqs[7:], dqs[2:9]]
            out[g] = list(loc.localize_stream(batches, bow_knn=15))
            comp.close()
        for a, b in zip(out[1], out[gang]):
            assert sorted(a) == sorted(b)
            for i in a:
                assert a[i]["ok"] == b[i]["ok"] and a[i]["n_inliers"] == b[i]["n_inliers"]
                np.testing.assert_array_equal(a[i]["pair_qfeat"], b[i]["pair_qfeat"])
                np.testing.assert_array_equal(a[i]["pair_landmark"], b[i]["pair_landmark"])
                np.testing.assert_array_equal(bits(a[i]["P"].ravel()), bits(b[i]["P"].ravel()))
        assert sum(int(r["ok"]) for r in out[1][0].values()) >= 4
        for dq in dqs:
            dq.close()


def test_three_batches_in_flight_give_the_results_of_two(monkeypatch):
    """ShardedLocalizer.localize_stream keeps three batches in flight when the compute object can queue stage 2 without
    waiting for it (HipShardCompute.stage2_begin / _end: stage 2 of batch b is queued, stage 1 of batch b + 2 follows, the
    poses of batch b are collected one batch later; two groups of stage-2 contexts take turns).  Five uneven batches, the
    last one empty-handed for stage 2 on this rank's turn: the same results, in the same order, as the two-deep stream
    (SFMLOC_SHARD_PIPELINE=2) and as batches run one at a time."""
    import torch
    from sfmlocalization_amd import dist as D
    m, bow, place_bow = scene(47)
    dev = torch.device("cuda", 0)
    with device_map(m, bow) as dm:
        qs = [synth.make_query(m, 900 + k, n_feat=600, n_copies=180, outlier_frac=0.3) for k in range(12)]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        for dq, q in zip(dqs, qs):
            dq.set_bow(place_bow[q.place])
        batches = [dqs[:5], dqs[5:6], dqs[6:12], dqs[3:4], dqs[1:9]]
        out = {}
        for mode in ("3", "2", "one at a time"):
            comp = D.HipShardCompute(dm, n_contexts=32, device=dev, gang=16)
            loc = D.ShardedLocalizer(comp, rank=0, world=1, n_views_global=m.n_views)
            if mode == "one at a time":
                out[mode] = [loc.localize_batch(b, gather_results=False, bow_knn=15) for b in batches]
            else:
                monkeypatch.setenv("SFMLOC_SHARD_PIPELINE", mode)
                out[mode] = list(loc.localize_stream(batches, bow_knn=15))
            comp.close()
        assert len(out["3"]) == len(out["2"]) == len(batches)
        for a, b, c in zip(out["3"], out["2"], out["one at a time"]):
            assert sorted(a) == sorted(b) == sorted(c)
            for i in a:
                assert a[i]["fingerprint"] == b[i]["fingerprint"] == c[i]["fingerprint"]
        assert sum(int(r["ok"]) for r in out["3"][0].values()) >= 3
        for dq in dqs:
            dq.close()


def test_a_share_of_stage_2_larger_than_one_group_of_contexts():
    """A rank of 2 owns 128 of a batch's 256 queries and a group of stage-2 contexts holds GANG_MAX = 32: stage2_begin
    queues the share as several sessions on groups made on demand (nothing awaited), where the blocking form alternated two
    groups.  Batches of 40 and 70 queries owned by this one rank (two and three sessions), a small one between them: the
    stream's results = the batches one at a time (blocking form), in order."""
    import torch
    from sfmlocalization_amd import dist as D
    m, bow, place_bow = scene(48)
    dev = torch.device("cuda", 0)
    with device_map(m, bow) as dm:
        qs = [synth.make_query(m, 950 + k, n_feat=500, n_copies=170, outlier_frac=0.3) for k in range(12)]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        for dq, q in zip(dqs, qs):
            dq.set_bow(place_bow[q.place])
        batches = [[dqs[i % 12] for i in range(40)], dqs[:5], [dqs[(5 * i) % 12] for i in range(70)], [dqs[i % 12] for i in range(33)]]
        out = {}
        for mode in ("stream", "one at a time"):
            comp = D.HipShardCompute(dm, n_contexts=32, device=dev, gang=16)
            loc = D.ShardedLocalizer(comp, rank=0, world=1, n_views_global=m.n_views)
            if mode == "stream":
                out[mode] = list(loc.localize_stream(batches, bow_knn=15))
                assert len(comp.ctx2) == 6          # three sessions x two turns
            else:
                out[mode] = [loc.localize_batch(b, gather_results=False, bow_knn=15) for b in batches]
            comp.close()
        for a, b, batch in zip(out["stream"], out["one at a time"], batches):
            assert sorted(a) == sorted(b) == list(range(len(batch)))
            for i in a:
                assert a[i]["fingerprint"] == b[i]["fingerprint"]
        assert sum(int(r["ok"]) for r in out["stream"][0].values()) >= 20
        for dq in dqs:
            dq.close()


@pytest.mark.parametrize("gang", [1, 4])
def test_queries_as_views_into_the_gathered_feature_blocks(gang):
    """images in on several ranks (dist.gather_queries): the owner packs a query's extracted features, the all-gather's
    buffer holds every query of the batch, and the queries are views into it (sfmloc_query_create_view) -- same poses as
    queries uploaded the usual way, through the sharded shortlist and both stages, with and without gangs."""
    import torch
    from sfmlocalization_amd import dist as D
    m, bow, place_bow = scene(44)
    dev = torch.device("cuda", 0)
    with device_map(m, bow) as dm:
        qs = [synth.make_query(m, 800 + k, n_feat=300 + 90 * k, n_copies=150, outlier_frac=0.3) for k in range(6)]
        qbow = [place_bow[q.place] for q in qs]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        for dq, b in zip(dqs, qbow):
            dq.set_bow(b)
        comp = D.HipShardCompute(dm, n_contexts=2 * gang if gang > 1 else 4, device=dev, gang=gang)
        loc = D.ShardedLocalizer(comp, rank=0, world=1, n_views_global=m.n_views)
        ref = loc.localize_batch(dqs, bow_knn=15)
        cap, bow_dim = 832, 64
        own = {i: D.pack_features(q.desc, q.kpt_xy, capi.feat_round_trip(q.kpt_xy), q.width, q.height, b, cap, bow_dim)
               for i, (q, b) in enumerate(zip(qs, qbow))}
        for _ in range(2):                                        # the second batch replaces the first one's views
            views = loc.gather_queries(own, len(qs), cap, bow_dim)
            assert [v.n for v in views] == [q.desc.shape[0] for q in qs]
            got = loc.localize_batch(views, bow_knn=15)
            for i in ref:
                assert got[i]["ok"] == ref[i]["ok"] and got[i]["n_inliers"] == ref[i]["n_inliers"]
                np.testing.assert_array_equal(got[i]["pair_qfeat"], ref[i]["pair_qfeat"])
                np.testing.assert_array_equal(bits(got[i]["P"].ravel()), bits(ref[i]["P"].ravel()))
        assert sum(int(r["ok"]) for r in ref.values()) >= 3
        assert loc.counters()["feature_allgather_bytes_per_batch_per_rank"] > 0
        with pytest.raises(S.SfmlocError):                        # a view's BoW vector is part of the view
            views[0].set_bow(qbow[0])
        with pytest.raises(OverflowError):
            D.pack_features(qs[5].desc, qs[5].kpt_xy, qs[5].kpt_xy, 1, 1, qbow[5], 64, bow_dim)
        comp.close()
        for dq in dqs:
            dq.close()


@pytest.mark.parametrize("world,images", [(4, ""), (8, "vga")])
def test_one_rank_of_n_on_this_gpu(world, images):
    """tools/rank_emulation.py at test size: rank 0 of N through the whole sharded pipeline in gang sessions (the other
    ranks' halves of the exchanges computed beforehand from their shards) -- its poses are the unsharded path's and every
    owned query localises; with `images`, the batch's queries are views into the gathered feature blocks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "rank_emulation.py"), "--of", str(world), "--views", "1600",
           "--desc-per-view", "600", "--nq", "700", "--bow-knn", "40", "--batch", "64", "--queries", "32", "--steps", "2",
           "--warmup", "1", "--gang", "8", "--in-flight", "16"]
    if images:
        cmd += ["--images", images, "--extract-workers", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    n_checked = int(d["same_as_unsharded"].split("/")[1])
    assert d["same_as_unsharded"] == f"{n_checked}/{n_checked}" and n_checked >= 64 // world
    own, total = (int(x) for x in d["localised_of_own"].split("/"))
    assert own == total == 2 * (64 // world) and d["redo"] == 0
    assert d["queries_per_launch_stage1"] == 8
    if images:
        assert d["images_in"]["frames_extracted_per_batch"] == 64 // world
