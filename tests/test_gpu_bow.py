"""GPU: BoW shortlist kernels (K7 BoF vector, K8 top-k over the views' .bow vectors) against the oracle:
selected view sets and histogram vectors bit-exact."""
import numpy as np
import pytest

import sfmlocalization_amd as S
import synthdata as synth

pytestmark = pytest.mark.gpu


def test_bow_select_matches_oracle(oracle_c):
    rng = np.random.Generator(np.random.PCG64(5))
    nv = 3000
    bow = np.sqrt(rng.random((nv, 500))).astype(np.float32)
    bow[1500] = bow[100]
    bow[2999] = bow[100]
    m = synth.make_map(1, n_views=nv, desc_per_view=2, views_per_place=100, landmarks_per_place=10, obs_per_view=1)
    with S.Map(m.view_id, m.view_off, m.desc, bow=bow) as dm:
        for trial in range(6):
            q = bow[rng.integers(0, nv)] + rng.normal(0, 0.02, 500).astype(np.float32)
            if trial == 0:
                q = bow[100].copy()                                   # three views tie at distance 0
            for k in (1, 2, 20, 100, 200):
                np.testing.assert_array_equal(dm.bow_select(q, k), oracle_c.bow_select(bow, q, k))
            cand = np.sort(rng.choice(nv, 700, replace=False)).astype(np.uint32)
            np.testing.assert_array_equal(dm.bow_select(q, 100, cand), oracle_c.bow_select(bow, q, 100, cand))
        assert list(dm.bow_select(bow[100], 2)) == [100, 1500]
        with pytest.raises(S.SfmlocError):
            dm.bow_select(q, nv)                                      # CV_Assert(knn < viewList.size())
    with S.Map(m.view_id, m.view_off, m.desc) as dm2:
        with pytest.raises(S.SfmlocError):
            dm2.bow_select(q, 5)                                      # no .bow vectors in this map


def test_bof_vector_matches_oracle(oracle_c):
    rng = np.random.Generator(np.random.PCG64(6))
    xs = (np.arange(50) * 6 + 3).astype(np.float32)
    kxy = np.repeat(np.stack(np.meshgrid(xs, xs), -1).reshape(-1, 2), 4, axis=0)      # 50 x 50 x 4 scales
    desc = rng.integers(0, 256, (len(kxy), 61)).astype(np.float32)
    K, n_pca = 100, 32
    mean = desc.mean(0).astype(np.float32)
    evec = np.linalg.qr(rng.normal(size=(61, 61)))[0][:n_pca].astype(np.float32)
    evals = np.linspace(5000, 100, n_pca).astype(np.float32)
    proj = ((desc - mean) @ evec.T) / evals
    centers = proj[rng.choice(len(proj), K, replace=False)].astype(np.float32)
    b = S.BofModel(centers, 61, pca_mean=mean, pca_eigvec=evec, pca_eigval=evals, n_pca=n_pca)
    assert b.dim == 500
    got = b.compute(desc, kxy)
    exp = oracle_c.bof(desc, kxy, centers, pca_mean=mean, pca_eigvec=evec, pca_eigval=evals, n_pca=n_pca)
    np.testing.assert_array_equal(got.view(np.uint64), exp.view(np.uint64))
    b.close()
    b2 = S.BofModel(desc[:30].copy(), 61, use_pyramid=False, norm="L2")
    np.testing.assert_array_equal(b2.compute(desc, kxy).view(np.uint64),
                                  oracle_c.bof(desc, kxy, desc[:30], levels=1, norm_type=1).view(np.uint64))
    b2.close()


@pytest.mark.parametrize("K,n_pca,n", [(100, 32, 9999), (100, 32, 31), (100, 32, 1), (37, 0, 5003), (400, 48, 3001), (7, 61, 650),
                                       (7000, 8, 400)])     # (a vocabulary too large for either LDS form: both fallbacks)
def test_bof_assignment_forms_match_oracle(oracle_c, K, n_pca, n):
    """The LDS-tiled assignment (models that fit its layout: the reference's 61 -> 32, K = 100) and the
    thread-per-descriptor one (anything larger: K = 400 x 48 does not fit) give the oracle's vector bit for bit: ragged
    last tile, a single descriptor, no PCA, repeated centres (the first of equal distances wins: BoFSpatialPyramids.cpp:
    230-243)."""
    rng = np.random.Generator(np.random.PCG64(K * 1000 + n))
    kxy = rng.uniform(0, 300, (n, 2)).astype(np.float32)
    desc = rng.integers(0, 256, (n, 61)).astype(np.float32)
    cdim = n_pca if n_pca else 61
    pca = {}
    if n_pca:
        pca = dict(pca_mean=rng.uniform(0, 255, 61).astype(np.float32), pca_eigvec=rng.normal(size=(n_pca, 61)).astype(np.float32),
                   pca_eigval=rng.uniform(0.5, 4.0, n_pca).astype(np.float32), n_pca=n_pca)
    centers = (rng.normal(size=(K, cdim)) * (30 if n_pca else 1) + (0 if n_pca else 128)).astype(np.float32)
    centers[K // 2] = centers[1]                                # a tie for every descriptor nearest to centre 1
    b = S.BofModel(centers, 61, **pca)
    got = b.compute(desc, kxy)
    exp = oracle_c.bof(desc, kxy, centers, **pca)
    np.testing.assert_array_equal(got.view(np.uint64), exp.view(np.uint64))
    b.close()


def test_sharded_shortlist_equals_unsharded(oracle_c):
    """sfmloc_bow_distances per shard + the (distance, view id) merge of dist.py = sfmloc_bow_select on the whole map;
    an empty local selection scans nothing (it must not fall back to "all views")."""
    from sfmlocalization_amd import dist as D
    rng = np.random.Generator(np.random.PCG64(14))
    m = synth.make_map(8, n_views=30, desc_per_view=150, views_per_place=10, landmarks_per_place=120, obs_per_view=60)
    bow = rng.uniform(0, 1, (30, 40)).astype(np.float32)
    bow[7] = bow[3]                                                     # exact tie between two views
    qbow = (bow[3] + rng.normal(0, 0.01, 40)).astype(np.float32)
    k = 6
    kw = dict(view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
              landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    with S.Map(m.view_id, m.view_off, m.desc, bow=bow, **kw) as full:
        ref = full.bow_select(qbow, k)
        d_all = full.bow_distances(qbow)
        assert set(np.lexsort((m.view_id, d_all))[:k].tolist()) == set(ref.tolist()) and d_all[7] == d_all[3]
    cuts = [(0, 11), (11, 12), (12, 30)]
    shards, dists = [], []
    for a, b in cuts:
        r0, r1 = int(m.view_off[a]), int(m.view_off[b])
        sm = S.Map(m.view_id[a:b], m.view_off[a:b + 1] - m.view_off[a], m.desc[r0:r1], bow=bow[a:b],
                   view_wh=m.view_wh[a:b], kpt_xy=m.kpt_xy[r0:r1], row_landmark=m.row_landmark[r0:r1],
                   landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic)
        shards.append(sm)
        dists.append(sm.bow_distances(qbow))
        np.testing.assert_array_equal(dists[-1], d_all[a:b])
    per_rank = []
    for (a, b), d in zip(cuts, dists):
        ids = m.view_id[a:b].astype(np.int64)
        order = np.lexsort((ids, d))[:k]
        mm = np.full((k, 2), np.inf)
        mm[:len(order), 0], mm[:len(order), 1] = d[order], ids[order]
        per_rank.append(mm)
    gathered = np.stack(per_rank)
    got = []
    q = synth.make_query(m, 5, n_feat=300, n_copies=100)
    for (a, b), d, sm in zip(cuts, dists, shards):
        sel = D.merge_bow_shortlists(d, m.view_id[a:b], k, 3, lambda mine: gathered)
        got += [a + int(i) for i in sel]
        dq = sm.query(q.desc, q.kpt_xy, q.width, q.height)
        c = sm.context()
        c.shard_begin(dq, sel)                                         # possibly an EMPTY selection
        c.sync()
        sm.match_putative(dq, sel)
        cnt = sm.putative_read()[0]
        assert cnt.sum() == 0 or len(sel) > 0
        assert (cnt[np.setdiff1d(np.arange(b - a), sel)] == 0).all()
        c.close()
        dq.close()
    assert sorted(got) == ref.tolist()
    for sm in shards:
        sm.close()


def test_shortlist_chain_on_the_device_equals_two_steps():
    """sfmloc_localize_bow_begin (shortlist -> block list -> scan, all on the device) against sfmloc_bow_select +
    sfmloc_localize_begin: the same pose, inliers and pairs bit for bit -- on a ragged map (views of very different
    sizes, empty ones, neighbours sharing a 64-row block), with and without a candidate restriction, with more than
    1024 views selected, with the exact (unscreened) scan, and when the shortlist does not apply (knn >= candidates)."""
    rng = np.random.Generator(np.random.PCG64(77))
    for seed, kw, nq in ((31, dict(n_views=60, desc_per_view=400, views_per_place=10, landmarks_per_place=300,
                                   obs_per_view=140, ragged=True), 900),
                         (32, dict(n_views=2600, desc_per_view=40, views_per_place=20, landmarks_per_place=60,
                                   obs_per_view=30), 800),
                         (33, dict(n_views=48, desc_per_view=500, views_per_place=8, landmarks_per_place=300,
                                   obs_per_view=150), 300)):      # nq < 768: the exact kernel, query rows split
        m = synth.make_map(seed, **kw)
        nv = m.n_views
        proto = rng.uniform(0, 1, (len(m.place_center), 500)).astype(np.float32)
        bow = (proto[m.view_place] + rng.normal(0, 0.05, (nv, 500))).astype(np.float32)
        with S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25), view_wh=m.view_wh,
                   kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X,
                   intrinsic=m.intrinsic, bow=bow) as dm:
            ctx = dm.context()
            n_ok = 0
            for trial in range(3):
                q = synth.make_query(m, 500 + trial, n_feat=nq, place=trial % len(m.place_center))
                dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
                qb = (proto[q.place] + rng.normal(0, 0.05, 500)).astype(np.float32)
                cands = [None, np.sort(rng.choice(nv, nv // 2, replace=False)).astype(np.uint32)]
                for cand in cands:
                    n_cand = nv if cand is None else len(cand)
                    for knn in (1, 7, min(n_cand - 1, 1500), n_cand, n_cand + 5):
                        if knn < n_cand:
                            sel = dm.bow_select(qb, knn, cand)
                        else:
                            sel = cand                               # the shortlist does not apply
                        ctx.begin(dq, sel)
                        ref = ctx.end()
                        ctx.begin_bow(dq, qb, knn, cand)
                        got = ctx.end()
                        tag = f"seed {seed} trial {trial} knn {knn} cand {n_cand}"
                        assert got[0].ok == ref[0].ok and got[0].n_inliers == ref[0].n_inliers, tag
                        n_ok += int(got[0].ok)
                        assert got[0].n_putative_views == ref[0].n_putative_views, tag
                        assert got[0].n_matches_2d3d == ref[0].n_matches_2d3d, tag
                        np.testing.assert_array_equal(got[1], ref[1], err_msg=tag)
                        np.testing.assert_array_equal(got[2], ref[2], err_msg=tag)
                        np.testing.assert_array_equal(np.array(got[0].P).view(np.uint64),
                                                      np.array(ref[0].P).view(np.uint64), err_msg=tag)
                dq.close()
            ctx.close()
            assert n_ok >= 6, (seed, n_ok)       # the comparison is not between two failures
    # errors: a map without .bow vectors
    m = synth.make_map(34, n_views=20, desc_per_view=100, views_per_place=10, landmarks_per_place=80, obs_per_view=40)
    with S.Map(m.view_id, m.view_off, m.desc, view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark,
               landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic) as dm:
        q = synth.make_query(m, 1, n_feat=100)
        dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
        ctx = dm.context()
        with pytest.raises(S.SfmlocError):
            ctx.begin_bow(dq, np.zeros(500, np.float32), 5)
        ctx.close()
        dq.close()


def test_sharded_shortlist_chain_on_the_device_equals_unsharded():
    """sfmloc_shard_bow_keys -> (all-gather of the key lists) -> sfmloc_shard_begin_bow -> parts -> sfmloc_merge_begin
    over three shards of one map on one GPU -- a ragged map, tied BoW distances, a one-view shard, knn larger than a
    shard -- against sfmloc_localize_bow on the whole map: pose, inliers, pairs bit for bit.  The same through
    dist.ShardedLocalizer (world 1, the resident query BoW vector, an exchange capacity small enough to force the
    second exchange at full capacity)."""
    import torch
    from sfmlocalization_amd import dist as D
    rng = np.random.Generator(np.random.PCG64(91))
    m = synth.make_map(41, n_views=40, desc_per_view=400, views_per_place=10, landmarks_per_place=300, obs_per_view=140,
                       ragged=True, view_id_stride=3)
    nv = m.n_views
    proto = rng.integers(0, 3, (len(m.place_center), 24)).astype(np.float32)
    bow = (proto[m.view_place] + rng.integers(0, 2, (nv, 24))).astype(np.float32)      # integer-valued: many exact ties
    kw = dict(params=S.default_params(ransac_round=25), landmark_id=m.landmark_id, landmark_X=m.landmark_X,
              intrinsic=m.intrinsic)
    full = S.Map(m.view_id, m.view_off, m.desc, view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark,
                 bow=bow, **kw)
    cuts = [(0, 17), (17, 18), (18, nv)]
    shards = []
    for a, b in cuts:
        r0, r1 = int(m.view_off[a]), int(m.view_off[b])
        shards.append(S.Map(m.view_id[a:b], m.view_off[a:b + 1] - m.view_off[a], m.desc[r0:r1], view_wh=m.view_wh[a:b],
                            kpt_xy=m.kpt_xy[r0:r1], row_landmark=m.row_landmark[r0:r1], bow=bow[a:b], **kw))
    ctxs = [sm.context() for sm in shards]
    cap = 4096
    pb = D.part_bytes(cap)
    n_ok = 0
    from sfmlocalization_amd import capi
    knns = (1, 5, 12, 25)
    for trial in range(4):
        q = synth.make_query(m, 300 + trial, n_feat=900, place=trial % len(m.place_center))
        qb = (proto[q.place] + rng.integers(0, 2, 24)).astype(np.float32)
        fq = full.query(q.desc, q.kpt_xy, q.width, q.height)
        # the four shortlist sizes also form one "batch" of the packed exchange (one buffer per shard); trial 3 gets a
        # budget that cannot hold it
        budget = 4 * 2048 if trial < 3 else 40
        ppb = capi.packed_bytes(len(knns), budget)
        assert ppb == D.packed_bytes(len(knns), budget)
        packed = torch.zeros((3, ppb), dtype=torch.uint8, device="cuda")
        refs_k = []
        for ki, knn in enumerate(knns):
            ref = full.localize_bow(fq, qb, knn)
            refs_k.append(ref)
            keys = torch.zeros((3, knn), dtype=torch.int64, device="cuda")
            parts = torch.zeros((3, pb), dtype=torch.uint8, device="cuda")
            sqs = [sm.query(q.desc, q.kpt_xy, q.width, q.height) for sm in shards]
            for s, (c, sq) in enumerate(zip(ctxs, sqs)):
                c.shard_bow_keys(sq, knn, keys.data_ptr() + s * knn * 8, bow=qb)
                c.sync()
            # the key lists are the shard's knn best (float32 distance bits << 32 | view id), padded with ~0
            hk = keys.cpu().numpy().view(np.uint64)
            d_all = full.bow_distances(qb)
            for s, (a, b) in enumerate(cuts):
                exp = np.sort(D.bow_key(d_all[a:b], m.view_id[a:b]))[:knn]
                np.testing.assert_array_equal(np.sort(hk[s][hk[s] != D.BOW_KEY_PAD]), exp)
            glob = set(full.bow_select(qb, knn).tolist())
            got_sel = set()
            for s, (c, sq) in enumerate(zip(ctxs, sqs)):
                got_sel |= set(cuts[s][0] + int(i) for i in D.select_from_keys(hk, knn, m.view_id[cuts[s][0]:cuts[s][1]]))
                c.shard_begin_bow(sq, keys.data_ptr(), 3, knn)
                c.shard_export(parts.data_ptr() + s * pb, cap)
                c.shard_export_packed(packed.data_ptr() + s * ppb, len(knns), budget, ki)
                c.sync()
            assert got_sel == glob
            ctxs[2].merge_begin(sqs[2], parts.data_ptr(), 3, cap)
            pose, pq, pl = ctxs[2].end()
            tag = f"trial {trial} knn {knn}"
            assert pose.ok == ref[0].ok and pose.n_inliers == ref[0].n_inliers, tag
            assert pose.n_matches_2d3d == ref[0].n_matches_2d3d, tag
            np.testing.assert_array_equal(pq, ref[1], err_msg=tag)
            np.testing.assert_array_equal(pl, ref[2], err_msg=tag)
            np.testing.assert_array_equal(np.array(pose.P).view(np.uint64), np.array(ref[0].P).view(np.uint64), err_msg=tag)
            n_ok += int(pose.ok)
            for sq in sqs:
                sq.close()
        # the packed parts: per shard the candidates of the whole batch back to back, equal (as sets per query) to the
        # plain parts; merged through sfmloc_merge_begin_packed -> the same poses
        hp = packed.cpu().numpy()
        totals = hp[:, :16].view(np.uint32)
        if trial < 3:
            assert (totals[:, 3] == 0).all() and (totals[:, 0] <= budget).all()
            mq = shards[0].query(q.desc, q.kpt_xy, q.width, q.height)
            for ki, knn in enumerate(knns):
                ctxs[0].merge_begin_packed(mq, packed.data_ptr(), 3, len(knns), budget, ki)
                pose, pq, pl = ctxs[0].end()
                ref = refs_k[ki]
                assert pose.ok == ref[0].ok and pose.n_inliers == ref[0].n_inliers, (trial, knn)
                np.testing.assert_array_equal(pq, ref[1])
                np.testing.assert_array_equal(np.array(pose.P).view(np.uint64), np.array(ref[0].P).view(np.uint64))
            mq.close()
        else:   # the budget does not hold the batch: the flag is set, the total still counts everything
            assert (totals[:, 0] > budget).any() and (totals[totals[:, 0] > budget, 3] & 1).all()
        fq.close()
    assert n_ok >= 6, n_ok
    for c in ctxs:
        c.close()
    # the torch.distributed layer, one rank, the whole map as the only shard
    qs, refs = [], []
    for trial in range(5):
        q = synth.make_query(m, 400 + trial, n_feat=900, place=trial % len(m.place_center))
        qb = (proto[q.place] + rng.integers(0, 2, 24)).astype(np.float32)
        dq = full.query(q.desc, q.kpt_xy, q.width, q.height)
        dq.set_bow(qb)
        qs.append(dq)
        refs.append(full.localize_bow(dq, qb, 9))
    comp = D.HipShardCompute(full, n_contexts=2)
    for xcap in (2048, 4):                    # 4: the batch is exchanged again with a budget that fits
        loc = D.ShardedLocalizer(comp, budget_per_query=xcap, rank=0, world=1, n_views_global=nv)
        outs = list(loc.localize_stream([qs[:2], qs[2:]], bow_knn=9))
        res = {0: outs[0][0], 1: outs[0][1], 2: outs[1][0], 3: outs[1][1], 4: outs[1][2]}
        for i, ref in enumerate(refs):
            assert res[i]["ok"] == bool(ref[0].ok), (xcap, i)
            np.testing.assert_array_equal(res[i]["pair_qfeat"], ref[1])
            if ref[0].ok:
                np.testing.assert_array_equal(res[i]["P"].ravel().view(np.uint64), np.array(ref[0].P).view(np.uint64))
        cn = loc.counters()
        # (the second batch was begun -- with the small budget -- before the first one's headers came back)
        redo = cn["batches_exchanged_again_with_a_larger_budget"]
        assert cn["batches"] == 2 and (redo in (1, 2) if xcap == 4 else redo == 0), cn
        assert cn["max_candidates_of_one_shard_for_one_batch"] > 4 and loc.budget_per_query >= xcap
        assert cn["bow_key_allgather_bytes_per_batch_per_rank"] == 2.5 * 9 * 8
    comp.close()
    for dq in qs:
        dq.close()
    for sm in shards:
        sm.close()
    full.close()
