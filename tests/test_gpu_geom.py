"""GPU: the f64 geometry kernels (F-matrix AC-RANSAC, 2D-3D de-duplication, P3P AC-RANSAC, KRt) through
the C ABI against the C oracle.  Bar (BASELINE.json north_star): inlier sets and indices bit-exact,
[R|t] within 1e-4 -- the kernels are built to reproduce the oracle's doubles exactly, so poses are compared
for equality first and the 1e-4 bound is the fallback the test states."""
import numpy as np
import pytest

import sfmlocalization_amd as S
import synthdata as synth
from sfmlocalization_amd import capi
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-4


def bits(a):
    """Bit patterns, with every NaN mapped to one canonical pattern: the sign/payload of a NaN produced by an
    invalid operation differs between x86 and gfx950, and NaN residuals are canonicalised to +inf anyway."""
    a = np.ascontiguousarray(a, np.float64)
    b = a.view(np.uint64).copy()
    b[np.isnan(a)] = 0x7FF8000000000000
    return b


def test_f64_primitives_bit_exact(oracle_c):
    rng = np.random.Generator(np.random.PCG64(11))
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 3000), rng.uniform(0.5, 2.0, 3000),
                        np.arange(1, 5000, dtype=np.float64), [1.19e-7, 5e-324, 1e308]])
    got = S.debug_math(0, x, 1)[:, 0]
    exp = np.array([oracle_c.det_log10(float(v)) for v in x])
    np.testing.assert_array_equal(bits(got), bits(exp))
    # IEEE sqrt and division (what every other block is built from)
    a = 10.0 ** rng.uniform(-150, 150, 20000) * rng.choice([1.0, 3.0, 7.0], 20000)
    b = 10.0 ** rng.uniform(-150, 150, 20000)
    got = S.debug_math(1, np.stack([a, b], 1), 2)
    np.testing.assert_array_equal(bits(got[:, 0]), bits(np.sqrt(a)))
    np.testing.assert_array_equal(bits(got[:, 1]), bits(a / b))


def test_wave_sort_fast_is_the_exact_order():
    """K3 / K5 sort a model's residuals with the index packed into the key's low bits (acransac.hip wave_sort_fast);
    the order must be the exact (key bits, index) order in every case, including the ones that force its fallback."""
    rng = np.random.default_rng(77)
    for P in (64, 128, 256, 512, 1024):
        rows = []
        for case in range(24):
            n = int(rng.integers(1, P + 1)) if case % 3 else P
            v = rng.random(P) * 10.0 ** rng.integers(-12, 6)
            if case % 4 == 1:      # exact duplicates
                v[rng.integers(0, P, P // 2)] = v[rng.integers(0, P, P // 2)]
            if case % 4 == 2:      # keys that differ only in the bits the index displaces
                base = v[:P // 4].view(np.uint64) & ~np.uint64(1023)
                v[:P // 4] = (base | rng.integers(0, 1024, P // 4).astype(np.uint64)).view(np.float64)
                v[P // 4:P // 2] = (base | rng.integers(0, 1024, P // 4).astype(np.uint64)).view(np.float64)
                v = rng.permutation(v)
            if case % 4 == 3:      # zeros, infinities
                v[rng.integers(0, P, 5)] = 0.0
                v[rng.integers(0, P, 5)] = np.inf
            rows.append(np.concatenate([v, [float(n)]]))
        x = np.array(rows)
        got = S.debug_math(9, x, 2 * P)
        for row, out in zip(x, got):
            n = int(row[P])
            key = row[:P].copy().view(np.uint64)
            key[n:] = np.uint64(0xFFFFFFFFFFFFFFFF)
            order = np.lexsort((np.arange(P), key))
            assert np.array_equal(out[:P].astype(np.int64), order), (P, n)
            assert np.array_equal(np.ascontiguousarray(out[P:2 * P]).view(np.uint64)[:n], key[order][:n]), (P, n)


def test_polynomial_solvers_bit_exact(oracle_c):
    rng = np.random.Generator(np.random.PCG64(12))
    c = rng.normal(size=(500, 4)) * 10.0 ** rng.uniform(-2, 2, (500, 4))
    c[:20, 0] = 0.0                                   # degenerate leading coefficient
    got = S.debug_math(2, c, 4)
    for row, g in zip(c, got):
        r = oracle_c.solve_cubic(*row)
        assert int(g[0]) == len(r)
        np.testing.assert_array_equal(bits(g[1:1 + len(r)]), bits(r))
    a = rng.normal(size=(500, 5)) * 10.0 ** rng.uniform(-1, 1, (500, 5))
    a[:10, 3] = 0.0
    a[:10, 1] = 0.0                                   # biquadratic branch
    got = S.debug_math(3, a, 4)
    for row, g in zip(a, got):
        np.testing.assert_array_equal(bits(g), bits(oracle_c.solve_quartic_real(row)))


def test_minimal_solvers_bit_exact(oracle_c):
    rng = np.random.Generator(np.random.PCG64(13))
    n = 200
    x1 = rng.uniform(-0.5, 0.5, (n, 7, 2))
    x2 = x1 + rng.normal(0, 0.05, (n, 7, 2))
    got = S.debug_math(4, np.concatenate([x1.reshape(n, 14), x2.reshape(n, 14)], 1), 28)
    got_w = S.debug_math(8, np.concatenate([x1.reshape(n, 14), x2.reshape(n, 14)], 1), 28)  # wave-parallel form
    for i in range(n):
        F = oracle_c.seven_point(x1[i], x2[i])
        assert int(got[i, 0]) == len(F) and int(got_w[i, 0]) == len(F)
        np.testing.assert_array_equal(bits(got[i, 1:1 + 9 * len(F)]), bits(F.ravel()))
        np.testing.assert_array_equal(bits(got_w[i, 1:1 + 9 * len(F)]), bits(F.ravel()))
    # degenerate samples (repeated points, collinear, zeros): both forms must agree with the oracle bit for bit
    xd1 = x1[:40].copy()
    xd2 = x2[:40].copy()
    xd1[:10, 3] = xd1[:10, 2]
    xd2[:10, 3] = xd2[:10, 2]
    xd1[10:20, :, 1] = 0.0
    xd2[20:30] = 0.0
    xd1[30:40] = 0.0
    xd2[30:40] = 0.0
    din = np.concatenate([xd1.reshape(40, 14), xd2.reshape(40, 14)], 1)
    for op in (4, 8):
        gd = S.debug_math(op, din, 28)
        for i in range(40):
            F = oracle_c.seven_point(xd1[i], xd2[i])
            assert int(gd[i, 0]) == len(F), (op, i)
            np.testing.assert_array_equal(bits(gd[i, 1:1 + 9 * len(F)]), bits(F.ravel()))
    x = rng.uniform(-0.4, 0.4, (n, 3, 2))
    X = rng.uniform(-3, 3, (n, 3, 3)) + np.array([0, 0, 9.0])
    got = S.debug_math(5, np.concatenate([x.reshape(n, 6), X.reshape(n, 9)], 1), 49)
    for i in range(n):
        M = oracle_c.p3p_kneip(x[i], X[i])
        assert int(got[i, 0]) == len(M)
        np.testing.assert_array_equal(bits(got[i, 1:1 + 12 * len(M)]), bits(M.ravel()))
    P = rng.normal(size=(n, 12))
    got = S.debug_math(6, P, 24)
    for i in range(n):
        K, R, t, c = oracle_c.krt_from_p(P[i])
        np.testing.assert_array_equal(bits(got[i]), bits(np.concatenate([K.ravel(), R.ravel(), t, c])))


def test_sampling_matches(oracle_c):
    seed = 0x5f3759df12345678
    rows = np.array([[seed & 0xFFFFFFFF, seed >> 32, n, stage, stream, it]
                     for n in (8, 16, 100, 2000) for stage in (1, 2) for stream in (0, 7, 123456) for it in (0, 1, 4095)],
                    dtype=np.float64)
    got = S.debug_math(7, rows, 7)
    for r, g in zip(rows, got):
        exp = oracle_c.ac_sample(7, np.arange(int(r[2]), dtype=np.int32), seed, int(r[3]), int(r[4]), int(r[5]))
        assert list(g.astype(np.int64)) == list(exp)


def make_scene(seed, **kw):
    args = dict(n_views=40, desc_per_view=400, views_per_place=10, landmarks_per_place=300, obs_per_view=140)
    args.update(kw)
    m = synth.make_map(seed, **args)
    return m


def dev_map(m, **params):
    p = S.default_params(ransac_round=25, **params)
    return S.Map(m.view_id, m.view_off, m.desc, params=p, view_wh=m.view_wh, kpt_xy=m.kpt_xy,
                 row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X,
                 intrinsic=m.intrinsic)


def compare_stages(m, q, dm, view_sel=None, **okw):
    exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=view_sel, **okw)
    dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
    dm.match_putative(dq, view_sel)
    cnt, mi, mj, md = dm.putative_read()
    np.testing.assert_array_equal(cnt, exp["put_count"])
    dm.geometric_filter(dq)
    gc, gi = dm.geometric_read()
    np.testing.assert_array_equal(gc, exp["geo_count"], err_msg="F-matrix filter: inlier counts per view")
    np.testing.assert_array_equal(gi, exp["geo_idx"], err_msg="F-matrix filter: inlier lists")
    dm.match_set(dq)
    qf, lm, p2, p3 = dm.match_set_read()
    np.testing.assert_array_equal(qf, exp["ms_qfeat"])
    np.testing.assert_array_equal(lm, exp["ms_landmark"])
    np.testing.assert_array_equal(bits(p2), bits(exp["pt2d"]))
    np.testing.assert_array_equal(bits(p3), bits(exp["pt3d"]))
    dm.resection(dq)
    pose, pq, pl, ii = dm.pose_read()
    assert bool(pose.ok) == exp["ok"]
    assert pose.n_matches_2d3d == len(exp["ms_qfeat"])
    if "p3p" in exp:
        assert pose.iterations == exp["p3p"]["iters"]
        assert pose.n_inliers == exp["n_inliers"]
    if exp["ok"]:
        np.testing.assert_array_equal(ii, exp["inlier_idx"])             # inlier set, AC-RANSAC order: bit-exact
        np.testing.assert_array_equal(pq, exp["pair_qfeat"])
        np.testing.assert_array_equal(pl, exp["pair_landmark"])
        R = np.array(pose.R).reshape(3, 3)
        c = np.array(pose.center)
        assert np.abs(R - exp["R"]).max() <= POSE_TOL and np.abs(c - exp["center"]).max() <= POSE_TOL
        np.testing.assert_array_equal(bits(np.array(pose.P)), bits(exp["P"].ravel()))   # in fact equal
        np.testing.assert_array_equal(bits(R), bits(exp["R"]))
        np.testing.assert_array_equal(bits(c), bits(exp["center"]))
        np.testing.assert_array_equal(bits(np.array(pose.K)), bits(exp["K"].ravel()))
        assert pose.nfa == exp["p3p"]["nfa"] and pose.error_max == exp["p3p"]["errmax"]
    dq.close()
    return exp, pose


def test_pipeline_stages_match_oracle_and_truth(oracle_c):
    m = make_scene(21)
    with dev_map(m) as dm:
        n_ok = 0
        for k in range(6):
            q = synth.make_query(m, 100 + k, n_feat=600, n_copies=200, outlier_frac=0.25)
            exp, pose = compare_stages(m, q, dm)
            if exp["ok"]:
                n_ok += 1
                # planted truth: the query's real pose
                R = np.array(pose.R).reshape(3, 3)
                assert np.abs(R - q.R_true).max() < 2e-2
                assert np.abs(np.array(pose.center) - q.C_true).max() < 0.25
                # inliers are true copies at their true position
                lm_slot_of_id = {int(i): s for s, i in enumerate(m.landmark_id)}
                _, pq, pl, _ = dm.pose_read()
                good = sum(1 for a, b in zip(pq, pl) if q.is_inlier[a] and q.landmark[a] == lm_slot_of_id[int(b)])
                assert good >= 0.95 * len(pq)
        assert n_ok >= 5


def test_pipeline_hard_cases(oracle_c):
    m = make_scene(22, ragged=True)
    with dev_map(m) as dm:
        # no copies at all: nothing survives the >=16 filter -> not localised, no error
        q = synth.make_query(m, 1, n_feat=300, n_copies=0)
        exp, pose = compare_stages(m, q, dm)
        assert not exp["ok"] and pose.n_matches_2d3d == 0
        # heavy outliers: F-filter mostly fails at 25 rounds
        q = synth.make_query(m, 2, n_feat=500, n_copies=200, outlier_frac=0.6)
        compare_stages(m, q, dm)
        # view shortlist
        sel = np.nonzero(m.view_place == q.place)[0].astype(np.uint32)[::2]
        compare_stages(m, q, dm, view_sel=sel)
        # tiny query
        q = synth.make_query(m, 3, n_feat=40, n_copies=30, outlier_frac=0.1)
        compare_stages(m, q, dm)


def test_pipeline_more_ransac_rounds(oracle_c):
    m = make_scene(23)
    with dev_map(m) as dm:
        pass
    p = dict(ransac_round=200)
    mm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(**p), view_wh=m.view_wh, kpt_xy=m.kpt_xy,
               row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X,
               intrinsic=m.intrinsic)
    q = synth.make_query(m, 5, n_feat=700, n_copies=250, outlier_frac=0.45)
    compare_stages(m, q, mm, ransac_round=200)
    mm.close()


def _map_with(m, **p):
    return S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(**p), view_wh=m.view_wh, kpt_xy=m.kpt_xy,
                 row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X,
                 intrinsic=m.intrinsic)


@pytest.mark.parametrize("rounds", [5, 9, 10, 33, 70, 400])
def test_fmatrix_filter_round_budgets(oracle_c, rounds):
    """The speculative (batched) uniform phase and the sequential tail must replay ACRANSAC's budget rules exactly:
    no reserve (< 10 rounds), end-of-uniform switch without a meaningful model, several batches, long tails."""
    m = make_scene(31)
    with _map_with(m, ransac_round=rounds) as dm:
        for seed, frac in ((7, 0.2), (8, 0.55), (9, 0.75)):
            q = synth.make_query(m, seed, n_feat=600, n_copies=220, outlier_frac=frac)
            compare_stages(m, q, dm, ransac_round=rounds)


def test_fmatrix_filter_large_views(oracle_c):
    """Views with more than 512 putative matches take the block-wide kernel; smaller ones the wave-parallel one."""
    m = make_scene(32, n_views=20, desc_per_view=1500, landmarks_per_place=1200, obs_per_view=1100)
    with _map_with(m, ransac_round=25) as dm:
        q = synth.make_query(m, 3, n_feat=1800, n_copies=1100, outlier_frac=0.3)
        exp, _ = compare_stages(m, q, dm, ransac_round=25)
        assert exp["put_count"].max() > 512
        q = synth.make_query(m, 4, n_feat=1800, n_copies=500, outlier_frac=0.3)
        exp, _ = compare_stages(m, q, dm, ransac_round=25)
        big = exp["put_count"][exp["put_count"] >= 16]
        assert len(big) and big.max() <= 512


def _distort_k3(xy, f, ppx, ppy, k1, k2, k3):
    """forward radial model of OpenMVG's pinhole_radial_k3: p_d = p_u (1 + k1 r^2 + k2 r^4 + k3 r^6)"""
    p = (np.asarray(xy, np.float64) - [ppx, ppy]) / f
    r2 = (p ** 2).sum(1, keepdims=True)
    return (p * (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) * f + [ppx, ppy]).astype(np.float32)


@pytest.mark.parametrize("disto", [(-0.12, 0.03, -0.002), (0.0, 0.0, 0.0), (0.25, -0.4, 0.1)])
def test_radial_k3_intrinsic(oracle_c, disto):
    """A10 with a pinhole_radial_k3 camera: pt2D = get_ud_pixel(query keypoint) (localization.cpp:484-487).  The
    query image is distorted with the forward model, so undistortion must bring the pose back to the truth; every
    stage is compared with the oracle bit for bit (zero coefficients still run remove_disto's bisection)."""
    m = make_scene(27)
    f, ppx, ppy = m.intrinsic[:3]
    m.intrinsic = (f, ppx, ppy) + tuple(disto)
    with dev_map(m) as dm:
        n_ok = 0
        for k in range(3):
            q = synth.make_query(m, 300 + k, n_feat=600, n_copies=220, outlier_frac=0.2)
            q.kpt_xy = _distort_k3(q.kpt_xy, f, ppx, ppy, *disto)
            exp, pose = compare_stages(m, q, dm)
            if exp["ok"]:
                n_ok += 1
                assert np.abs(np.array(pose.R).reshape(3, 3) - q.R_true).max() < 2e-2
                assert np.abs(np.array(pose.center) - q.C_true).max() < 0.25
            if len(exp["pt2d"]):
                und = oracle_c.ud_pixel_k3(q.kpt_xy[exp["ms_qfeat"]].astype(np.float64), f, ppx, ppy, *disto)
                np.testing.assert_array_equal(bits(exp["pt2d"]), bits(und))
                if any(disto):
                    assert np.abs(exp["pt2d"] - q.kpt_xy[exp["ms_qfeat"]]).max() > 0.05   # it really moved points
        assert n_ok >= 2


def test_localize_one_call(oracle_c):
    m = make_scene(24)
    with dev_map(m) as dm:
        for k in range(3):
            q = synth.make_query(m, 300 + k, n_feat=800, n_copies=220, outlier_frac=0.3)
            exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height))
            dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            pose, pq, pl = dm.localize(dq)
            assert bool(pose.ok) == exp["ok"]
            if exp["ok"]:
                np.testing.assert_array_equal(pq, exp["pair_qfeat"])
                np.testing.assert_array_equal(pl, exp["pair_landmark"])
                np.testing.assert_array_equal(bits(np.array(pose.center)), bits(exp["center"]))
            dq.close()


def test_concurrent_contexts_equal_sequential(oracle_c):
    """sfmloc_localize_begin/_end on several contexts and sfmloc_localize_batch give exactly what one query
    at a time gives (the RNG is keyed by query-independent counters, workspaces are per context)."""
    m = make_scene(25)
    with dev_map(m) as dm:
        qs = [synth.make_query(m, 400 + k, n_feat=700, n_copies=200, outlier_frac=0.3) for k in range(7)]
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in qs]
        seq = [dm.localize(dq) for dq in dqs]
        ctxs = [dm.context() for _ in range(3)]
        got = [None] * len(dqs)
        for i, dq in enumerate(dqs):
            c = ctxs[i % 3]
            if i >= 3:
                got[i - 3] = c.end()
            c.begin(dq)
        for i in range(len(dqs) - 3, len(dqs)):
            got[i] = ctxs[i % 3].end()
        poses, pq, pl = dm.localize_batch(dqs, n_contexts=4, cap=1024)
        for i, (a, b) in enumerate(zip(seq, got)):
            assert a[0].ok == b[0].ok == poses[i].ok
            assert a[0].n_inliers == b[0].n_inliers == poses[i].n_inliers
            assert a[0].n_putative_views == b[0].n_putative_views and a[0].n_geometric_views == b[0].n_geometric_views
            np.testing.assert_array_equal(a[1], b[1])
            np.testing.assert_array_equal(a[2], b[2])
            np.testing.assert_array_equal(bits(np.array(a[0].P)), bits(np.array(b[0].P)))
            np.testing.assert_array_equal(bits(np.array(a[0].P)), bits(np.array(poses[i].P)))
            if a[0].ok:
                np.testing.assert_array_equal(a[1], pq[i, :a[0].n_inliers])
                np.testing.assert_array_equal(a[2], pl[i, :a[0].n_inliers])
        assert sum(p.ok for p in poses) >= 5
        assert seq[0][0].n_putative_views >= seq[0][0].n_geometric_views > 0
        # misuse: begin twice on one context
        ctxs[0].begin(dqs[0])
        with pytest.raises(S.SfmlocError):
            ctxs[0].begin(dqs[1])
        ctxs[0].end()
        with pytest.raises(S.SfmlocError):
            ctxs[0].end()
        for c in ctxs:
            c.close()
        for dq in dqs:
            dq.close()


def test_no_size_limits_near_duplicate_of_a_large_frame(oracle_c):
    """The reference has no limit on the matches of a view (MatchUtils.cpp:346-355) or on the 2D-3D correspondences
    (localization.cpp:479-509).  A 5 200-feature query that nearly duplicates 5 000-row map frames: several thousand
    putative matches per view (beyond the 2 048 the LDS form of K3 sorts -> k_fmatrix_large) and more than 4 096
    correspondences (beyond the LDS form of K5 -> global sort segments, regrown P3P workspace, pair lists outside the
    result record).  Every stage equal to the oracle, as for small inputs; whole-path and concurrent-context forms too."""
    m = synth.make_map(71, n_views=5, desc_per_view=5000, views_per_place=5, landmarks_per_place=6000, obs_per_view=4900,
                       map_flips=8)
    p3p_it = 400                                   # the oracle's P3P sorts n residuals per model: keep the test short
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=p3p_it),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    q = synth.make_query(m, 710, n_feat=5200, n_copies=4900, outlier_frac=0.05, query_flips=10)
    exp, pose = compare_stages(m, q, dm, p3p_max_iteration=p3p_it)
    assert exp["put_count"].max() > 2048, exp["put_count"]          # the K3 case
    assert len(exp["ms_qfeat"]) > 4096, len(exp["ms_qfeat"])        # the K5 case
    assert exp["ok"] and exp["n_inliers"] > 4096                    # ... and more inliers than the result record holds
    # the whole path in one call, and through a context (the workspace regrows once per context)
    dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
    p1, pq1, pl1 = dm.localize(dq)
    ctx = dm.context()
    ctx.begin(dq)
    p2, pq2, pl2 = ctx.end()
    for p, pq, pl in ((p1, pq1, pl1), (p2, pq2, pl2)):
        assert p.ok and p.n_inliers == exp["n_inliers"]
        np.testing.assert_array_equal(pq, exp["pair_qfeat"])
        np.testing.assert_array_equal(pl, exp["pair_landmark"])
        np.testing.assert_array_equal(bits(np.array(p.P)), bits(exp["P"].ravel()))
    # a small query afterwards on the same (regrown) context: unchanged behaviour
    qs = synth.make_query(m, 711, n_feat=600, n_copies=250)
    es = opipe.localize(m, qs.desc, qs.kpt_xy, (qs.width, qs.height), p3p_max_iteration=p3p_it)
    dqs = dm.query(qs.desc, qs.kpt_xy, qs.width, qs.height)
    ctx.begin(dqs)
    p3, pq3, pl3 = ctx.end()
    assert bool(p3.ok) == es["ok"]
    if es["ok"]:
        np.testing.assert_array_equal(pq3, es["pair_qfeat"])
        np.testing.assert_array_equal(bits(np.array(p3.P)), bits(es["P"].ravel()))
    ctx.close()
    dq.close()
    dqs.close()
    dm.close()


def test_wide_p3p_rounds_above_16384_features(oracle_c):
    """a query with more than 16 384 features (P3P workspace cap 32 768) and more than 4 096 correspondences,
    run after a query with more than 512 correspondences on the same map -- the map's recent queries then ask for WIDE
    rounds (four workgroups per hypothesis), and beyond the LDS forms a wide launch must keep its inlier lists in the plain
    slots (slot = hypothesis): with slot = 4 x hypothesis the lists of hypotheses >= 32 lay past the allocation.  Every
    stage equal to the oracle."""
    m = synth.make_map(73, n_views=2, desc_per_view=17000, views_per_place=2, landmarks_per_place=18000, obs_per_view=16500,
                       map_flips=8)
    p3p_it = 300
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=p3p_it),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    q0 = synth.make_query(m, 730, n_feat=2500, n_copies=1500, outlier_frac=0.1, query_flips=10)
    exp0, _ = compare_stages(m, q0, dm, p3p_max_iteration=p3p_it)
    assert 512 < len(exp0["ms_qfeat"]) <= 4096, len(exp0["ms_qfeat"])   # (wide rounds from now on)
    q = synth.make_query(m, 731, n_feat=17000, n_copies=9000, outlier_frac=0.05, query_flips=10)
    assert q.desc.shape[0] > 16384
    exp, pose = compare_stages(m, q, dm, p3p_max_iteration=p3p_it)
    assert len(exp["ms_qfeat"]) > 4096, len(exp["ms_qfeat"])
    assert exp["ok"] and pose.ok
    dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)   # ... and the whole path in one call
    p1, pq1, pl1 = dm.localize(dq)
    assert p1.ok and p1.n_inliers == exp["n_inliers"]
    np.testing.assert_array_equal(pq1, exp["pair_qfeat"])
    np.testing.assert_array_equal(pl1, exp["pair_landmark"])
    np.testing.assert_array_equal(bits(np.array(p1.P)), bits(exp["P"].ravel()))
    dq.close()
    dm.close()


def test_failed_regrowth_of_the_p3p_workspace_leaves_the_context_usable(monkeypatch):
    """ctx_p3p_reserve allocates the larger set before it lets go of the old one: when an allocation fails (injected:
    sfmloc_debug_fail_p3p_alloc(index of the allocation that fails), one shot) the query gets SFMLOC_ENOMEM, the context keeps its
    arrays and capacity, an ordinary query on it gives its usual result, the large query succeeds once memory is there,
    and the map's memory account grows only then."""
    m = synth.make_map(72, n_views=5, desc_per_view=1500, views_per_place=5, landmarks_per_place=1800, obs_per_view=1400)
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=200),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    small = synth.make_query(m, 721, n_feat=700, n_copies=300)
    big = synth.make_query(m, 722, n_feat=4500, n_copies=1200)      # > kP3pMaxN = 4 096 features: the workspace regrows
    dqs = dm.query(small.desc, small.kpt_xy, small.width, small.height)
    dqb = dm.query(big.desc, big.kpt_xy, big.width, big.height)
    ctx = dm.context()
    ctx.begin(dqs)
    ref = ctx.end()
    assert ref[0].ok
    bytes_before = dm.info()["hbm_bytes"]
    for k in (0, 5, 11):
        capi.debug_fail_p3p_alloc(k)
        with pytest.raises(S.SfmlocError) as ei:
            ctx.begin(dqb)
        assert ei.value.code == capi.ENOMEM, ei.value
        assert dm.info()["hbm_bytes"] == bytes_before
        ctx.begin(dqs)
        got = ctx.end()
        assert capi.result_fingerprint(*got) == capi.result_fingerprint(*ref)
    ctx.begin(dqb)
    pb = ctx.end()
    assert pb[0].ok
    assert dm.info()["hbm_bytes"] > bytes_before
    ctx.begin(dqs)
    assert capi.result_fingerprint(*ctx.end()) == capi.result_fingerprint(*ref)
    ctx.close()
    dqs.close()
    dqb.close()
    dm.close()


def test_between_the_register_sort_and_the_default_workspace(oracle_c):
    """1 024 < correspondences <= 4 096: beyond what K5's one-wave-per-model register sort holds, inside the default
    workspace -- the block-wide LDS sort of k_p3p_round (acransac.hip, the `else` of the fast path)."""
    m = synth.make_map(73, n_views=4, desc_per_view=2600, views_per_place=4, landmarks_per_place=3200, obs_per_view=2500,
                       map_flips=8)
    p3p_it = 400
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=p3p_it),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    q = synth.make_query(m, 730, n_feat=2700, n_copies=2400, outlier_frac=0.05, query_flips=10)
    exp, pose = compare_stages(m, q, dm, p3p_max_iteration=p3p_it)
    assert 1024 < len(exp["ms_qfeat"]) <= 4096, len(exp["ms_qfeat"])
    assert exp["ok"]
    dm.close()


def test_guided_matching_in_the_query_path(oracle_c):
    """params.guided_matching (-gm, localization.cpp:82,183,451 / LocalizeEngine.cc:459): every view that passes the
    F-matrix filter gets OpenMVG's guided matches under its estimated F; the 2D-3D set then keeps a guided match only when
    its query feature has a putative distance for that view (featDist, SfMDataUtils.cpp:105-106).  Stage lists, the
    2D-3D set, inliers and pose against the oracle, bit for bit; a radial intrinsic and the sharded path included."""
    for seed, radial in ((61, False), (62, True)):
        m = make_scene(seed)
        if radial:
            m.intrinsic = tuple(m.intrinsic[:3]) + (0.05, -0.02, 0.001)
        dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, guided_matching=1),
                   view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
                   landmark_X=m.landmark_X, intrinsic=m.intrinsic)
        n_ok = n_changed = 0
        for k in range(4):
            q = synth.make_query(m, 640 + k, n_feat=700, n_copies=230, outlier_frac=0.3, place=k % 4)
            exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), guided=True)
            plain = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height))
            dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            dm.match_putative(dq)
            dm.geometric_filter(dq)
            gc, gi, gj = dm.geometric_read_pairs()
            np.testing.assert_array_equal(gc, exp["geo_count"])
            for v in np.nonzero(gc)[0]:
                o0, n = int(m.view_off[v]), int(gc[v])
                np.testing.assert_array_equal(gi[o0:o0 + n], exp["geo_idx"][o0:o0 + n])
                np.testing.assert_array_equal(gj[o0:o0 + n], exp["geo_j"][o0:o0 + n])
            with pytest.raises(S.SfmlocError):
                dm.geometric_read()                     # guided matches are not indices into the putative lists
            dm.match_set(dq)
            qf, lm, p2, p3 = dm.match_set_read()
            np.testing.assert_array_equal(qf, exp["ms_qfeat"])
            np.testing.assert_array_equal(lm, exp["ms_landmark"])
            pose, pq, pl = dm.localize(dq)
            assert bool(pose.ok) == exp["ok"]
            if exp["ok"]:
                n_ok += 1
                np.testing.assert_array_equal(pq, exp["pair_qfeat"])
                np.testing.assert_array_equal(pl, exp["pair_landmark"])
                np.testing.assert_array_equal(bits(np.array(pose.P)), bits(exp["P"].ravel()))
            n_changed += int(not np.array_equal(exp["geo_count"], plain["geo_count"]))
            dq.close()
        assert n_ok >= 3 and n_changed >= 1          # guided matching does change the lists
        dm.close()


def test_sharded_map_equals_unsharded(oracle_c):
    """Two shards of one map on one GPU: parts exported by sfmloc_shard_begin/_export, concatenated as the
    all-gather would, merged by sfmloc_merge_begin -> exactly the unsharded sfmloc_localize result; the exported
    candidates are the oracle's."""
    import torch
    from sfmlocalization_amd import dist as D
    m = make_scene(26, ragged=True)
    cap = 4096
    full = dev_map(m)
    ranges = D.shard_views(m.view_off, 2)
    shards = []
    for v0, v1 in ranges:
        r0, r1 = int(m.view_off[v0]), int(m.view_off[v1])
        shards.append(S.Map(m.view_id[v0:v1], m.view_off[v0:v1 + 1] - m.view_off[v0], m.desc[r0:r1],
                            params=S.default_params(ransac_round=25), view_wh=m.view_wh[v0:v1],
                            kpt_xy=m.kpt_xy[r0:r1], row_landmark=m.row_landmark[r0:r1], landmark_id=m.landmark_id,
                            landmark_X=m.landmark_X, intrinsic=m.intrinsic))
    pb = D.part_bytes(cap)
    for k in range(4):
        q = synth.make_query(m, 800 + k, n_feat=600, n_copies=200, outlier_frac=0.3, place=k % 4)
        fq = full.query(q.desc, q.kpt_xy, q.width, q.height)
        ref = full.localize(fq)
        parts = torch.zeros((2, pb), dtype=torch.uint8, device="cuda")
        qs = []
        for s, sm in enumerate(shards):
            sq = sm.query(q.desc, q.kpt_xy, q.width, q.height)
            qs.append(sq)
            c = sm.context()
            c.shard_begin(sq)
            c.shard_export(parts.data_ptr() + s * pb, cap)
            c.sync()
            c.close()
            v0, v1 = ranges[s]
            # a shard exports its winners only: per query feature the candidate with the smallest order key
            exp_c = D.reduce_candidates(opipe.shard_candidates(m, q.desc, q.kpt_xy, (q.width, q.height), v0, v1))
            got_c = D.unpack_part(parts[s].cpu().numpy(), cap)
            assert len(np.unique(got_c["qfeat"])) == len(got_c)
            got_c, exp_c = np.sort(got_c, order="order"), np.sort(exp_c, order="order")   # arrival order is free
            assert len(got_c) == len(exp_c)
            for f in ("order", "qfeat", "landmark_id", "X"):
                np.testing.assert_array_equal(got_c[f], exp_c[f], err_msg=f)
        c = shards[1].context()                                               # any rank can own the merge
        c.merge_begin(qs[1], parts.data_ptr(), 2, cap)
        pose, pq, pl = c.end()
        c.close()
        assert pose.ok == ref[0].ok and pose.n_inliers == ref[0].n_inliers
        np.testing.assert_array_equal(pq, ref[1])
        np.testing.assert_array_equal(pl, ref[2])
        np.testing.assert_array_equal(bits(np.array(pose.P)), bits(np.array(ref[0].P)))
        np.testing.assert_array_equal(bits(np.array(pose.center)), bits(np.array(ref[0].center)))
        for sq in qs:
            sq.close()
        fq.close()
    # the torch.distributed layer on one rank (world 1) drives the same entry points
    comp = D.HipShardCompute(full, cap, n_contexts=2)
    qobjs = [synth.make_query(m, 800 + k, n_feat=600, n_copies=200, outlier_frac=0.3, place=k % 4) for k in range(3)]
    dqs = [full.query(q.desc, q.kpt_xy, q.width, q.height) for q in qobjs]
    loc = D.ShardedLocalizer(comp, cap, rank=0, world=1)
    res = loc.localize_batch(dqs)
    for i, dq in enumerate(dqs):
        ref = full.localize(dq)
        assert res[i]["ok"] == bool(ref[0].ok)
        np.testing.assert_array_equal(res[i]["pair_qfeat"], ref[1])
        np.testing.assert_array_equal(bits(res[i]["P"].ravel()), bits(np.array(ref[0].P)))
    # the two-slot pipeline (stage 1 of batch b+1 queued before batch b's exchange and P3P stage): same results
    batches = [[dqs[0], dqs[1]], [dqs[2]], [dqs[1], dqs[0], dqs[2]]]
    outs = list(loc.localize_stream(batches))
    assert len(outs) == 3
    for b, out in zip(batches, outs):
        for i, dq in enumerate(b):
            k = dqs.index(dq)
            assert out[i]["ok"] == res[k]["ok"]
            np.testing.assert_array_equal(out[i]["pair_qfeat"], res[k]["pair_qfeat"])
            np.testing.assert_array_equal(bits(out[i]["P"].ravel()), bits(res[k]["P"].ravel()))
    comp.close()
    for dq in dqs:
        dq.close()
    for sm in shards:
        sm.close()
    full.close()


def test_pose_refinement_extension(oracle_c):
    """refine_pose = 1 (A13, north-star extension with no reference counterpart -> parity unpinned): the kernel's
    LM (normal equations on f64 MFMA) against the oracle's plain-loop LM within 1e-7, cost never worse, inlier set
    untouched; refine_pose = 0 stays the reference-equivalent output."""
    m = make_scene(27)
    base = dev_map(m)
    ref = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, refine_pose=1),
                view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
                landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    f, ppx, ppy = m.intrinsic
    err0, err1 = [], []
    for k in range(5):
        q = synth.make_query(m, 900 + k, n_feat=600, n_copies=220, outlier_frac=0.25)
        exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height))
        q0 = base.query(q.desc, q.kpt_xy, q.width, q.height)
        q1 = ref.query(q.desc, q.kpt_xy, q.width, q.height)
        p0, pq0, pl0 = base.localize(q0)
        p1, pq1, pl1 = ref.localize(q1)
        assert bool(p0.ok) == bool(p1.ok) == exp["ok"]
        if not exp["ok"]:
            continue
        np.testing.assert_array_equal(pq0, pq1)
        np.testing.assert_array_equal(pl0, pl1)
        np.testing.assert_array_equal(bits(np.array(p0.P)), bits(exp["P"].ravel()))       # unrefined = oracle, bit for bit
        o = oracle_c.refine_pose(exp["pt2d"], exp["pt3d"], exp["inlier_idx"], f, ppx, ppy, exp["R"], exp["t"])
        R1, c1 = np.array(p1.R).reshape(3, 3), np.array(p1.center)
        assert np.abs(R1 - o["R"]).max() < 1e-7 and np.abs(c1 - o["center"]).max() < 1e-7
        assert 1 <= p1.reserved <= 20      # LM iterations (the last accept/reject steps are decided by rounding noise,
                                           # fused on the MFMA path, so the count is not compared)
        np.testing.assert_allclose(np.array(p1.K).reshape(3, 3), [[f, 0, ppx], [0, f, ppy], [0, 0, 1]])
        P1 = np.array(p1.P).reshape(3, 4)
        np.testing.assert_allclose(P1, np.array(p1.K).reshape(3, 3) @ np.c_[R1, np.array(p1.t)], rtol=1e-12, atol=1e-9)
        assert o["cost"] <= o["cost0"]
        err0.append(np.abs(np.array(p0.center) - q.C_true).max())
        err1.append(np.abs(c1 - q.C_true).max())
        q0.close()
        q1.close()
    assert len(err0) >= 4 and np.mean(err1) <= np.mean(err0) * 1.05      # refinement does not hurt, usually helps
    base.close()
    ref.close()


def test_shared_gpu_round_sizes_do_not_change_the_result():
    """When other contexts have work queued, K5 evaluates fewer speculative hypotheses per round (acransac.hip
    p3p_next_batch_limit); the acceptance rule is replayed exactly for any partition into rounds, so every stage must
    still equal the oracle bit for bit.  The policy is chosen when the library first runs, hence the child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"SFMLOC_P3P_ADAPTIVE": "1"},
                  {"SFMLOC_P3P_ADAPTIVE": "1", "SFMLOC_P3P_ADAPT_QUARTERS": "4", "SFMLOC_P3P_ADAPT_FLOOR": "16",
                   "SFMLOC_P3P_ROUNDS": "3"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_parity.py"), "12", "77000"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "every stage bit-exact" in r.stdout



def test_small_p3p_rounds_prepared_hypotheses_and_the_fallback_to_the_full_form(oracle_c):
    """Match sets of at most 512 correspondences take K5's small launch form (k_p3p_round_small: 152 VGPRs instead of
    248, 11 KB of LDS instead of 50) when the query has no more features than that or the map's last 8 queries were that
    small; the form is chosen before the set's size is known, a small round on a larger set leaves at once and the
    query's rounds are queued again in the full form (acransac.hip kP3pSmallN, capi.hip ctx_resection_wait).  Ten small
    queries (the prediction builds up), a large one (refuted), small ones again, the large one again: every result is
    the oracle's, bit for bit.  Then whole campaigns in child processes with the forms forced (SFMLOC_P3P_SMALL = 2: every
    query starts small, 0: never) and with the coming round's hypotheses solved by the replaying workgroup, one per lane,
    always or never (SFMLOC_P3P_PREP_AHEAD = 2 / 0; by default only while the GPU is shared)."""
    import os
    import subprocess
    import sys
    m = synth.make_map(74, n_views=5, desc_per_view=1600, views_per_place=5, landmarks_per_place=1900, obs_per_view=1500,
                       map_flips=8)
    p3p_it = 300
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=p3p_it),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    ctx = dm.context()
    plan = [("small", 7400 + k) for k in range(10)] + [("large", 7420), ("small", 7421), ("small", 7422), ("large", 7423)]
    sizes = []
    for kind, seed in plan:
        q = (synth.make_query(m, seed, n_feat=600, n_copies=260) if kind == "small" else
             synth.make_query(m, seed, n_feat=1700, n_copies=1300, outlier_frac=0.1, query_flips=10))
        exp = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), p3p_max_iteration=p3p_it)
        sizes.append(len(exp["ms_qfeat"]))
        dq = dm.query(q.desc, q.kpt_xy, q.width, q.height)
        ctx.begin(dq)
        p, pq, pl = ctx.end()
        dq.close()
        assert bool(p.ok) == exp["ok"], (kind, seed)
        if exp["ok"]:
            np.testing.assert_array_equal(pq, exp["pair_qfeat"])
            np.testing.assert_array_equal(pl, exp["pair_landmark"])
            np.testing.assert_array_equal(bits(np.array(p.P)), bits(exp["P"].ravel()))
    assert max(sizes[:10]) <= 512 and sizes[10] > 512 and sizes[13] > 512, sizes
    ctx.close()
    dm.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"SFMLOC_P3P_SMALL": "2", "SFMLOC_P3P_PREP_AHEAD": "2"}, {"SFMLOC_P3P_SMALL": "0", "SFMLOC_P3P_PREP_AHEAD": "2"},
                  {"SFMLOC_P3P_SMALL": "2", "SFMLOC_P3P_PREP_AHEAD": "0"}):
        env = dict(os.environ, **extra)
        for tool, args, want in (("fuzz_p3p_large.py", ["8", "95000"], "bit-exact"), ("fuzz_parity.py", ["10", "78000"], "every stage bit-exact")):
            r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", tool)] + args, env=env,
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (extra, r.stdout[-2000:] + r.stderr[-2000:])
            assert want in r.stdout


def test_sequential_p3p_form_equals_the_round_form():
    """k_p3p_seq (acransac.hip): while the GPU is shared a query's whole P3P AC-RANSAC runs in ONE workgroup of 4 / 8 / 16
    waves -- the iterations in order, a model per wave up to 256 correspondences, W sorted runs of 256 merged through LDS
    beyond -- instead of rounds of speculative hypotheses.  Forced for every query (SFMLOC_P3P_SEQ = 2) in child
    processes: the localisation campaign (small sets, radial intrinsics, guided matching, the shortlist chain) and the
    large-set campaign (600 ... 5 000 correspondences: every W, and sets the launch does not hold -- more than 256 x
    waves, or more than 4 096 -- which come back untouched and take the round form) give the oracle's result bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for waves in ("8", "16", "4"):
        env = dict(os.environ, SFMLOC_P3P_SEQ="2", SFMLOC_P3P_SEQ_WAVES=waves)
        for tool, args, want in (("fuzz_p3p_large.py", ["8", "96000"], "bit-exact"), ("fuzz_parity.py", ["12", "76000"], "every stage bit-exact")):
            r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", tool)] + args, env=env,
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (waves, tool, r.stdout[-2000:] + r.stderr[-2000:])
            assert want in r.stdout


def test_k3_waves_per_view_do_not_change_the_result():
    """k_fmatrix_fast<W>: 16 waves per view for a query alone on the GPU, 4 while the GPU is shared (a 16-wave workgroup
    is a compute unit's whole register file); 8 exists for comparison runs.  Same arithmetic, same replay: every stage
    equals the oracle whichever W serves either situation (the choice is read when the library first runs: child
    processes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for w in ("4", "8"):
        env = dict(os.environ, SFMLOC_K3_WAVES_ALONE=w, SFMLOC_K3_WAVES_SHARED=w)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_parity.py"), "16", "79000"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (w, r.stdout[-2000:] + r.stderr[-2000:])
        assert "every stage bit-exact" in r.stdout


def test_k3_wide_form_and_plain_form_give_the_same_results():
    """k_fmatrix_fast<4, 1024, wide>: a query alone on the GPU with a short view list takes gridDim.y workgroups per view,
    one per iteration of the first uniform batch; the last to arrive replays.  The suite's lone queries take it by default;
    here whole campaigns with the form forced for every query (SFMLOC_K3_WIDE=2: shared GPU, gang sessions, long lists up
    to 256 views) and switched off (0: the plain forms, as before round 4), small and large match sets, in child processes
    (the choice is read when the library first runs)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for wide in ("2", "0"):
        env = dict(os.environ, SFMLOC_K3_WIDE=wide, SFMLOC_K3_WIDE_2048=wide)
        for tool, args in (("fuzz_parity.py", ["24", "83000"]), ("fuzz_p3p_large.py", ["6", "84000"]),
                           ("fuzz_gang.py", ["40", "85000"])):
            r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", tool)] + args, env=env,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, (wide, tool, r.stdout[-2000:] + r.stderr[-2000:])
            assert r.stdout.strip().splitlines()[-1].startswith("OK"), (wide, tool, r.stdout[-500:])


def test_k3_wide_form_for_1025_to_2048_matches_per_view(oracle_c):
    """A query alone whose views have 1 025 ... 2 048 putative matches: the first one finds them in k_fmatrix_filter (the
    block-wide LDS sort), the following ones -- Map::k3_huge_credit -- in k_fmatrix_fast<4, 2048, wide> (32 residuals per
    lane in the register sort).  Every stage equals the oracle either way; a smaller query afterwards too."""
    m = synth.make_map(77, n_views=3, desc_per_view=2100, views_per_place=3, landmarks_per_place=2500, obs_per_view=2000,
                       map_flips=8)
    p3p_it = 200
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=p3p_it),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    seen = []
    for k in range(3):
        q = synth.make_query(m, 770 + k, n_feat=2300, n_copies=2000, outlier_frac=0.1, query_flips=10)
        exp, pose = compare_stages(m, q, dm, p3p_max_iteration=p3p_it)
        seen.append(int(exp["put_count"].max()))
        assert exp["ok"]
    assert all(1024 < c <= 2048 for c in seen), seen
    q = synth.make_query(m, 779, n_feat=700, n_copies=300, outlier_frac=0.2)
    compare_stages(m, q, dm, p3p_max_iteration=p3p_it)
    dm.close()


def test_k3_register_form_for_513_to_1024_matches_per_view(oracle_c):
    """A query that nearly duplicates map frames of ~1 000 features has views with 513 ... 1 024 putative matches: beyond
    k_fmatrix_fast<W, 512>, which leaves them to the block-wide LDS form -- unless the map's recent queries had such views
    (Map::k3_big_credit), in which case k_fmatrix_fast<W, 1024> takes them (16 residuals per lane in the register sort).
    First query: the LDS form; the following ones: the register form.  Every stage equals the oracle either way, and a
    whole large-set campaign with the form forced (SFMLOC_K3_BIG=2) and switched off (0) in child processes."""
    import os
    import subprocess
    import sys
    m = synth.make_map(75, n_views=4, desc_per_view=1100, views_per_place=4, landmarks_per_place=1300, obs_per_view=1000,
                       map_flips=8)
    p3p_it = 300
    dm = S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25, p3p_max_iteration=p3p_it),
               view_wh=m.view_wh, kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id,
               landmark_X=m.landmark_X, intrinsic=m.intrinsic)
    seen = []
    for k in range(4):
        q = synth.make_query(m, 750 + k, n_feat=1200, n_copies=1000, outlier_frac=0.1, query_flips=10)
        exp, pose = compare_stages(m, q, dm, p3p_max_iteration=p3p_it)
        seen.append(int(exp["put_count"].max()))
        assert exp["ok"]
    assert all(512 < c <= 1024 for c in seen), seen
    dm.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for form in ("2", "0"):
        env = dict(os.environ, SFMLOC_K3_BIG=form)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_p3p_large.py"), "10", "98000"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (form, r.stdout[-2000:] + r.stderr[-2000:])
        assert "bit-exact" in r.stdout
