"""CPU: the geometry half of the oracle (oracle/sfm_oracle_geom.c) against known answers.

The reference holds no vectors for these stages and OpenMVG is not in the image (parity unpinned); what can
be pinned is (a) published known-answer vectors (Philox), (b) analytic truth (planted geometry), (c) NumPy
cross-checks of the numerics."""
import math

import numpy as np
import pytest

import synthdata as synth


def test_det_log10_matches_libm(oracle_c):
    rng = np.random.Generator(np.random.PCG64(1))
    xs = np.concatenate([10.0 ** rng.uniform(-300, 300, 2000), rng.uniform(0.5, 2.0, 2000),
                         np.arange(1, 3000, dtype=np.float64), [1.0, 2.0, 10.0, 1e-7, 1.19e-7, 5e-324, 1e308]])
    for x in xs:
        got = oracle_c.det_log10(float(x))
        exp = math.log10(float(x))
        assert abs(got - exp) <= 4e-16 * max(1.0, abs(exp)), (x, got, exp)
    assert oracle_c.det_log10(1.0) == 0.0
    assert oracle_c.det_log10(0.0) == -math.inf
    assert oracle_c.det_log10(math.inf) == math.inf
    assert math.isnan(oracle_c.det_log10(-1.0))


def test_philox_known_answers(oracle_c):
    # Random123 kat_vectors for philox4x32-10
    assert oracle_c.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle_c.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6,
                                                                           0x6d5451fd]
    assert oracle_c.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                  [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_sample_is_sorted_distinct_and_mapped(oracle_c):
    idx = np.arange(100, 130, dtype=np.int32)
    seen = set()
    for it in range(200):
        s = oracle_c.ac_sample(7, idx, 1234, 1, 5, it)
        assert (np.diff(s) > 0).all() and s.min() >= 100 and s.max() < 130
        seen.update(int(v) for v in s)
    assert len(seen) == 30
    # different streams / stages / iterations decorrelate
    a = oracle_c.ac_sample(3, idx, 1, 2, 0, 0)
    assert not (a == oracle_c.ac_sample(3, idx, 1, 2, 0, 1)).all() or not (a == oracle_c.ac_sample(3, idx, 1, 2, 1, 0)).all()
    # n == X: the only possible sample
    assert list(oracle_c.ac_sample(3, np.array([4, 9, 11], np.int32), 7, 2, 0, 3)) == [4, 9, 11]


def test_cubic_and_quartic_solvers(oracle_c):
    rng = np.random.Generator(np.random.PCG64(2))
    for _ in range(300):
        c = rng.normal(size=4) * 10.0 ** rng.uniform(-2, 2, 4)
        got = oracle_c.solve_cubic(*c)
        exp = sorted(r.real for r in np.roots(c) if abs(r.imag) < 1e-9 * max(1, abs(r)))
        if len(got) == len(exp):
            np.testing.assert_allclose(got, exp, rtol=1e-7, atol=1e-9)
        for r in got:  # every reported root is a root
            scale = sum(abs(ci) * abs(r) ** (3 - i) for i, ci in enumerate(c))
            assert abs(np.polyval(c, r)) <= 1e-9 * scale + 1e-300
    for _ in range(300):
        a = rng.normal(size=5) * 10.0 ** rng.uniform(-1, 1, 5)
        got = sorted(oracle_c.solve_quartic_real(a))
        exp = sorted(r.real for r in np.roots(a))
        np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-6)
    # biquadratic branch (q == 0)
    got = sorted(oracle_c.solve_quartic_real([1.0, 0.0, -5.0, 0.0, 4.0]))
    np.testing.assert_allclose(got, [-2, -1, 1, 2], atol=1e-12)
    got = sorted(oracle_c.solve_quartic_real([1.0, 0.0, 2.0, 0.0, 5.0]))
    np.testing.assert_allclose(got, sorted(r.real for r in np.roots([1, 0, 2, 0, 5])), atol=1e-12)


def _two_views(rng, n):
    X = rng.uniform(-2, 2, size=(n, 3)) + np.array([0, 0, 8.0])
    R1, C1 = np.eye(3), np.zeros(3)
    ang = 0.2
    R2 = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    C2 = np.array([1.0, 0.2, 0.1])
    f, ppx, ppy = 800.0, 320.0, 240.0
    x1, _ = synth.project(R1, C1, X, f, ppx, ppy)
    x2, _ = synth.project(R2, C2, X, f, ppx, ppy)
    return X, x1, x2, (R2, C2, f, ppx, ppy)


def test_seven_point_contains_true_fundamental(oracle_c):
    rng = np.random.Generator(np.random.PCG64(3))
    for _ in range(20):
        _, x1, x2, _ = _two_views(rng, 12)
        s = 1.0 / math.sqrt(640 * 480)
        a = x1 * s - np.array([320, 240]) * s
        b = x2 * s - np.array([320, 240]) * s
        Fs = oracle_c.seven_point(a[:7], b[:7])
        assert 1 <= len(Fs) <= 3
        best = np.inf
        for F in Fs:
            # exact on the 7 sample points, rank 2
            r7 = [abs(np.r_[b[i], 1] @ F @ np.r_[a[i], 1]) for i in range(7)]
            assert max(r7) < 1e-10 * np.abs(F).max()
            assert abs(np.linalg.det(F)) < 1e-9 * np.abs(F).max() ** 3
            rest = max(abs(np.r_[b[i], 1] @ F @ np.r_[a[i], 1]) / np.abs(F).max() for i in range(7, 12))
            best = min(best, rest)
        assert best < 1e-8  # one of the solutions is the true epipolar geometry


def test_p3p_contains_true_pose(oracle_c):
    rng = np.random.Generator(np.random.PCG64(4))
    hits = 0
    for _ in range(50):
        X, _, x2, (R2, C2, f, ppx, ppy) = _two_views(rng, 3)
        xn = (x2 - np.array([ppx, ppy])) / f
        Ms = oracle_c.p3p_kneip(xn, X)
        assert len(Ms) == 4
        t2 = -R2 @ C2
        err = min(np.abs(M[:, :3] - R2).max() + np.abs(M[:, 3] - t2).max() for M in Ms if np.isfinite(M).all())
        hits += err < 1e-6
    assert hits >= 48
    # collinear world points -> no model
    assert len(oracle_c.p3p_kneip([[0, 0], [0.1, 0], [0.2, 0]], [[0, 0, 5], [1, 0, 5], [2, 0, 5]])) == 0


def test_logcombi_tables(oracle_c):
    a, b = oracle_c.logcombi_tables(7, 60)
    for k in (0, 1, 7, 30, 59, 60):
        assert abs(a[k] - math.log10(math.comb(60, k))) < 1e-5
    for n in (0, 6, 7, 8, 40, 60):
        exp = math.log10(math.comb(n, 7)) if n > 7 else 0.0
        assert abs(b[n] - exp) < 1e-5


def test_krt_from_p(oracle_c):
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(20):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q = -Q
        K = np.array([[800.0, 0, 320], [0, 800, 240], [0, 0, 1]])
        t = rng.normal(size=3)
        P = K @ np.c_[Q, t]
        sgn = rng.choice([-1.0, 1.0])  # P is only defined up to sign/scale for KRt_From_P
        K2, R2, t2, c2 = oracle_c.krt_from_p(P * sgn * rng.uniform(0.5, 2))
        np.testing.assert_allclose(K2, K, rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(R2, Q, atol=1e-10)
        np.testing.assert_allclose(t2, t, atol=1e-9)
        np.testing.assert_allclose(c2, -Q.T @ t, atol=1e-9)


def test_fmatrix_acransac_recovers_planted_inliers(oracle_c):
    rng = np.random.Generator(np.random.PCG64(6))
    ok = 0
    for trial in range(10):
        _, x1, x2, _ = _two_views(rng, 120)
        x2 = x2 + rng.normal(0, 0.5, x2.shape)
        out = rng.choice(120, 24, replace=False)       # 20 % outliers
        x2[out] = rng.uniform([0, 0], [640, 480], size=(24, 2))
        truth = np.ones(120, bool)
        truth[out] = False
        r = oracle_c.fmatrix_filter(x1, (640, 480), x2, (640, 480), 4.0, 200, 99 + trial, stream=trial)
        if r["n"] == 0:
            continue
        assert r["nfa"] < 0 and r["n"] > 17
        inl = np.zeros(120, bool)
        inl[r["inliers"]] = True
        prec = (inl & truth).sum() / inl.sum()
        rec = (inl & truth).sum() / truth.sum()
        ok += (prec > 0.93 and rec > 0.85)
        assert r["iters"] <= 200
    assert ok >= 8
    # too few points: nData <= sizeSample -> nothing
    assert oracle_c.fmatrix_filter(x1[:7], (640, 480), x2[:7], (640, 480), 4.0, 25, 1, 0)["n"] == 0
    # pure noise: no meaningful model (NFA >= 0) or too few inliers
    r = oracle_c.fmatrix_filter(rng.uniform(0, 480, (40, 2)), (640, 480), rng.uniform(0, 480, (40, 2)), (640, 480),
                                4.0, 25, 3, 0)
    assert r["n"] == 0


def test_p3p_acransac_recovers_planted_pose(oracle_c):
    rng = np.random.Generator(np.random.PCG64(7))
    for trial in range(5):
        X, _, x2, (R2, C2, f, ppx, ppy) = _two_views(rng, 200)
        x2 = x2 + rng.normal(0, 1.0, x2.shape)
        out = rng.choice(200, 100, replace=False)      # 50 % outliers
        x2[out] = rng.uniform([0, 0], [640, 480], size=(100, 2))
        r = oracle_c.p3p_localize(x2, X, f, ppx, ppy, 4096, 1000 + trial)
        assert r["n"] >= 80 and r["nfa"] < 0
        assert r["iters"] < 4096                      # meaningful model found early -> budget cut to the reserve
        inl = np.zeros(200, bool)
        inl[r["inliers"]] = True
        assert (inl[out]).sum() <= 6
        K, R, t, c = oracle_c.krt_from_p(r["P"])
        np.testing.assert_allclose(K, [[f, 0, ppx], [0, f, ppy], [0, 0, 1]], atol=1e-6)
        assert np.abs(R - R2).max() < 5e-3
        assert np.abs(c - C2).max() < 5e-2
        assert 0.5 < r["errmax"] < 10.0               # pixels
    # same seed -> same answer; other seed -> may differ but still valid
    a = oracle_c.p3p_localize(x2, X, f, ppx, ppy, 4096, 5)
    b = oracle_c.p3p_localize(x2, X, f, ppx, ppy, 4096, 5)
    np.testing.assert_array_equal(a["inliers"], b["inliers"])
    np.testing.assert_array_equal(a["P"], b["P"])
    assert oracle_c.p3p_localize(x2[:3], X[:3], f, ppx, ppy, 4096, 5)["n"] == 0


def test_match_set_semantics(oracle_c):
    # two views; query feature 5 matched from both; view 0's putative list has two rows hitting j=5
    view_off = np.array([0, 4, 8], np.uint32)
    put_count = np.array([3, 2], np.uint32)
    put_i = np.array([0, 1, 3, 0, 0, 2, 0, 0], np.uint32)
    put_j = np.array([5, 7, 5, 0, 5, 9, 0, 0], np.uint32)
    put_d = np.array([30, 10, 12, 0, 12, 40, 0, 0], np.uint32)   # featDist[v0][5] = 12 (last wins)
    row_landmark = np.array([100, 101, -1, 103, 200, -1, 202, -1], np.int32)
    # geometric matches in std::map order: view 0 list then view 1 list
    gv = [0, 0, 0, 1, 1]
    gi = [0, 1, 3, 0, 2]
    gj = [5, 7, 5, 5, 9]
    q, lm = oracle_c.match_set(gv, gi, gj, view_off, put_count, put_i, put_j, put_d, row_landmark, 12)
    # j=5: candidates (v0,i0,lm100,d=12), (v0,i3,lm103,d=12) tie -> first kept, (v1,i0,lm200,d=12) tie -> first kept
    # j=7: lm101 ; j=9: (v1,i2) has landmark 202
    assert list(q) == [5, 7, 9]
    assert list(lm) == [100, 101, 202]
    # strictly smaller distance replaces
    put_d2 = put_d.copy()
    put_d2[4] = 11
    q, lm = oracle_c.match_set(gv, gi, gj, view_off, put_count, put_i, put_j, put_d2, row_landmark, 12)
    assert list(lm) == [200, 101, 202]


def test_refine_pose_extension(oracle_c):
    """A13 (north-star extension, absent from the reference): LM on the inliers lowers the reprojection cost and
    moves a perturbed pose back to the truth."""
    rng = np.random.Generator(np.random.PCG64(9))
    X, _, x2, (R2, C2, f, ppx, ppy) = _two_views(rng, 150)
    x2n = x2 + rng.normal(0, 0.5, x2.shape)
    ang = 0.01
    dR = np.array([[1, -ang, 0], [ang, 1, 0], [0, 0, 1.0]])
    U, _, Vt = np.linalg.svd(dR @ R2)
    R0 = U @ Vt
    t0 = -R0 @ (C2 + np.array([0.05, -0.03, 0.04]))
    r = oracle_c.refine_pose(x2n, X, np.arange(150), f, ppx, ppy, R0, t0)
    assert r["cost"] < 0.05 * r["cost0"] and 1 <= r["iters"] <= 20
    assert np.abs(r["R"] - R2).max() < 2e-3 and np.abs(r["center"] - C2).max() < 2e-2
    np.testing.assert_allclose(r["R"] @ r["R"].T, np.eye(3), atol=1e-12)
    # already optimal (noise-free): stays put
    r2 = oracle_c.refine_pose(x2, X, np.arange(150), f, ppx, ppy, R2, -R2 @ C2)
    assert np.abs(r2["R"] - R2).max() < 1e-9 and r2["cost"] <= r2["cost0"]


def test_ud_pixel_k3_inverts_the_radial_model():
    """Pinhole_Intrinsic_Radial_K3::get_ud_pixel restatement: undistort(distort(p)) = p to the bisection's 1e-8 (in
    squared normalised radius), identity at the principal point, and exactly reproducible."""
    from oracle import oracle_c
    oracle_c.build()
    rng = np.random.Generator(np.random.PCG64(4))
    f, ppx, ppy = 800.0, 320.0, 240.0
    for k1, k2, k3 in ((-0.12, 0.03, -0.002), (0.25, -0.4, 0.1), (0.0, 0.0, 0.0)):
        pu = np.stack([rng.uniform(0, 640, 300), rng.uniform(0, 480, 300)], 1)
        pn = (pu - [ppx, ppy]) / f
        r2 = (pn ** 2).sum(1, keepdims=True)
        pd = pn * (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) * f + [ppx, ppy]
        back = oracle_c.ud_pixel_k3(pd, f, ppx, ppy, k1, k2, k3)
        assert np.abs(back - pu).max() < 5e-3      # the bisection stops at 1e-8 ABSOLUTE in r^2: coarse near the centre
        assert np.array_equal(back, oracle_c.ud_pixel_k3(pd, f, ppx, ppy, k1, k2, k3))
    c = oracle_c.ud_pixel_k3(np.array([[ppx, ppy]]), f, ppx, ppy, -0.1, 0.01, 0.0)
    assert c.tolist() == [[ppx, ppy]]


def test_golden_geometry_scenes():
    """tests/golden/geometry_scenes.npz (make_golden.py): planted scenes with the analytic answer and the oracle's exact
    output at minting time.  Pins the C restatement against regressions (sampling, solvers, NFA, budget rules) and
    checks it against the truth it was planted with."""
    import os
    from oracle import oracle_c
    oracle_c.build()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry_scenes.npz"))
    f, ppx, ppy = g["intrinsic"]
    r = oracle_c.p3p_localize(g["pt2d"], g["pt3d"], f, ppx, ppy, 4096, 0x5f3759df12345678, stream=0)
    assert r["n"] == int(g["p3p_n"]) and r["iters"] == int(g["p3p_iters"])
    np.testing.assert_array_equal(r["inliers"], g["p3p_inliers"])
    np.testing.assert_array_equal(r["P"].view(np.uint64), g["p3p_P"].view(np.uint64))
    assert r["nfa"] == float(g["p3p_nfa"]) and r["errmax"] == float(g["p3p_errmax"])
    truth_in = ~g["is_outlier"]
    assert truth_in[r["inliers"]].mean() > 0.98 and r["n"] >= 0.95 * truth_in.sum()
    K, R, t, c = oracle_c.krt_from_p(r["P"])
    assert np.abs(c - g["C_true"]).max() < 0.05 and np.abs(R - g["R_true"]).max() < 5e-3
    w, h = (int(v) for v in g["wh"])
    fr = oracle_c.fmatrix_filter(g["f_x1"], (w, h), g["f_x2"], (w, h), 4.0, 200, 0x5f3759df12345678, stream=7)
    assert fr["n"] == int(g["f_n"]) and fr["iters"] == int(g["f_iters"]) and fr["nfa"] == float(g["f_nfa"])
    np.testing.assert_array_equal(fr["inliers"], g["f_inliers"])
    np.testing.assert_array_equal(fr["F"].view(np.uint64), g["f_F"].view(np.uint64))
    f_truth = ~g["f_is_outlier"]
    assert f_truth[fr["inliers"]].mean() > 0.95 and fr["n"] >= 0.9 * f_truth.sum()
