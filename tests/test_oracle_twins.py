"""CPU: the C oracle's f64 stages against independent NumPy / SciPy formulations (oracle/twins_np.py) at a tolerance.

The oracle is PARITY UNPINNED (OpenMVG is absent from the reference tree and the image), and the device code was written
against it, so a shared misreading of the published algorithms would pass every device-vs-oracle test.  These twins use
different formulations and library routines: SVD null space + numpy.roots for the 7-point solver, Grunert's distance
formulation + Kabsch for P3P, scipy.linalg.rq for KRt_From_P, gammaln for the logcombi tables, a vectorised NFA, and the
AC-RANSAC loop in plain Python replayed over the oracle's own (Random123-pinned) sample sequence."""
import numpy as np
import pytest

from oracle import twins_np as T

STAGE_FMATRIX, STAGE_P3P = 1, 2


def _scene(rng, n, noise=0.0):
    """n world points seen by two pinhole cameras; -> X, x1, x2 (pixels), (R2, t2, f, ppx, ppy)"""
    f, ppx, ppy = 800.0, 320.0, 240.0
    X = rng.uniform(-3, 3, (n, 3)) + np.array([0, 0, 10.0])

    def cam():
        w = rng.normal(0, 0.15, 3)
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        return R, rng.normal(0, 0.8, 3)

    def proj(R, t):
        p = X @ R.T + t
        return np.c_[f * p[:, 0] / p[:, 2] + ppx, f * p[:, 1] / p[:, 2] + ppy] + rng.normal(0, noise, (n, 2)) if noise \
            else np.c_[f * p[:, 0] / p[:, 2] + ppx, f * p[:, 1] / p[:, 2] + ppy]

    (R1, t1), (R2, t2) = cam(), cam()
    return X, proj(R1, t1), proj(R2, t2), (R2, t2, f, ppx, ppy)


def test_seven_point_same_solution_set(oracle_c):
    """Both solvers return the same set of fundamental matrices (up to scale) for random minimal samples."""
    rng = np.random.Generator(np.random.PCG64(101))
    n_cmp = 0
    for _ in range(60):
        _, x1, x2, _ = _scene(rng, 7)
        a = (x1 - [320, 240]) / 800.0          # normalised, as the kernel adaptor feeds the solver
        b = (x2 - [320, 240]) / 800.0
        Fo = oracle_c.seven_point(a, b)
        Ft = T.seven_point(a, b)
        assert len(Fo) in (1, 3) and len(Ft) in (1, 3)
        # every oracle solution satisfies the 7 constraints and has rank 2 ...
        for F in Fo:
            F = np.asarray(F).reshape(3, 3)
            assert np.abs(np.einsum("ij,jk,ik->i", np.c_[b, np.ones(7)], F, np.c_[a, np.ones(7)])).max() < 1e-9 * np.abs(F).max()
            assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-9
        # ... and is one of the twin's (a pair of nearly double roots may be resolved differently: compare when the
        # counts agree, which is the overwhelming majority)
        if len(Fo) == len(Ft):
            n_cmp += 1
            for F in Fo:
                assert any(T.same_up_to_scale(np.asarray(F).reshape(3, 3), G, 1e-6) for G in Ft)
    assert n_cmp >= 55


def test_p3p_same_pose_set(oracle_c):
    """Kneip (oracle) and Grunert + Kabsch (twin) find the same poses; both contain the true one."""
    rng = np.random.Generator(np.random.PCG64(102))
    n_match = n_total = 0
    for _ in range(60):
        X, _, x2, (R2, t2, f, ppx, ppy) = _scene(rng, 3)
        xn = (x2 - [ppx, ppy]) / f
        Mo = [np.asarray(M) for M in oracle_c.p3p_kneip(xn, X) if np.isfinite(np.asarray(M)).all()]
        Mt = T.p3p_grunert(xn, X)
        assert any(np.abs(R - R2).max() + np.abs(t - t2).max() < 1e-6 for R, t in Mt), "twin misses the true pose"
        assert any(np.abs(M[:, :3] - R2).max() + np.abs(M[:, 3] - t2).max() < 1e-6 for M in Mo), "oracle misses it"
        # every geometrically valid oracle model (a rotation that reprojects the three points in front of the camera)
        # is one of the twin's poses
        for M in Mo:
            R, t = M[:, :3], M[:, 3]
            p = X @ R.T + t
            valid = (np.abs(R @ R.T - np.eye(3)).max() < 1e-8 and (p[:, 2] > 0).all()
                     and np.abs(p[:, :2] / p[:, 2:3] - xn).max() < 1e-8)
            if valid:
                n_total += 1
                n_match += any(np.abs(R - Rt).max() + np.abs(t - tt).max() < 1e-5 for Rt, tt in Mt)
    assert n_total >= 60 and n_match >= n_total - 2, (n_match, n_total)


def test_krt_from_p_vs_rq(oracle_c):
    rng = np.random.Generator(np.random.PCG64(103))
    for _ in range(40):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(Q) < 0:
            Q = -Q
        K = np.array([[rng.uniform(400, 1500), rng.normal(0, 2), rng.uniform(200, 900)],
                      [0, rng.uniform(400, 1500), rng.uniform(200, 700)], [0, 0, 1.0]])
        P = K @ np.c_[Q, rng.normal(size=3)] * rng.choice([-1.0, 1.0]) * rng.uniform(0.3, 3)
        Ko, Ro, to, _ = oracle_c.krt_from_p(P)
        Kt, Rt, tt = T.krt_from_p(P)
        np.testing.assert_allclose(Ko, Kt, rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(Ro, Rt, atol=1e-10)
        np.testing.assert_allclose(to, tt, atol=1e-8)


def test_logcombi_and_nfa_vs_gammaln(oracle_c):
    rng = np.random.Generator(np.random.PCG64(104))
    for s, n in ((7, 60), (3, 300), (7, 2048), (3, 5000)):
        a, b = oracle_c.logcombi_tables(s, n)
        k = np.arange(0, n + 1)
        np.testing.assert_allclose(a, T.logcombi(k, n), atol=2e-3 * max(1, n / 2000), rtol=2e-6)
        m = np.arange(s + 1, n + 1)
        np.testing.assert_allclose(np.asarray(b)[m], T.logcombi(np.full(len(m), s), m), atol=1e-4, rtol=2e-6)
    # the NFA minimum over sorted residuals: the oracle's AC-RANSAC on a planted two-view scene reports the NFA of its
    # best model; the twin formula evaluated on that model's residuals gives the same value and the same inlier count
    X, x1, x2, _ = _scene(rng, 120, noise=0.4)
    x2[80:] = rng.uniform(0, 640, (40, 2))                 # outliers
    r = oracle_c.fmatrix_filter(x1, (640, 480), x2, (640, 480), 4.0, 200, 12345, 3)
    assert r["n"] > 60
    s1 = 1.0 / np.sqrt(640 * 480)
    N = np.array([[s1, 0, -320 * s1], [0, s1, -240 * s1], [0, 0, 1.0]])
    a = (np.c_[x1, np.ones(120)] @ N.T)[:, :2]
    b = (np.c_[x2, np.ones(120)] @ N.T)[:, :2]
    e = np.sort(T.epipolar_error(r["F"], a, b))
    logalpha0 = np.log10(2.0 * np.hypot(640, 480) / (640 * 480) / s1)
    nfa, k = T.best_nfa(e, 7, 3, logalpha0, 0.5, max_threshold=16.0 * s1 * s1)
    assert k == r["n"] and abs(nfa - r["nfa"]) < 1e-3 * abs(r["nfa"]) + 1e-3
    assert abs(e[k - 1] - r["errmax"]) < 1e-12 + 1e-9 * e[k - 1]


def _planted_two_view(rng, n, n_in):
    X, x1, x2, _ = _scene(rng, n, noise=0.5)
    x2[n_in:] = rng.uniform(0, 640, (n - n_in, 2))
    s1 = 1.0 / np.sqrt(640 * 480)
    N = np.array([[s1, 0, -320 * s1], [0, s1, -240 * s1], [0, 0, 1.0]])
    a = (np.c_[x1, np.ones(n)] @ N.T)[:, :2]
    b = (np.c_[x2, np.ones(n)] @ N.T)[:, :2]
    return x1, x2, a, b, s1


def test_acransac_fundamental_replayed_in_python(oracle_c):
    """GeometricFilter_FMatrix_AC for one pair, replayed by the plain-Python AC-RANSAC loop with the gammaln NFA over the
    oracle's sample sequence.
    (a) With the ORACLE's 7-point solutions plugged in, every acceptance decision, the iteration count, the inlier list
        (order included), NFA and errorMax come out identical: the loop, the budget rule and the NFA are the same
        algorithm in two implementations.
    (b) With the SVD solver the path is a different, equally valid one -- the seven sample points have residuals of
        rounding-noise size whose order is implementation defined, and that order decides which points later samples draw
        -- so the comparison is on the outcome: nearly the same inlier set, all of it planted inliers, an NFA as good."""
    rng = np.random.Generator(np.random.PCG64(105))
    seed, stream, n_iter = 0x5F3759DF12345678, 11, 120
    for trial in range(4):
        n, n_in = 90, 60
        x1, x2, a, b, s1 = _planted_two_view(rng, n, n_in)
        r = oracle_c.fmatrix_filter(x1, (640, 480), x2, (640, 480), 4.0, n_iter, seed, stream)

        def sample(vec_index, n_index, it):
            vi = np.arange(n_index, dtype=np.int32) if vec_index is None else np.asarray(vec_index, np.int32)
            return oracle_c.ac_sample(7, vi, seed, STAGE_FMATRIX, stream, it)

        logalpha0 = np.log10(2.0 * np.hypot(640, 480) / (640 * 480) / s1)
        fits = {"oracle": lambda smp: [np.asarray(F).reshape(3, 3) for F in
                                       oracle_c.seven_point(a[np.asarray(smp)], b[np.asarray(smp)])],
                "svd": lambda smp: T.seven_point(a[np.asarray(smp)], b[np.asarray(smp)])}
        for name, fit in fits.items():
            err = T.epipolar_error_ordered if name == "oracle" else T.epipolar_error
            inl, F, errmax, nfa, iters = T.acransac(n, 7, 3, fit, lambda F: err(F, a, b), logalpha0, 0.5,
                                                    n_iter, sample, max_threshold=16.0 * s1 * s1)
            if len(inl) <= 17:
                inl = []
            assert len(inl) >= 50 and r["n"] >= 50
            if name == "oracle":
                assert iters == r["iters"] and list(r["inliers"]) == list(inl)
                assert abs(nfa - r["nfa"]) < 1e-9 * abs(nfa) and abs(errmax - r["errmax"]) <= 1e-9 * errmax
            else:
                both = set(inl) & set(int(i) for i in r["inliers"])
                assert len(both) >= 0.8 * max(len(inl), r["n"])
                assert max(inl) < n_in + 3 and sum(i >= n_in for i in inl) <= 2      # planted inliers, not outliers
                assert nfa < -20.0 and r["nfa"] < -20.0       # both highly meaningful (two different random paths)


def test_acransac_resection_replayed_in_python(oracle_c):
    """SfM_Localizer::Localize (P3P AC-RANSAC): the same two comparisons with Kneip's solutions (identical path) and
    Grunert's (equivalent outcome)."""
    rng = np.random.Generator(np.random.PCG64(106))
    seed, n_iter = 0x5F3759DF12345678, 150
    for trial in range(3):
        n, n_in = 70, 45
        X, _, x2, (R2, t2, f, ppx, ppy) = _scene(rng, n, noise=0.6)
        x2[n_in:] = rng.uniform(0, 640, (n - n_in, 2))
        r = oracle_c.p3p_localize(x2, X, f, ppx, ppy, n_iter, seed)
        inv_f = 1.0 / f                                    # K^-1 applied as the kernel adaptor does: x / f + (-pp / f)
        xn = x2 * inv_f + np.array([-ppx * inv_f, -ppy * inv_f])

        def sample(vec_index, n_index, it):
            vi = np.arange(n_index, dtype=np.int32) if vec_index is None else np.asarray(vec_index, np.int32)
            return oracle_c.ac_sample(3, vi, seed, STAGE_P3P, 0, it)

        def fit_kneip(smp):
            smp = np.asarray(smp)
            return [np.asarray(M) for M in oracle_c.p3p_kneip(xn[smp], X[smp])]

        def fit_grunert(smp):
            smp = np.asarray(smp)
            return [np.c_[R, t] for R, t in T.p3p_grunert(xn[smp], X[smp])]

        K = np.array([[f, 0, ppx], [0, f, ppy], [0, 0, 1.0]])
        for name, fit in (("kneip", fit_kneip), ("grunert", fit_grunert)):
            err = T.resection_error_ordered if name == "kneip" else T.resection_error
            inl, M, errmax, nfa, iters = T.acransac(n, 3, 4, fit, lambda M: err(M, X, xn), np.log10(np.pi),
                                                    1.0, n_iter, sample)
            if len(inl) <= 7:
                inl = []
            assert len(inl) >= 38 and r["n"] >= 38
            if name == "kneip":
                assert iters == r["iters"] and list(r["inliers"]) == list(inl)
                assert abs(nfa - r["nfa"]) < 1e-9 * abs(nfa)
                assert abs(r["errmax"] - np.sqrt(errmax) / inv_f) < 1e-9 * r["errmax"]
                assert T.same_up_to_scale(r["P"], K @ M, 1e-12)
            else:
                both = set(inl) & set(int(i) for i in r["inliers"])
                assert len(both) >= 0.8 * max(len(inl), r["n"]) and sum(i >= n_in for i in inl) <= 2
                Rm, tm = M[:, :3], M[:, 3]
                assert np.abs(Rm - R2).max() < 5e-3 and np.abs(tm - t2).max() < 5e-2     # the planted pose


def test_akaze_scale_change_is_an_octave_shift(oracle_c):
    """AKAZE + M-LDB under a change of scale (the invariance a query taken closer to / farther from the scene relies on):
    a 2x box-downsampled image gives keypoints at half the position and half the size of the original's, found one
    octave lower, with descriptors that still match under the ratio test."""
    from oracle import oracle_np as onp
    import synthdata as synth
    g = synth.texture_image(4, 960, 1280)                        # h, w
    g2 = g.reshape(480, 2, 640, 2).astype(np.float32).mean(axis=(1, 3)).round().astype(np.uint8)
    kp, desc = oracle_c.akaze_detect_and_compute(g)
    kp2, desc2 = oracle_c.akaze_detect_and_compute(g2)
    assert len(kp) > 300 and len(kp2) > 150

    def pad64(d):
        out = np.zeros((len(d), 64), np.uint8)
        out[:, :61] = d
        return out

    j0, d0, j1, d1 = onp.hamming_2nn(pad64(desc), pad64(desc2))            # each small-image feature among the large one's
    ok = onp.ratio_accept(d0, d1, 0.8)
    assert ok.sum() > 0.25 * len(kp2), (int(ok.sum()), len(kp2))
    a, b = kp[j0[ok]], kp2[ok]
    # pixel centres: x_small = (x_large - 0.5) / 2
    err = np.abs((a[:, :2] - 0.5) / 2.0 - b[:, :2])
    assert np.median(err) < 1.0 and (err.max(axis=1) < 3.0).mean() > 0.8
    ratio = a[:, 2] / b[:, 2]
    assert abs(np.median(ratio) - 2.0) < 0.25
    # the class_id (evolution level) of the large image's feature sits about one octave (4 sublevels) above
    assert abs(np.median(a[:, 5] - b[:, 5]) - 4.0) <= 1.0


def test_akaze_scale_space_and_response_vs_scipy_twin(oracle_c):
    """The non-linear scale space (Gaussian start, contrast factor, Perona-Malik conductivity, FED cycles, 2x
    half-sampling) and the Hessian-determinant response of every evolution level against oracle/twins_akaze_np.py: SciPy
    correlations in float64 with the FED steps in natural order.  Agreement is at float32 rounding (1e-6); a wrong tap,
    border rule, FED cycle or derivative scale would be orders of magnitude above."""
    from oracle import twins_akaze_np as TA
    import synthdata as synth
    n_kp = 0
    for seed, (h, w) in ((3, (128, 160)), (5, (96, 192)), (8, (240, 320))):
        g = synth.texture_image(seed, h, w)
        kp, desc, ldet, lt = oracle_c.akaze_detect_and_compute(g, want_levels=True)
        lv, Lt, Ls = TA.scale_space(g)
        assert [(l["w"], l["h"]) for l in lv] == [tuple(x) for x in oracle_c.akaze_levels(w, h)]
        off = 0
        resp = []
        for i, l in enumerate(lv):
            n = l["w"] * l["h"]
            o = lt[off:off + n].reshape(l["h"], l["w"]).astype(np.float64)
            od = ldet[off:off + n].reshape(l["h"], l["w"]).astype(np.float64)
            d = TA.hessian_response(Ls[i], l["sigma_size"])
            resp.append(d)
            assert np.abs(o - Lt[i]).max() < 2e-6, (seed, i, np.abs(o - Lt[i]).max())
            assert np.abs(od - d).max() < 2e-5 * np.abs(od).max(), (seed, i)
            off += n
        # every keypoint the oracle reports sits on a local maximum of the twin's response at its level, above the
        # detector threshold, with the response value it reports
        n_kp += len(kp)
        for x, y, size, ang, response, cls in kp:
            l = lv[int(cls)]
            r = 2 ** l["octave"]
            cx, cy = int(round(x / r)), int(round(y / r))
            win = resp[int(cls)][max(cy - 2, 0):cy + 3, max(cx - 2, 0):cx + 3]
            assert win.max() > 0.001 * (1 - 1e-3)
            assert abs(win.max() - response) < 1e-4 * response + 1e-9, (x, y, cls, win.max(), response)
    assert n_kp >= 30, n_kp


def test_fed_cycle_reaches_the_stopping_time(oracle_c):
    """A FED cycle's steps sum to the diffusion time between two evolution levels (the property the schedule is built
    on), for the times AKAZE's default 4 x 4 levels ask for, and its largest step exceeds the explicit scheme's stability
    limit (which is the point of FED)."""
    from oracle import twins_akaze_np as TA
    lv = TA.levels(640, 480)
    for a, b in zip(lv, lv[1:]):
        T_ = b["etime"] - a["etime"]
        tau = TA.fed_steps(T_)
        assert abs(tau.sum() - T_) < 1e-9 * T_
        assert tau.max() > 0.25 or len(tau) <= 1


def test_akaze_orientation_and_mldb_vs_numpy_twin(oracle_c):
    """Main orientation and the 486-bit M-LDB descriptor of every keypoint the oracle reports, recomputed by the twin on
    ITS OWN float64 scale space and derivatives: angles to 1e-5 rad (the oracle's fixed-order float32 atan2 / sincos
    included), descriptor bits equal but for comparisons of near-equal cell means."""
    from oracle import twins_akaze_np as TA
    import synthdata as synth
    n_kp, n_bits, n_diff = 0, 0, 0
    for seed, (h, w) in ((8, (240, 320)), (11, (300, 400))):
        g = synth.texture_image(seed, h, w)
        kp, desc = oracle_c.akaze_detect_and_compute(g)
        lv, Lt, Ls = TA.scale_space(g)
        D = [TA.derivatives(Ls[i], l["sigma_size"]) for i, l in enumerate(lv)]
        for (x, y, size, ang, resp, cls), d in zip(kp, desc):
            c = int(cls)
            a = TA.orientation(lv[c], D[c][0], D[c][1], float(x), float(y), float(size))
            da = abs(a - float(ang))
            assert min(da, 2 * np.pi - da) < 1e-5, (x, y, c, a, ang)
            b = TA.mldb(lv[c], Lt[c], D[c][0], D[c][1], float(x), float(y), float(size), float(ang))
            ob = np.unpackbits(d, bitorder="little")[:486]
            n_diff += int((b != ob).sum())
            n_bits += 486
            n_kp += 1
    assert n_kp >= 80, n_kp
    assert n_diff <= 2e-4 * n_bits, (n_diff, n_bits)


def test_guided_matching_vs_numpy_twin(oracle_c):
    """Geometry_guided_matching (the -gm path): the C oracle against the vectorised twin (matrix-form unnormalisation of
    F, all point-line distances by one product, Hamming distances from unpacked bits).  The accepted (i, j) lists must be
    equal; a pair whose epipolar distance sits within rounding of the threshold may differ, so the scenes are checked
    for such pairs and there are none in these seeds."""
    import synthdata as synth
    rng = np.random.default_rng(5)
    total = 0
    for trial in range(6):
        n1, n2 = int(rng.integers(150, 400)), int(rng.integers(150, 400))
        wh1, wh2 = (640, 480), (int(rng.choice([640, 800])), int(rng.choice([480, 600])))
        X, a1, a2, _ = _scene(rng, 120)                       # exact projections: the 7-point F below is the true one
        # a fundamental matrix of the two views from eight exact correspondences (normalised frames), via the twin
        def nrm(x, wh):
            s = 1.0 / np.sqrt(wh[0] * wh[1])
            return (x - 0.5 * np.array(wh)) * s
        Fs = T.seven_point(nrm(a1[:7], wh1), nrm(a2[:7], wh2))
        # the solution with the smallest residual on the other points
        def resid(F):
            return np.abs(T.epipolar_error(F, nrm(a1, wh1), nrm(a2, wh2))).sum()
        F = min(Fs, key=resid)
        # features: the true correspondences (descriptors a few bits apart) plus clutter on both sides
        d_true = synth.random_descriptors(rng, 120)
        xy1 = np.r_[a1, rng.uniform(0, 1, (n1 - 120, 2)) * wh1].astype(np.float32)
        xy2 = np.r_[a2, rng.uniform(0, 1, (n2 - 120, 2)) * wh2].astype(np.float32)
        desc1 = np.r_[d_true, synth.random_descriptors(rng, n1 - 120)]
        desc2 = np.r_[synth.flip_bits(rng, d_true, 25), synth.random_descriptors(rng, n2 - 120)]
        errmax = float(rng.uniform(0.5, 4.0) ** 2 / (wh2[0] * wh2[1]))        # (pixels / sqrt(w h))^2: normalised frame
        ei, ej = oracle_c.guided_match(F, errmax, wh1, wh2, xy1, desc1, xy2, desc2, 0.6)
        ti, tj = T.guided_match(F, errmax, wh1, wh2, xy1, desc1, xy2, desc2, 0.6)
        assert np.array_equal(ei, ti) and np.array_equal(ej, tj), trial
        # (a feature needs two candidates inside its epipolar band for the ratio test, so only part of the true
        # correspondences can be accepted in a sparse scene; the accepted ones are the true ones)
        assert len(ei) >= 10, len(ei)
        assert (ei[ej < 120] == ej[ej < 120]).mean() > 0.95
        total += len(ei)
    assert total > 150, total
