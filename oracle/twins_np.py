"""Independent second implementations (NumPy / SciPy) of the f64 stages of the oracle -- TEST INFRASTRUCTURE ONLY.

The reference's arithmetic for these stages lives in OpenMVG 1.1, which is neither under /root/reference nor in this
image, so the C oracle (sfm_oracle_geom.c) restates the published algorithms and is PARITY UNPINNED.  The device code
and the C oracle were written together; a shared misreading would pass every device-vs-oracle test.  The functions here
use different formulations and library routines (SVD null spaces, numpy.roots, scipy.linalg.rq, scipy.special.gammaln,
a distance-based P3P), agree with the oracle only to rounding, and are compared at a tolerance by
tests/test_oracle_twins.py:

  seven_point        7-point F via the SVD null space of the 7 x 9 system + numpy.roots on det(F1 + x F2)
                     (oracle: Gaussian elimination with complete pivoting + its own cubic solver)
  p3p_grunert        P3P by Grunert's distance formulation (quartic via numpy.roots) + Kabsch absolute orientation
                     (oracle: Kneip's rotation parametrisation + Ferrari)
  krt_from_p         KRt_From_P via scipy.linalg.rq (oracle: Givens rotations as OpenMVG does)
  logcombi           log10 C(n, k) via gammaln (oracle: running sums of a log10 table)
  best_nfa           the NFA minimum over the sorted residuals, vectorised
  guided_match       Geometry_guided_matching vectorised: F to pixels as N2^T F N1, all point-line distances by one matrix
                     product, Hamming distances from unpacked bits
  acransac           the AC-RANSAC loop in pure Python over the ORACLE's sample sequence (oracle_c.ac_sample is pinned by
                     the Random123 known answers), with the twin solvers and NFA: same acceptance decisions, same
                     inlier set
"""
import numpy as np

FLT_EPSILON = float(np.finfo(np.float32).eps)


# ----- two-view -----------------------------------------------------------------------------------------------------
def seven_point(x1, x2):
    """x1, x2: [7, 2] -> list of 3x3 F with x2^T F x1 = 0 (1 or 3 real solutions), each scaled to unit Frobenius norm."""
    x1 = np.asarray(x1, np.float64)
    x2 = np.asarray(x2, np.float64)
    A = np.zeros((7, 9))
    for i in range(7):
        a, b = np.append(x1[i], 1.0), np.append(x2[i], 1.0)
        A[i] = np.outer(b, a).ravel()                # x2^T F x1 = sum F[r, c] x2[r] x1[c]
    _, _, vt = np.linalg.svd(A)
    F1, F2 = vt[-1].reshape(3, 3), vt[-2].reshape(3, 3)
    # det(F1 + x F2) is a cubic in x: fit it through four points (exact for a cubic)
    xs = np.array([-1.0, 0.0, 1.0, 2.0])
    coef = np.polyfit(xs, [np.linalg.det(F1 + t * F2) for t in xs], 3)
    out = []
    for r in np.roots(coef):
        if abs(r.imag) < 1e-9 * max(1.0, abs(r.real)):
            F = F1 + r.real * F2
            out.append(F / np.linalg.norm(F))
    return out


def same_up_to_scale(A, B, tol):
    A = A / np.linalg.norm(A)
    B = B / np.linalg.norm(B)
    return min(np.abs(A - B).max(), np.abs(A + B).max()) < tol


def epipolar_error(F, x1, x2):
    """EpipolarDistanceError: squared distance of x2 to the line F x1."""
    x1h = np.c_[np.asarray(x1, np.float64), np.ones(len(x1))]
    x2h = np.c_[np.asarray(x2, np.float64), np.ones(len(x2))]
    l = x1h @ F.T
    return np.einsum("ij,ij->i", l, x2h) ** 2 / (l[:, 0] ** 2 + l[:, 1] ** 2)


def epipolar_error_ordered(F, x1, x2):
    """The same quantity with the additions in the oracle's order, so that residuals of rounding-noise size (the seven
    sample points) sort identically -- needed only where a test wants the very same AC-RANSAC path."""
    M = np.asarray(F, np.float64).ravel()
    x, y = np.asarray(x1, np.float64)[:, 0], np.asarray(x1, np.float64)[:, 1]
    u, v = np.asarray(x2, np.float64)[:, 0], np.asarray(x2, np.float64)[:, 1]
    l0 = (M[0] * x + M[1] * y) + M[2]
    l1 = (M[3] * x + M[4] * y) + M[5]
    l2 = (M[6] * x + M[7] * y) + M[8]
    num = (l0 * u + l1 * v) + l2
    return (num * num) / (l0 * l0 + l1 * l1)


# ----- resection ----------------------------------------------------------------------------------------------------
def p3p_grunert(x, X):
    """x: [3, 2] normalised image points (K^-1 applied), X: [3, 3] world points -> list of (R, t) with
    lambda (x, 1) = R X + t, lambda > 0.  Grunert 1841 as in Haralick et al. 1994: solve for the three distances, then
    the rigid motion between the camera-frame and world-frame triangles."""
    f = np.c_[np.asarray(x, np.float64), np.ones(3)]
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    X = np.asarray(X, np.float64)
    a, b, c = np.linalg.norm(X[1] - X[2]), np.linalg.norm(X[0] - X[2]), np.linalg.norm(X[0] - X[1])
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    a2, b2, c2 = a * a, b * b, c * c
    q1, q2 = (a2 - c2) / b2, (a2 + c2) / b2
    A4 = (q1 - 1) ** 2 - 4 * c2 / b2 * ca ** 2
    A3 = 4 * (q1 * (1 - q1) * cb - (1 - q2) * ca * cg + 2 * c2 / b2 * ca ** 2 * cb)
    A2 = 2 * (q1 ** 2 - 1 + 2 * q1 ** 2 * cb ** 2 + 2 * (b2 - c2) / b2 * ca ** 2 - 4 * q2 * ca * cb * cg
              + 2 * (b2 - a2) / b2 * cg ** 2)
    A1 = 4 * (-q1 * (1 + q1) * cb + 2 * a2 / b2 * cg ** 2 * cb - (1 - q2) * ca * cg)
    A0 = (1 + q1) ** 2 - 4 * a2 / b2 * cg ** 2
    sols = []
    for v in np.roots([A4, A3, A2, A1, A0]):
        if abs(v.imag) > 1e-7 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        den = 2 * (cg - v * ca)
        if abs(den) < 1e-14:
            continue
        u = ((-1 + q1) * v * v - 2 * q1 * cb * v + 1 + q1) / den
        if u <= 0:
            continue
        s1sq = c2 / (1 + u * u - 2 * u * cg)
        if s1sq <= 0:
            continue
        s1 = np.sqrt(s1sq)
        s = np.array([s1, u * s1, v * s1])
        P = f * s[:, None]                            # the three points in the camera frame
        # Kabsch: R, t with P = R X + t
        cp, cx = P.mean(0), X.mean(0)
        H = (X - cx).T @ (P - cp)
        U, _, Vt = np.linalg.svd(H)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
        R = Vt.T @ D @ U.T
        t = cp - R @ cx
        if np.abs(R @ X.T + t[:, None] - P.T).max() < 1e-7 * max(1.0, np.abs(P).max()):
            sols.append((R, t))
    return sols


def resection_error(P, X, x):
    Xh = np.c_[np.asarray(X, np.float64), np.ones(len(X))]
    p = Xh @ np.asarray(P, np.float64).reshape(3, 4).T
    return ((p[:, :2] / p[:, 2:3] - np.asarray(x, np.float64)) ** 2).sum(1)


def resection_error_ordered(P, X, x):
    M = np.asarray(P, np.float64).ravel()
    X = np.asarray(X, np.float64)
    x = np.asarray(x, np.float64)
    p0 = ((M[0] * X[:, 0] + M[1] * X[:, 1]) + M[2] * X[:, 2]) + M[3]
    p1 = ((M[4] * X[:, 0] + M[5] * X[:, 1]) + M[6] * X[:, 2]) + M[7]
    p2 = ((M[8] * X[:, 0] + M[9] * X[:, 1]) + M[10] * X[:, 2]) + M[11]
    dx, dy = p0 / p2 - x[:, 0], p1 / p2 - x[:, 1]
    return dx * dx + dy * dy


def krt_from_p(P):
    """P ~ K [R | t] (defined up to scale and sign) with K upper triangular, positive diagonal, K[2,2] = 1, det R = +1."""
    from scipy.linalg import rq
    P = np.asarray(P, np.float64).reshape(3, 4)
    K, R = rq(P[:, :3])
    S = np.diag(np.sign(np.diag(K)))
    K, R = K @ S, S @ R                       # K R unchanged, diag(K) > 0
    t = np.linalg.solve(K, P[:, 3])
    if np.linalg.det(R) < 0:                  # -P is the same camera
        R, t = -R, -t
    return K / K[2, 2], R, t


# ----- a contrario --------------------------------------------------------------------------------------------------
def logcombi(k, n):
    from scipy.special import gammaln
    k = np.asarray(k, np.float64)
    return (gammaln(n + 1.0) - gammaln(k + 1.0) - gammaln(n - k + 1.0)) / np.log(10.0)


def best_nfa(sorted_err, s, n_models, logalpha0, mult_error, max_threshold=np.inf):
    """min over k in (s, n] with e_k <= max_threshold of
       log10(n_models (n - s)) + (logalpha0 + mult * log10(e_k + eps)) (k - s) + log10 C(n, k) + log10 C(k, s)
    -> (nfa, k) or (inf, s)."""
    e = np.asarray(sorted_err, np.float64)
    n = len(e)
    k = np.arange(s + 1, n + 1)
    ek = e[k - 1]
    ok = ek <= max_threshold
    if not ok.any():
        return np.inf, s
    # the oracle's loop stops at the first residual above the threshold; the residuals ascend, so that is a prefix
    last = np.nonzero(~ok)[0]
    if len(last):
        k, ek = k[:last[0]], ek[:last[0]]
    if len(k) == 0:
        return np.inf, s
    logalpha = logalpha0 + mult_error * np.log10(ek + FLT_EPSILON)
    # OpenMVG keeps the two logcombi tables as float (std::vector<float>): the rounding to float32 decides which of two
    # nearly equal NFA values is the minimum, so it is part of the algorithm, not of the implementation
    lc_n = logcombi(k, n).astype(np.float32).astype(np.float64)
    lc_k = logcombi(np.full(len(k), s), k).astype(np.float32).astype(np.float64)
    nfa = np.log10(n_models * (n - s)) + logalpha * (k - s) + lc_n + lc_k
    i = int(np.argmin(nfa))
    return float(nfa[i]), int(k[i])


def acransac(n, s, n_models, fit, error, logalpha0, mult_error, n_iter, sample, max_threshold=np.inf):
    """ACRANSAC (Moisan, Moulon, Monasse, IPOL 2012; OpenMVG robust_estimator_ACRansac.hpp) in plain Python.
    fit(sample) -> models; error(model) -> [n] residuals; sample(vec_index or None, n_index, it) -> indices.
    -> (inliers in ascending-residual order, model, error_max, min_nfa, iterations run)"""
    if n <= s:
        return [], None, np.inf, np.inf, 0
    vec_index = None
    n_index = n
    min_nfa, err_max, best_model, inl = np.inf, np.inf, None, []
    n_reserve = n_iter // 10
    n_iter -= n_reserve
    it = 0
    while it < n_iter:
        smp = sample(vec_index, n_index, it)
        better = False
        for M in fit(smp):
            e = np.asarray(error(M), np.float64)
            e = np.where(np.isnan(e), np.inf, e)
            order = np.lexsort((np.arange(n), e))
            nfa, k = best_nfa(e[order], s, n_models, logalpha0, mult_error, max_threshold)
            if nfa < min_nfa:
                better, min_nfa, err_max, best_model = True, nfa, e[order][k - 1], M
                inl = order[:k].tolist()
        if (better and min_nfa < 0) or (it + 1 == n_iter and n_reserve):
            if not inl:
                n_iter += 1
                n_reserve -= 1
            else:
                vec_index, n_index = list(inl), len(inl)
                if n_reserve:
                    n_iter = it + 1 + n_reserve
                    n_reserve = 0
        it += 1
    if min_nfa >= 0:
        inl = []
    return inl, best_model, err_max, min_nfa, it


# ----- guided matching ----------------------------------------------------------------------------------------------
def guided_match(F_norm, errmax_norm, wh1, wh2, xy1, desc1, xy2, desc2, dist_ratio=0.6):
    """Geometry_guided_matching (descriptor variant) for one pair, vectorised and formulated independently of the C
    oracle: F is taken to pixels as N2^T F N1 with the normalising similarities written out as matrices, the point-line
    distances of ALL feature pairs come from one matrix product, the Hamming distances from numpy.unpackbits, the best /
    second-best by a partial sort.  -> (i[], j[]) ascending in i."""
    def norm_mat(w, h):
        s = 1.0 / np.sqrt(float(w) * float(h))
        return np.array([[s, 0, -0.5 * w * s], [0, s, -0.5 * h * s], [0, 0, 1.0]])
    N1, N2 = norm_mat(*wh1), norm_mat(*wh2)
    F = N2.T @ np.asarray(F_norm, np.float64).reshape(3, 3) @ N1
    thr = (np.sqrt(errmax_norm) * np.sqrt(float(wh2[0]) * float(wh2[1]))) ** 2
    x1 = np.c_[np.asarray(xy1, np.float64).reshape(-1, 2), np.ones(len(xy1))]
    x2 = np.c_[np.asarray(xy2, np.float64).reshape(-1, 2), np.ones(len(xy2))]
    lines = x1 @ F.T                                   # epipolar line of every feature of image 1 in image 2
    num = lines @ x2.T                                 # [n1, n2]
    e = num * num / (lines[:, 0] ** 2 + lines[:, 1] ** 2)[:, None]
    b1 = np.unpackbits(np.asarray(desc1, np.uint8).reshape(-1, 64), axis=1).astype(np.int32)
    b2 = np.unpackbits(np.asarray(desc2, np.uint8).reshape(-1, 64), axis=1).astype(np.int32)
    ham = b1.sum(1)[:, None] + b2.sum(1)[None, :] - 2 * (b1 @ b2.T)
    d2 = np.where(e < thr, (ham * ham).astype(np.float64), np.inf)
    oi, oj = [], []
    for i in range(d2.shape[0]):
        row = d2[i]
        if np.isfinite(row).sum() < 2:
            continue
        j = int(np.argmin(row))                        # first minimum = the lowest j on ties, as the sequential scan
        second = np.partition(row, 1)[1]
        if row[j] < dist_ratio * dist_ratio * second:
            oi.append(i)
            oj.append(j)
    return np.array(oi, np.uint32), np.array(oj, np.uint32)
