"""Independent second implementation (NumPy / SciPy, float64) of AKAZE's non-linear scale space and detector response --
TEST INFRASTRUCTURE ONLY.

The device code (akaze.hip) and the C oracle (sfm_oracle_akaze.c) were written together and agree bit for bit; OpenCV is
not in this image, so neither is pinned.  This twin restates the same published algorithm (Alcantarilla et al., "Fast
Explicit Diffusion for Accelerated Features in Nonlinear Scale Spaces", and OpenCV 3's AKAZEFeatures.cpp) with library
routines and in double precision, so that a coding slip in the float32 restatement -- a tap order, a border rule, a wrong
FED cycle -- shows up as a difference far above rounding (tests/test_oracle_twins.py compares at a tolerance):

  fed_steps          the FED cycle for a stopping time T from its closed form, in natural order (within a level the
                     conductivity is fixed, so the steps are powers of one linear operator and commute; the oracle applies
                     them in the "kappa" order that keeps float32 stable)
  gaussian / scharr  scipy.ndimage.correlate1d / correlate with mode "nearest" (BORDER_REPLICATE) / "mirror"
                     (BORDER_REFLECT_101)
  halfsample         a 2 x 2 block mean (INTER_AREA at an exact factor of two)
  contrast factor    70th percentile of the gradient magnitude's 300-bin histogram (numpy.histogram)
  scale_space        Lt of every level;  hessian_response  the determinant of the scaled second derivatives
  orientation, mldb  Compute_Main_Orientation and the full 3-channel M-LDB descriptor of a given keypoint, vectorised
"""
import numpy as np
from scipy import ndimage


def levels(w, h, omax=4, nsub=4):
    out = []
    for i in range(omax):
        lw, lh = int(w / 2 ** i), int(h / 2 ** i)
        if (lw < 80 or lh < 40) and i != 0:
            break
        for j in range(nsub):
            esigma = 1.6 * 2.0 ** (j / nsub + i)
            out.append(dict(w=lw, h=lh, octave=i, sublevel=j, esigma=esigma, etime=0.5 * esigma * esigma,
                            sigma_size=int(esigma * 1.5 / 2 ** i + 0.5)))
    return out


def fed_steps(T, tau_max=0.25):
    n = int(np.ceil(np.sqrt(3.0 * T / tau_max + 0.25) - 0.5 - 1e-8) + 0.5)
    if n <= 0:
        return np.zeros(0)
    scale = 3.0 * T / (tau_max * n * (n + 1))
    k = np.arange(n)
    return scale * tau_max / (2.0 * np.cos(np.pi * (2 * k + 1) / (4 * n + 2)) ** 2)


def gauss_kernel(ksize, sigma):
    x = np.arange(ksize) - (ksize - 1) / 2.0
    k = np.exp(-0.5 * x * x / (sigma * sigma))
    return k / k.sum()


def gaussian(img, ksize, sigma):
    k = gauss_kernel(ksize, sigma)
    return ndimage.correlate1d(ndimage.correlate1d(img, k, axis=1, mode="nearest"), k, axis=0, mode="nearest")


def scharr(img, xorder, scale=1, ws=3.0, wm=10.0):
    """dx (xorder = 1) or dy of the Scharr pair spread to `scale` pixels: weights ws, wm, ws across, -1 0 +1 along."""
    n = 2 * scale + 1
    k = np.zeros((n, n))
    for a, wgt in ((0, ws), (scale, wm), (n - 1, ws)):
        if xorder:
            k[a, 0], k[a, n - 1] = -wgt, wgt
        else:
            k[0, a], k[n - 1, a] = -wgt, wgt
    return ndimage.correlate(img, k, mode="mirror")


def contrast_factor(img):
    s = gaussian(img, 5, 1.0)
    m = np.hypot(scharr(s, 1), scharr(s, 0))[1:-1, 1:-1].ravel()
    hmax = m.max()
    m = m[m != 0]
    nbin = np.minimum(np.floor(300.0 * (m / hmax)).astype(int), 299)
    hist = np.bincount(nbin, minlength=300)
    nth = int(len(m) * 0.7)
    c = np.cumsum(hist)
    if c[-1] < nth:
        return 0.03
    k = int(np.searchsorted(c, nth, side="left")) + 1 if nth > 0 else 0
    return hmax * k / 300.0


def diffuse(L, g, tau):
    """one explicit step L += tau/2 * div((g + g_neighbour) grad L), zero flux across the border"""
    out = L.copy()
    fx = (g[:, :-1] + g[:, 1:]) * (L[:, 1:] - L[:, :-1])
    fy = (g[:-1, :] + g[1:, :]) * (L[1:, :] - L[:-1, :])
    out[:, :-1] += 0.5 * tau * fx
    out[:, 1:] -= 0.5 * tau * fx
    out[:-1, :] += 0.5 * tau * fy
    out[1:, :] -= 0.5 * tau * fy
    return out


def scale_space(gray, omax=4, nsub=4):
    """-> (levels, [Lt], [Lsmooth]) in float64"""
    img = np.asarray(gray, np.float64) / 255.0
    lv = levels(img.shape[1], img.shape[0], omax, nsub)
    Lt = [gaussian(img, 9, 1.6)]
    Ls = [Lt[0].copy()]
    kc = contrast_factor(img)
    for i in range(1, len(lv)):
        L, Lp = lv[i], lv[i - 1]
        cur = Lt[i - 1]
        if L["octave"] > Lp["octave"]:
            assert Lp["w"] == 2 * L["w"] and Lp["h"] == 2 * L["h"], "the twin halves exact factors of two only"
            cur = cur.reshape(L["h"], 2, L["w"], 2).mean(axis=(1, 3))
            kc *= 0.75
        sm = gaussian(cur, 5, 1.0)
        lx, ly = scharr(sm, 1), scharr(sm, 0)
        g = 1.0 / (1.0 + (lx * lx + ly * ly) / (kc * kc))
        for tau in fed_steps(L["etime"] - Lp["etime"]):
            cur = diffuse(cur, g, tau)
        Lt.append(cur)
        Ls.append(sm)
    return lv, Lt, Ls


def hessian_response(Lsmooth, sigma_size):
    s = sigma_size
    norm = 1.0 / (2.0 * s * (10.0 / 3.0 + 2.0))
    ws, wm = norm, (10.0 / 3.0) * norm
    lx, ly = scharr(Lsmooth, 1, s, ws, wm), scharr(Lsmooth, 0, s, ws, wm)
    lxx, lyy, lxy = scharr(lx, 1, s, ws, wm), scharr(ly, 0, s, ws, wm), scharr(lx, 0, s, ws, wm)
    s2 = float(s * s)
    return (lxx * s2) * (lyy * s2) - (lxy * s2) ** 2


def derivatives(Lsmooth, sigma_size):
    """the first derivatives the orientation and the descriptor sample (Scharr at the level's scale, times the scale)"""
    s = sigma_size
    norm = 1.0 / (2.0 * s * (10.0 / 3.0 + 2.0))
    ws, wm = norm, (10.0 / 3.0) * norm
    return scharr(Lsmooth, 1, s, ws, wm) * s, scharr(Lsmooth, 0, s, ws, wm) * s


def _fround(v):
    return np.trunc(np.asarray(v, np.float64) + 0.5).astype(int)      # (int)(f + 0.5f)


def _at(img, y, x):
    h, w = img.shape
    return img[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)]


def orientation(level, Lx, Ly, x, y, size):
    """Compute_Main_Orientation: Gaussian-weighted first derivatives on a disc of radius 6 s, the dominant direction of
    a pi/3 window sliding in steps of 0.15 rad -> angle in [0, 2 pi)"""
    ratio = float(1 << level["octave"])
    s = int(_fround(0.5 * size / ratio))
    xf, yf = x / ratio, y / ratio
    ii, jj = np.meshgrid(np.arange(-6, 7), np.arange(-6, 7), indexing="ij")
    keep = ii * ii + jj * jj < 36
    ii, jj = ii[keep], jj[keep]
    g = np.exp(-(ii * ii + jj * jj) / 12.5) / (2.0 * np.pi * 6.25)
    iy, ix = _fround(yf + jj * s), _fround(xf + ii * s)
    rx, ry = g * _at(Lx, iy, ix), g * _at(Ly, iy, ix)
    ang = np.mod(np.arctan2(ry, rx), 2 * np.pi)
    best, out = 0.0, 0.0
    a1 = 0.0
    while a1 < 2 * np.pi:
        a2 = a1 - 5 * np.pi / 3 if a1 + np.pi / 3 > 2 * np.pi else a1 + np.pi / 3
        if a1 < a2:
            m = (a1 < ang) & (ang < a2)
        else:
            m = ((ang > 0) & (ang < a2)) | ((ang > a1) & (ang < 2 * np.pi))
        sx, sy = rx[m].sum(), ry[m].sum()
        if sx * sx + sy * sy > best:
            best, out = sx * sx + sy * sy, np.mod(np.arctan2(sy, sx), 2 * np.pi)
        a1 += 0.15
    return out


def mldb(level, Lt, Lx, Ly, x, y, size, angle):
    """Get_MLDB_Full_Descriptor (3 channels, pattern 10): 486 bits as a 0/1 array -- mean intensity and rotated mean
    derivatives of the cells of a 2 x 2, a 3 x 3 and a 4 x 4 grid over the rotated, scaled patch, every pair compared"""
    ratio = float(1 << level["octave"])
    scale = int(_fround(0.5 * size / ratio))
    xf, yf = x / ratio, y / ratio
    co, si = np.cos(angle), np.sin(angle)
    bits = []
    for step in (10, 7, 5):
        vals = []
        for i in range(-10, 10, step):
            for j in range(-10, 10, step):
                kk, ll = np.meshgrid(np.arange(i, i + step), np.arange(j, j + step), indexing="ij")
                sy = yf + (ll * co * scale + kk * si * scale)
                sx = xf + (-ll * si * scale + kk * co * scale)
                y1, x1 = _fround(sy), _fround(sx)
                ri, rx, ry = _at(Lt, y1, x1), _at(Lx, y1, x1), _at(Ly, y1, x1)
                vals.append((ri.mean(), (-rx * si + ry * co).mean(), (rx * co + ry * si).mean()))
        vals = np.array(vals)
        for pos in range(3):
            v = vals[:, pos]
            for a in range(len(v)):
                bits.extend((v[a] > v[a + 1:]).astype(np.uint8))
    return np.array(bits, np.uint8)
