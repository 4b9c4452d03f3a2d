"""TEST ORACLE (not product code): NumPy restatement of the server's per-user image undistortion
(VisionLocalizeServer/src/localizeImage.cc:149-177):

    newCameraMat = getOptimalNewCameraMatrix(K, dist, size, 1.0, size, &validRoi)
    undistort(image, undistortImage, K, dist, newCameraMat);  undistortImage = undistortImage(validRoi)

PARITY UNPINNED: OpenCV is not in this image and the reference holds no fixture for this step, so the functions below
restate OpenCV 3.0's published algorithms (imgproc/src/undistort.cpp `undistort` + `initUndistortRectifyMap` with
CV_16SC2 maps, imgproc/src/imgwarp.cpp `remap` INTER_LINEAR / BORDER_CONSTANT in 15-bit fixed point,
calib3d/src/calibration.cpp `cvGetOptimalNewCameraMatrix` + `icvGetRectangles`, imgproc/src/undistort.cpp
`cvUndistortPoints`) from their documentation and source as remembered; float/double stages follow those sources.
dist = (k1, k2, p1, p2[, k3[, k4, k5, k6]])."""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def _dist8(dist):
    d = np.zeros(8, np.float64)
    dist = np.asarray(dist, np.float64).ravel()
    d[:min(8, len(dist))] = dist[:8]
    return d


def undistort_points(pts, K, dist, P):
    """cvUndistortPoints on float32 points [n, 2] with R = I and new camera matrix P (3x3): five fixed-point
    iterations in double, result stored as float32."""
    k = _dist8(dist)
    K = np.asarray(K, np.float64).reshape(3, 3)
    P = np.asarray(P, np.float64).reshape(3, 3)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy
    out = np.zeros((len(pts), 2), np.float32)
    for i, (px, py) in enumerate(np.asarray(pts, np.float32)):
        x = (float(px) - cx) * ifx
        y = (float(py) - cy) * ify
        x0, y0 = x, y
        for _ in range(5):
            r2 = x * x + y * y
            icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
            dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
            x = (x0 - dx) * icdist
            y = (y0 - dy) * icdist
        xx = P[0, 0] * x + P[0, 1] * y + P[0, 2]
        yy = P[1, 0] * x + P[1, 1] * y + P[1, 2]
        ww = 1.0 / (P[2, 0] * x + P[2, 1] * y + P[2, 2])
        out[i] = (np.float32(xx * ww), np.float32(yy * ww))
    return out


def get_rectangles(K, dist, P, size):
    """icvGetRectangles: a 9x9 grid of image points undistorted into the new camera; inner = the largest rectangle
    inside all four warped edges, outer = the bounding box.  All comparisons in float32.  -> (inner, outer) as
    (x, y, w, h) float32 tuples."""
    w, h = size
    N = 9
    pts = np.zeros((N * N, 2), np.float32)
    for y in range(N):
        for x in range(N):
            pts[y * N + x] = (np.float32(x) * np.float32(w) / np.float32(N - 1),
                              np.float32(y) * np.float32(h) / np.float32(N - 1))
    u = undistort_points(pts, K, dist, P)
    big = np.float32(np.finfo(np.float32).max)
    iX0, iX1, iY0, iY1 = -big, big, -big, big
    oX0, oX1, oY0, oY1 = big, -big, big, -big
    for y in range(N):
        for x in range(N):
            p = u[y * N + x]
            oX0, oX1 = min(oX0, p[0]), max(oX1, p[0])
            oY0, oY1 = min(oY0, p[1]), max(oY1, p[1])
            if x == 0:
                iX0 = max(iX0, p[0])
            if x == N - 1:
                iX1 = min(iX1, p[0])
            if y == 0:
                iY0 = max(iY0, p[1])
            if y == N - 1:
                iY1 = min(iY1, p[1])
    inner = (np.float32(iX0), np.float32(iY0), np.float32(iX1 - iX0), np.float32(iY1 - iY0))
    outer = (np.float32(oX0), np.float32(oY0), np.float32(oX1 - oX0), np.float32(oY1 - oY0))
    return inner, outer


def get_optimal_new_camera_matrix(K, dist, size, alpha=1.0):
    """cvGetOptimalNewCameraMatrix(K, dist, size, alpha, size, &validRoi, centerPrincipalPoint = 0)
    -> (3x3 float64 new camera matrix, validRoi (x, y, w, h) ints)."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    w, h = size
    # inscribed / circumscribed rectangles in NORMALISED coordinates (no new camera matrix): the new camera then
    # maps that rectangle onto the viewport
    inner, outer = get_rectangles(K, dist, np.eye(3), size)
    f32 = np.float32
    fx0 = float(f32(w - 1) / inner[2]); fy0 = float(f32(h - 1) / inner[3])      # int / float -> float
    cx0 = -fx0 * float(inner[0]); cy0 = -fy0 * float(inner[1])
    fx1 = float(f32(w - 1) / outer[2]); fy1 = float(f32(h - 1) / outer[3])
    cx1 = -fx1 * float(outer[0]); cy1 = -fy1 * float(outer[1])
    M = np.zeros((3, 3), np.float64)
    M[0, 0] = fx0 * (1 - alpha) + fx1 * alpha
    M[1, 1] = fy0 * (1 - alpha) + fy1 * alpha
    M[0, 2] = cx0 * (1 - alpha) + cx1 * alpha
    M[1, 2] = cy0 * (1 - alpha) + cy1 * alpha
    M[2, 2] = 1.0
    inner2, _ = get_rectangles(K, dist, M, size)
    # cv::Rect r = inner (saturate_cast<int> = round half to even), clipped to the image
    rx, ry, rw, rh = (int(np.rint(float(v))) for v in inner2)
    x0, y0 = max(rx, 0), max(ry, 0)
    x1, y1 = min(rx + rw, w), min(ry + rh, h)
    roi = (x0, y0, x1 - x0, y1 - y0) if (x1 > x0 and y1 > y0) else (0, 0, 0, 0)
    return M, roi


def _inv3(S):
    """cv::invert of a 3x3 double matrix (the closed form OpenCV uses for n <= 3)."""
    det = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
           + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    d = 1.0 / det
    t = np.zeros(9, np.float64)
    t[0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[3] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[4] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[5] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[6] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[7] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[8] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t


def undistort_maps(K, dist, P, size):
    """The CV_16SC2 / CV_16UC1 maps cv::undistort builds stripe by stripe (stripe = max(1, 4096 / width) rows, the new
    camera's cy shifted by the stripe's first row).  -> map_xy int16 [h, w, 2], map_frac uint16 [h, w]."""
    w, h = size
    k = _dist8(dist)
    k1, k2, p1, p2, k3, k4, k5, k6 = k
    A = np.asarray(K, np.float64).reshape(3, 3)
    Ar = np.asarray(P, np.float64).reshape(3, 3).copy()
    fx, fy, u0, v0 = A[0, 0], A[1, 1], A[0, 2], A[1, 2]
    stripe0 = min(max(1, (1 << 12) // max(w, 1)), h)
    vv0 = Ar[1, 2]
    mxy = np.zeros((h, w, 2), np.int16)
    mfr = np.zeros((h, w), np.uint16)
    for y in range(0, h, stripe0):
        n = min(stripe0, h - y)
        Ar[1, 2] = vv0 - y
        ir = _inv3(Ar)
        for i in range(n):
            # _x += ir[0] per column: a sequential accumulation, which np.cumsum reproduces term by term
            X = np.cumsum(np.concatenate([[i * ir[1] + ir[2]], np.full(w - 1, ir[0])]))
            Y = np.cumsum(np.concatenate([[i * ir[4] + ir[5]], np.full(w - 1, ir[3])]))
            W = np.cumsum(np.concatenate([[i * ir[7] + ir[8]], np.full(w - 1, ir[6])]))
            wi = 1.0 / W
            x = X * wi
            yy = Y * wi
            x2, y2 = x * x, yy * yy
            r2 = x2 + y2
            _2xy = 2 * x * yy
            kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
            u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0
            v = fy * (yy * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0
            lim = 2147483647.0
            iu = np.rint(np.clip(u * INTER_TAB_SIZE, -lim - 1, lim)).astype(np.int64)
            iv = np.rint(np.clip(v * INTER_TAB_SIZE, -lim - 1, lim)).astype(np.int64)
            mxy[y + i, :, 0] = (iu >> INTER_BITS).astype(np.int16)          # (short) cast wraps
            mxy[y + i, :, 1] = (iv >> INTER_BITS).astype(np.int16)
            mfr[y + i] = ((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1))).astype(np.uint16)
    return mxy, mfr


def bilinear_weights():
    """BilinearTab_i: 1024 x 4 weights, scale 2^15.  (1 - fy)(1 - fx) etc. are exact multiples of 32 here, so the sums
    are exact -- except the identity entry, where saturate_cast<short>(32768) = 32767 and OpenCV's fix-up adds the
    missing 1 to the last weight."""
    tab = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), np.int64)
    for fy in range(INTER_TAB_SIZE):
        for fx in range(INTER_TAB_SIZE):
            tab[fy * INTER_TAB_SIZE + fx] = [(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32,
                                             fy * fx * 32]
    tab[0] = [32767, 0, 0, 1]
    return tab


def remap_linear(src, mxy, mfr):
    """cv::remap(src, dst, map1, map2, INTER_LINEAR, BORDER_CONSTANT(0)) for 8-bit images, 1 or 3 channels."""
    src = np.asarray(src, np.uint8)
    if src.ndim == 2:
        src = src[:, :, None]
    H, W, C = src.shape
    h, w = mfr.shape
    tab = bilinear_weights()[mfr.astype(np.int64)]                  # [h, w, 4]
    sx = mxy[:, :, 0].astype(np.int64)
    sy = mxy[:, :, 1].astype(np.int64)
    acc = np.zeros((h, w, C), np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        px = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        px[~inside] = 0
        acc += px * tab[:, :, k][:, :, None]
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out[:, :, 0] if out.shape[2] == 1 else out


def undistort_image(src, K, dist):
    """localizeImage.cc:149-177 -> (cropped undistorted image, new camera matrix, validRoi)."""
    h, w = np.asarray(src).shape[:2]
    P, roi = get_optimal_new_camera_matrix(K, dist, (w, h), 1.0)
    mxy, mfr = undistort_maps(K, dist, P, (w, h))
    full = remap_linear(src, mxy, mfr)
    x, y, rw, rh = roi
    return full[y:y + rh, x:x + rw].copy(), P, roi
