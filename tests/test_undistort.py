"""The server's per-user undistortion (localizeImage.cc:149-177): sfmloc_undistorter_* against the NumPy restatement
(oracle/oracle_undistort.py).  The once-per-camera plan (new camera matrix, valid region, fixed-point maps) is host
arithmetic and is compared here without a GPU, bit for bit; the per-image remap is compared on the GPU
(tests/test_gpu_undistort.py).  Parity with OpenCV itself is unpinned (no OpenCV in this image); what pins the
restatement is geometry: undistorting an image that was rendered THROUGH the distortion model gives back the
pinhole rendering."""
import numpy as np
import pytest

import sfmlocalization_amd as S
from oracle import oracle_undistort as ou
from undistort_cameras import CAMERAS

@pytest.mark.parametrize("cam", range(len(CAMERAS)))
def test_plan_equals_restatement(cam):
    K, dist, (w, h) = CAMERAS[cam]
    P, roi = ou.get_optimal_new_camera_matrix(K, dist, (w, h), 1.0)
    with S.Undistorter(K, dist, w, h) as u:
        np.testing.assert_array_equal(u.new_camera.view(np.uint64), P.view(np.uint64))
        assert u.roi == tuple(roi)
        if w * h <= 640 * 480:                      # the NumPy maps of a 1080p image take a while
            xy, fr = u.maps()
            exy, efr = ou.undistort_maps(K, dist, P, (w, h))
            np.testing.assert_array_equal(xy, exy)
            np.testing.assert_array_equal(fr, efr)
    assert roi[2] > w // 2 and roi[3] > h // 2


def test_errors():
    K = CAMERAS[0][0]
    with pytest.raises(S.SfmlocError):
        S.Undistorter(K, [0.1, 0.2, 0.3], 640, 480)          # 3 coefficients is not a model OpenCV knows
    with pytest.raises(S.SfmlocError):
        S.Undistorter(np.zeros((3, 3)), [], 640, 480)
    with pytest.raises(S.SfmlocError):
        S.Undistorter(K, [], 1, 480)


def test_apply_needs_the_gpu():
    """The plan is host arithmetic; the per-image remap is a kernel and there is no CPU fallback for it."""
    if S.device_count() > 0:
        pytest.skip("GPU present")
    K, dist, (w, h) = CAMERAS[2]
    with S.Undistorter(K, dist, w, h) as u:
        with pytest.raises(S.SfmlocError) as ei:
            u.apply(np.zeros((h, w), np.uint8))
        assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)      # SFMLOC_ENODEV


def _pattern(x, y):
    """A smooth scene on the normalised image plane."""
    return 128 + 60 * np.sin(14 * x) * np.cos(11 * y) + 50 * np.cos(5 * x * y + 3 * y)


def test_restatement_inverts_the_distortion_model():
    """Render a scene through (K, dist) -> distorted image; undistort; compare with the scene rendered through the new
    pinhole camera inside the valid region.  Bilinear resampling of a smooth scene: a few gray levels."""
    K, dist, (w, h) = CAMERAS[0]
    k1, k2, p1, p2, k3 = dist
    # distorted image: pixel (u, v) sees the normalised point found by inverting the model (fixed-point iteration)
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    xd, yd = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
    x, y = xd.copy(), yd.copy()
    for _ in range(40):
        r2 = x * x + y * y
        kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (xd - dx) / kr, (yd - dy) / kr
    img = np.clip(_pattern(x, y), 0, 255).astype(np.uint8)
    out, P, roi = ou.undistort_image(img, K, dist)
    rx, ry, rw, rh = roi
    vv, uu = np.mgrid[ry:ry + rh, rx:rx + rw].astype(np.float64)
    exp = _pattern((uu - P[0, 2]) / P[0, 0], (vv - P[1, 2]) / P[1, 1])
    inner = (slice(2, -2), slice(2, -2))
    err = np.abs(out[inner].astype(np.float64) - exp[inner])
    assert np.percentile(err, 99) < 4.0 and err.mean() < 1.2, (np.percentile(err, 99), err.mean())
    # without distortion the whole image is valid and the remap is the identity
    K5, _, (w5, h5) = CAMERAS[4]
    img5 = (np.arange(w5 * h5) % 251).astype(np.uint8).reshape(h5, w5)
    out5, P5, roi5 = ou.undistort_image(img5, K5, [])
    assert roi5[2] >= w5 - 2 and roi5[3] >= h5 - 2
