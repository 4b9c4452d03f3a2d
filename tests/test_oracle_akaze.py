"""CPU: the AKAZE / M-LDB restatement (oracle/sfm_oracle_akaze.c) behaves like a feature extractor should.
OpenCV is not in the image (parity unpinned): checks are structural -- descriptor geometry, invariances."""
import math

import numpy as np

from oracle import oracle_np as onp
import synthdata as synth


def pad64(d):
    out = np.zeros((len(d), 64), np.uint8)
    out[:, :61] = d
    return out


def test_levels_follow_opencv_allocation(oracle_c):
    lv = oracle_c.akaze_levels(640, 480)
    assert len(lv) == 16 and [tuple(x) for x in lv[::4]] == [(640, 480), (320, 240), (160, 120), (80, 60)]
    assert len(oracle_c.akaze_levels(300, 300)) == 8           # 75 x 75 is below the 80 x 40 limit -> 2 octaves
    assert len(oracle_c.akaze_levels(1920, 1080)) == 16


def test_fixed_order_trig_is_accurate(oracle_c):
    rng = np.random.Generator(np.random.PCG64(1))
    for _ in range(2000):
        x, y = rng.normal(size=2) * 10.0 ** rng.uniform(-3, 3)
        a, s, c = oracle_c.akaze_math(float(np.float32(x)), float(np.float32(y)))
        ref = math.atan2(np.float32(y), np.float32(x)) % (2 * math.pi)
        assert abs(a - ref) < 2e-6 or abs(abs(a - ref) - 2 * math.pi) < 2e-6
        assert abs(s - math.sin(a)) < 5e-7 and abs(c - math.cos(a)) < 5e-7
    assert list(oracle_c.akaze_math(0.0, 0.0)) == [0.0, 0.0, 1.0]


def test_descriptor_layout_and_translation_invariance(oracle_c):
    g = synth.texture_image(1)
    kp, desc = oracle_c.akaze_detect_and_compute(g)
    assert len(kp) > 150 and desc.shape[1] == 61
    assert (desc[:, 60] & 0xC0 == 0).all()                     # 486 = 60*8 + 6 valid bits
    assert (kp[:, 2] > 4).all() and (kp[:, 3] >= 0).all() and (kp[:, 3] < 2 * math.pi + 1e-5).all()
    assert (kp[:, 0] > 10).all() and (kp[:, 0] < 630).all()   # descriptor support stays inside the image
    g2 = np.roll(np.roll(g, 8, axis=0), 16, axis=1)
    kp2, desc2 = oracle_c.akaze_detect_and_compute(g2)
    j0, d0, j1, d1 = onp.hamming_2nn(pad64(desc2), pad64(desc))
    ok = onp.ratio_accept(d0, d1, 0.6)
    assert ok.sum() > 0.8 * len(kp)
    shift = kp2[j0[ok], :2] - kp[ok][:, :2]
    assert np.abs(np.median(shift, axis=0) - [16, 8]).max() < 0.05
    assert (d0[ok] == 0).sum() > 0.6 * ok.sum()                # away from the wrapped border the bits are identical


def test_rotation_by_90_degrees_is_matched(oracle_c):
    g = synth.texture_image(2, 480, 480)
    kp, desc = oracle_c.akaze_detect_and_compute(g)
    kp2, desc2 = oracle_c.akaze_detect_and_compute(np.ascontiguousarray(np.rot90(g)))
    j0, d0, j1, d1 = onp.hamming_2nn(pad64(desc2), pad64(desc))
    ok = onp.ratio_accept(d0, d1, 0.7)
    assert ok.sum() > 0.4 * len(kp)
    # rot90 (counter-clockwise): (x, y) -> (y, W - 1 - x)
    a, b = kp[ok], kp2[j0[ok]]
    err = np.abs(np.stack([a[:, 1] - b[:, 0], (479 - a[:, 0]) - b[:, 1]], 1))
    assert np.median(err) < 1.0
    dang = (a[:, 3] - b[:, 3]) % (2 * math.pi)
    assert np.abs(np.median(dang) - math.pi / 2) < 0.2 or np.abs(np.median(dang) - 3 * math.pi / 2) < 0.2


def test_dense_compute_on_grid_keypoints(oracle_c):
    g = synth.texture_image(3, 300, 300)
    xs = np.arange(0, 300, 6, dtype=np.float32)
    kin = []
    size = 4.0
    for s in range(4):                                          # DenseFeatureDetector.cpp:44-69
        for y in xs:
            for x in xs:
                kin.append((x, y, size, s))
        size *= 1.5
    kin = np.array(kin, np.float32)
    assert len(kin) == 10000
    desc, ang = oracle_c.akaze_compute(g, kin)
    assert desc.shape == (10000, 61) and np.isfinite(ang).all()
    assert 0.2 < np.unpackbits(desc, axis=1)[:, :486].mean() < 0.8
