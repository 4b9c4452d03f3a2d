"""GPU: the image in front of the path without leaving the device.

sfmloc_akaze_detect_resident leaves a frame's features on the device as a query block (descriptor rows + zero padding,
keypoints, keypoints after the .feat text round trip), sfmloc_imgbow_compute leaves the frame's BoW vector there, and
sfmloc_query_create_view makes the query over them: frame -> pose with one 4-byte count crossing PCIe.  The result
must be the one the staged route gives -- extractAKAZESingleImg's outputs downloaded (sfmloc_akaze_detect_and_compute),
the query uploaded (sfmloc_query_create, which applies the .feat round trip on the host: AKAZEOpenCV.cpp:80-81,106-111),
the BoW vector computed by the three stage-level calls and uploaded (localization.cpp:323,346-368) -- bit for bit."""
import numpy as np
import pytest

import imageworld as iw
import sfmlocalization_amd as S
import synthdata as synth
from sfmlocalization_amd import capi, engine, fileio

pytestmark = pytest.mark.gpu
W, H = 640, 480


def test_round6_on_the_device_is_the_text_round_trip():
    """geom::round6_dev (the .feat round trip of a coordinate inside k_orient_describe) against the host helper
    (snprintf %.6g + strtof): random pixel coordinates, exact decimal ties (x.xx5 representable in binary), powers of
    ten and their neighbours, small and large magnitudes, negatives."""
    rng = np.random.Generator(np.random.PCG64(4))
    v = [rng.uniform(0, 20000, 400000), rng.uniform(0, 10, 50000), rng.uniform(0, 1e-3, 20000),
         rng.uniform(1e5, 2e7, 50000), -rng.uniform(0, 5000, 20000)]
    ties = []
    for k in range(3, 13):                                   # n + odd / 2^k: the 6th digit falls on an exact half
        base = rng.integers(1, 16384, 20000).astype(np.float64)
        ties.append(base + (2 * rng.integers(0, 2 ** (k - 1), 20000) + 1) / 2.0 ** k)
    pw = np.array([10.0 ** e for e in range(-9, 14)])
    v += ties + [pw, np.nextafter(pw.astype(np.float32), 0), np.nextafter(pw.astype(np.float32), np.float32(1e30)),
                 np.array([999999.5, 999999.4, 99999.95, 9.9999949, 9.999995, 123456.5, 1234565.0, 1234575.0, 0.0])]
    x = np.concatenate([np.asarray(a, np.float64) for a in v]).astype(np.float32)
    want = capi.feat_round_trip(x)
    got = S.debug_math(10, x.astype(np.float64), 1)[:, 0].astype(np.float32)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, (len(bad), x[bad[:5]], got[bad[:5]], want[bad[:5]])


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    root = tmp_path_factory.mktemp("imgworld")
    rng = np.random.Generator(np.random.PCG64(12))
    K, npca = 40, 16
    pca = {"DimPCA": npca, "EigenVectorsPCA": rng.normal(size=(61, 61)).astype(np.float32),
           "EigenValuesPCA": rng.uniform(0.5, 4.0, (61, 1)).astype(np.float32),
           "MeanPCA": rng.uniform(0, 255, (1, 61)).astype(np.float32)}
    bowm = {"ResizedImageSize": 300, "UseSpatialPyramid": 1, "PyramidLevel": 2, "NormBofFeatureType": "L1",
            "Centers": (rng.normal(size=(K, npca)) * 30).astype(np.float32)}
    fileio.write_cv_yaml(root / "PCAfile.yml", pca)
    fileio.write_cv_yaml(root / "BOWfile.yml", bowm)
    dense = engine.DenseBow(str(root / "BOWfile.yml"), str(root / "PCAfile.yml"))
    w = iw.build(S, 5, 36, 6, tiles=2, n_pad_views=30, pad_desc_per_view=500, dense_bow=dense)
    dense.close()
    return w, str(root / "BOWfile.yml"), str(root / "PCAfile.yml")


def test_frame_to_pose_on_the_device_equals_the_staged_route(world):
    w, bow_file, pca_file = world
    m = w.m
    staged_bow = engine.DenseBow(bow_file, pca_file)
    with S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25), view_wh=m.view_wh, kpt_xy=m.kpt_xy,
               row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
               bow=w.bow) as dm:
        ctx = dm.context()
        ak = S.Akaze(W, H)
        ak.share_stream(ctx)
        ib = S.ImgBow.from_files(bow_file, pca_file, W, H, 1)
        ib.share_stream(ctx)
        n_ok = 0
        for i, frame in enumerate(w.frames):
            # staged: everything through the host
            kp, desc = ak.detect_and_compute(frame)
            dq = dm.query(desc, kp[:, :2], W, H)
            dq.set_bow(staged_bow.compute(np.stack([frame, frame, frame], 2), staged=True).astype(np.float32))
            ctx.begin_bow(dq, None, 12)
            want = ctx.end()
            dq.close()
            # resident: nothing but the count leaves the device
            n = ak.detect_resident(frame)
            assert n == len(kp) and n > 500
            ib.compute(frame, None, want_vector=False)
            q = ak.query_view(dm, n, ib.vector_dev())
            ctx.begin_bow(q, None, 12)
            got = ctx.end()
            q.close()
            assert capi.result_fingerprint(*got) == capi.result_fingerprint(*want), i
            n_ok += int(got[0].ok)
            if got[0].ok:
                assert np.abs(np.array(got[0].center) - w.frame_C[i]).max() < 0.3
        assert n_ok >= 4
        # a batch of frames in one call, each then localised: the same results again
        aks = [S.Akaze(W, H) for _ in range(3)]
        for a in aks:
            a.share_stream(ctx)
        ns = S.Akaze.detect_resident_batch(aks, list(w.frames[:3]))
        for a, n, frame in zip(aks, ns, w.frames[:3]):
            kp, desc = ak.detect_and_compute(frame)
            assert n == len(kp)
            dq = dm.query(desc, kp[:, :2], W, H)
            ctx.begin(dq)
            want = ctx.end()
            dq.close()
            q = a.query_view(dm, n)
            ctx.begin(q)
            got = ctx.end()
            q.close()
            assert capi.result_fingerprint(*got) == capi.result_fingerprint(*want)
        for a in aks + [ak]:
            a.close()
        ib.close()
        ctx.close()
    staged_bow.close()
