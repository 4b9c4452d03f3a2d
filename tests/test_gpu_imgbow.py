"""GPU: sfmloc_imgbow -- the query-side BoW vector from the image as one resident chain -- against the stage-level calls
(sfmloc_dense_gray -> sfmloc_akaze_compute -> sfmloc_bof_compute, engine.DenseBow) and against the CPU restatement of
DenseLocalFeatureWrapper::calcDenseLocalFeature -> PcaWrapper::calcPcaProject -> BoFSpatialPyramids::calcBoF
(DenseLocalFeatureWrapper.cpp:83-183, PcaWrapper.cpp:67-89, BoFSpatialPyramids.cpp:108-302): the same float64 vector, bit
for bit; a gray source equals its three-equal-channel colour read; the vector written into a query's resident slot is the
float32 vector the shortlist would be given."""
import numpy as np
import pytest

import sfmlocalization_amd as S
import synthdata as synth
from sfmlocalization_amd import engine, fileio

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    root = tmp_path_factory.mktemp("bowmodel")
    rng = np.random.Generator(np.random.PCG64(11))
    K, npca = 100, 32
    pca = {"DimPCA": npca, "EigenVectorsPCA": rng.normal(size=(61, 61)).astype(np.float32),
           "EigenValuesPCA": rng.uniform(0.5, 4.0, (61, 1)).astype(np.float32),
           "MeanPCA": rng.uniform(0, 255, (1, 61)).astype(np.float32)}
    bow = {"ResizedImageSize": 300, "UseSpatialPyramid": 1, "PyramidLevel": 2, "NormBofFeatureType": "L1",
           "Centers": (rng.normal(size=(K, npca)) * 30).astype(np.float32)}
    fileio.write_cv_yaml(root / "PCAfile.yml", pca)
    fileio.write_cv_yaml(root / "BOWfile.yml", bow)
    return str(root / "BOWfile.yml"), str(root / "PCAfile.yml"), pca, bow


@pytest.mark.parametrize("shape", [(480, 640), (300, 300), (271, 483)])
def test_resident_chain_equals_the_staged_calls_and_the_oracle(model, oracle_c, shape):
    bow_file, pca_file, pca, bowm = model
    h, w = shape
    rng = np.random.Generator(np.random.PCG64(h))
    gray = synth.texture_image(60 + h, h, w, n_blobs=700, n_rects=120)
    bgr = np.stack([np.roll(gray, 3, 1), gray, (gray.astype(int) * 3 // 4 + rng.integers(0, 40, gray.shape)).astype(np.uint8)], 2)
    staged = engine.DenseBow(bow_file, pca_file)
    want = staged.compute(bgr, staged=True)
    ib = S.ImgBow.from_files(bow_file, pca_file, w, h, 3)
    assert ib.dim == 500
    for _ in range(3):                                      # repeated calls reuse every buffer
        got = ib.compute(bgr)
        np.testing.assert_array_equal(bits(got), bits(want))
    # the CPU restatement of the whole chain
    og = oracle_c.dense_gray(bgr, 300)
    grid = engine.dense_grid_keypoints(300)
    od, _ = oracle_c.akaze_compute(og, grid)
    ob = oracle_c.bof(od[:, :61].astype(np.float32), grid[:, :2].copy(), bowm["Centers"], 300, 2, 2, pca["MeanPCA"],
                      pca["EigenVectorsPCA"], pca["EigenValuesPCA"], int(pca["DimPCA"]))
    np.testing.assert_array_equal(bits(got), bits(ob))
    # a gray source = the colour read of a gray file (three equal channels)
    ibg = S.ImgBow.from_files(bow_file, pca_file, w, h, 1)
    g1 = ibg.compute(gray)
    g3 = ib.compute(np.stack([gray, gray, gray], 2))
    np.testing.assert_array_equal(bits(g1), bits(g3))
    np.testing.assert_array_equal(bits(g1), bits(staged.compute(np.stack([gray, gray, gray], 2), staged=True)))
    for o in (ib, ibg, staged):
        o.close()


def test_vector_lands_in_the_query_slot_and_drives_the_shortlist(model):
    """compute(image, query) on the localising context's stream, then sfmloc_localize_bow_begin(ctx, query, NULL, k):
    the shortlist and the pose are those of set_bow(float32(vector)) -- no host copy of the vector in between."""
    bow_file, pca_file, pca, bowm = model
    m = synth.make_map(5, n_views=60, desc_per_view=300, views_per_place=10, landmarks_per_place=250, obs_per_view=120)
    rng = np.random.Generator(np.random.PCG64(9))
    imgs = [synth.texture_image(300 + k, 480, 640, n_blobs=500, n_rects=100) for k in range(4)]
    ib = S.ImgBow.from_files(bow_file, pca_file, 640, 480, 1)
    vecs = [ib.compute(g) for g in imgs]
    # a map whose views' .bow vectors are noisy copies of the four images' vectors
    bow = np.stack([vecs[v % 4] for v in range(m.n_views)]).astype(np.float32)
    bow += rng.normal(0, 0.002, bow.shape).astype(np.float32)
    with S.Map(m.view_id, m.view_off, m.desc, params=S.default_params(ransac_round=25), view_wh=m.view_wh,
               kpt_xy=m.kpt_xy, row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X,
               intrinsic=m.intrinsic, bow=bow) as dm:
        ctx = dm.context()
        ib.share_stream(ctx)
        for k, g in enumerate(imgs):
            q = synth.make_query(m, 70 + k, n_feat=400, n_copies=150)
            dq_a = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            dq_b = dm.query(q.desc, q.kpt_xy, q.width, q.height)
            dq_a.set_bow(vecs[k].astype(np.float32))
            ctx.begin_bow(dq_a, None, 15)
            want = ctx.end()
            assert ib.compute(g, dq_b) is None              # asynchronous: nothing returned, nothing waited for
            ctx.begin_bow(dq_b, None, 15)
            got = ctx.end()
            assert S.capi.result_fingerprint(*got) == S.capi.result_fingerprint(*want)
            sel = dm.bow_select(vecs[k].astype(np.float32), 15)
            assert set(int(v) % 4 for v in sel) == {k}
            dq_a.close()
            dq_b.close()
        ib.share_stream(None)
        ctx.close()
    ib.close()


def test_a_batch_of_frames_in_one_session_equals_the_single_calls(model):
    """sfmloc_imgbow_compute_batch: the extractors of n frames record their chains for ONE gang session -- resize + gray +
    min-max, the 300 x 300 scale space, orientation + M-LDB at the grid, PCA / words / histogram: one launch per kernel for
    all frames -- and every frame's vector is the single call's, bit for bit; batches of different sizes on the same
    extractors, single calls in between, extractors that never had a stream of their own, and the session's stream
    ordered in front of a context (order_before) all work."""
    bow_file, pca_file, pca, bowm = model
    w, h = 640, 480
    imgs = [synth.texture_image(500 + k, h, w, n_blobs=300 + 90 * k, n_rects=60 + 10 * k) for k in range(6)]
    single = S.ImgBow.from_files(bow_file, pca_file, w, h, 1)
    want = [single.compute(g) for g in imgs]
    ibs = [S.ImgBow.from_files(bow_file, pca_file, w, h, 1) for _ in range(6)]
    for n, first in ((6, 0), (3, 2), (1, 5), (4, 1)):
        sel = [(first + k) % 6 for k in range(n)]
        S.ImgBow.compute_batch(ibs[:n], [imgs[i] for i in sel])
        for e, i in zip(ibs, sel):
            np.testing.assert_array_equal(bits(e.vector_read()), bits(want[i]))
        np.testing.assert_array_equal(bits(ibs[2].compute(imgs[4])), bits(want[4]))      # a single call on a member
    for o in ibs + [single]:
        o.close()
