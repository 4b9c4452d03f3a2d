"""GPU: BoW shortlist kernels (K7 BoF vector, K8 top-k over the views' .bow vectors) against the oracle:
selected view sets and histogram vectors bit-exact."""
import numpy as np
import pytest

import sfmlocalization_amd as S
from sfmlocalization_amd import synth

pytestmark = pytest.mark.gpu


def test_bow_select_matches_oracle(oracle_c):
    rng = np.random.Generator(np.random.PCG64(5))
    nv = 3000
    bow = np.sqrt(rng.random((nv, 500))).astype(np.float32)
    bow[1500] = bow[100]
    bow[2999] = bow[100]
    m = synth.make_map(1, n_views=nv, desc_per_view=2, views_per_place=100, landmarks_per_place=10, obs_per_view=1)
    with S.Map(m.view_id, m.view_off, m.desc, bow=bow) as dm:
        for trial in range(6):
            q = bow[rng.integers(0, nv)] + rng.normal(0, 0.02, 500).astype(np.float32)
            if trial == 0:
                q = bow[100].copy()                                   # three views tie at distance 0
            for k in (1, 2, 20, 100, 200):
                np.testing.assert_array_equal(dm.bow_select(q, k), oracle_c.bow_select(bow, q, k))
            cand = np.sort(rng.choice(nv, 700, replace=False)).astype(np.uint32)
            np.testing.assert_array_equal(dm.bow_select(q, 100, cand), oracle_c.bow_select(bow, q, 100, cand))
        assert list(dm.bow_select(bow[100], 2)) == [100, 1500]
        with pytest.raises(S.SfmlocError):
            dm.bow_select(q, nv)                                      # CV_Assert(knn < viewList.size())
    with S.Map(m.view_id, m.view_off, m.desc) as dm2:
        with pytest.raises(S.SfmlocError):
            dm2.bow_select(q, 5)                                      # no .bow vectors in this map


def test_bof_vector_matches_oracle(oracle_c):
    rng = np.random.Generator(np.random.PCG64(6))
    xs = (np.arange(50) * 6 + 3).astype(np.float32)
    kxy = np.repeat(np.stack(np.meshgrid(xs, xs), -1).reshape(-1, 2), 4, axis=0)      # 50 x 50 x 4 scales
    desc = rng.integers(0, 256, (len(kxy), 61)).astype(np.float32)
    K, n_pca = 100, 32
    mean = desc.mean(0).astype(np.float32)
    evec = np.linalg.qr(rng.normal(size=(61, 61)))[0][:n_pca].astype(np.float32)
    evals = np.linspace(5000, 100, n_pca).astype(np.float32)
    proj = ((desc - mean) @ evec.T) / evals
    centers = proj[rng.choice(len(proj), K, replace=False)].astype(np.float32)
    b = S.BofModel(centers, 61, pca_mean=mean, pca_eigvec=evec, pca_eigval=evals, n_pca=n_pca)
    assert b.dim == 500
    got = b.compute(desc, kxy)
    exp = oracle_c.bof(desc, kxy, centers, pca_mean=mean, pca_eigvec=evec, pca_eigval=evals, n_pca=n_pca)
    np.testing.assert_array_equal(got.view(np.uint64), exp.view(np.uint64))
    b.close()
    b2 = S.BofModel(desc[:30].copy(), 61, use_pyramid=False, norm="L2")
    np.testing.assert_array_equal(b2.compute(desc, kxy).view(np.uint64),
                                  oracle_c.bof(desc, kxy, desc[:30], levels=1, norm_type=1).view(np.uint64))
    b2.close()
