"""CPU: the BoW oracle (oracle/sfm_oracle_bow.c) against NumPy."""
import numpy as np


def dense_grid(rng, n_side=50, step=6, in_dim=61):
    xs = (np.arange(n_side) * step + 3).astype(np.float32)
    kxy = np.stack(np.meshgrid(xs, xs), -1).reshape(-1, 2)
    desc = rng.integers(0, 256, (len(kxy), in_dim)).astype(np.float32)
    return desc, kxy


def test_bow_select_is_exact_knn_with_set_semantics(oracle_c):
    rng = np.random.Generator(np.random.PCG64(1))
    bow = np.sqrt(rng.random((300, 500)).astype(np.float32))
    q = bow[17] + rng.normal(0, 0.01, 500).astype(np.float32)
    bow[200] = bow[40]                                            # exact tie: lower view index wins
    d = ((bow.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1)
    for k in (1, 5, 20, 100, 299):
        sel = oracle_c.bow_select(bow, q, k)
        assert (np.diff(sel.astype(np.int64)) > 0).all() and len(sel) == k
        exp = np.sort(np.argsort(d, kind="stable")[:k])
        assert set(sel) == set(exp) or abs(np.sort(d)[k - 1] - np.sort(d)[k]) < 1e-5
    assert 17 in oracle_c.bow_select(bow, q, 1)
    cand = np.arange(0, 300, 3, dtype=np.uint32)
    sel = oracle_c.bow_select(bow, q, 10, cand)
    assert set(sel) <= set(cand) and len(sel) == 10
    q2 = bow[40].copy()
    sel = oracle_c.bow_select(bow, q2, 1)
    assert list(sel) == [40]                                      # 40 and 200 tie at distance 0
    assert abs(oracle_c.bow_dist(bow[3], q) - d[3]) < 1e-4 * max(1.0, d[3])


def test_bof_matches_numpy(oracle_c):
    rng = np.random.Generator(np.random.PCG64(2))
    desc, kxy = dense_grid(rng)
    K, n_pca = 100, 32
    mean = desc.mean(0)
    evec = np.linalg.qr(rng.normal(size=(61, 61)))[0][:n_pca].astype(np.float32)
    evals = (np.linspace(5000, 100, n_pca)).astype(np.float32)
    proj = ((desc - mean) @ evec.T) / evals
    centers = proj[rng.choice(len(proj), K, replace=False)] + rng.normal(0, 1e-3, (K, n_pca)).astype(np.float32)
    got = oracle_c.bof(desc, kxy, centers, pca_mean=mean, pca_eigvec=evec, pca_eigval=evals, n_pca=n_pca)
    assert got.shape == (500,)
    idx = ((proj[:, None, :] - centers[None]) ** 2).sum(2).argmin(1)
    exp = np.zeros((5, K))
    for i, (x, y) in zip(idx, kxy):
        exp[0, i] += 1
        exp[1 + int(y >= 150) * 2 + int(x >= 150), i] += 1
    exp /= len(kxy)
    exp = np.sqrt(exp / exp.sum(1, keepdims=True))
    np.testing.assert_allclose(got.reshape(5, K), exp, atol=1e-12)
    assert abs((got.reshape(5, K) ** 2).sum(1) - 1).max() < 1e-12      # L1-sqrt: squares of a cell sum to one
    # no PCA, no pyramid, L2 norm
    c2 = desc[rng.choice(len(desc), 20, replace=False)]
    g2 = oracle_c.bof(desc, kxy, c2, levels=1, norm_type=1)
    assert g2.shape == (20,) and abs((g2 ** 2).sum() - 1) < 1e-12
