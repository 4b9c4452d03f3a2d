#!/usr/bin/env python3
"""Randomised parity campaign for the BoW side: sfmloc_dense_gray (cubic resize + BGR2GRAY +
min-max), sfmloc_bof_compute (PCA / eigenvalue, nearest centre, pyramid histogram, norm) and sfmloc_bow_select (exact
top-k with ties) against the CPU restatement, on random sizes, models and heavily tied distance sets.
usage: fuzz_bow.py [n_cases] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402
from oracle import oracle_c  # noqa: E402


def one(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    # dense gray
    w, h = int(rng.integers(17, 900)), int(rng.integers(17, 700))
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if rng.uniform() < 0.5:                                  # smooth content: the cubic taps see real gradients
        y, x = np.mgrid[0:h, 0:w]
        img = np.clip(np.stack([128 + 100 * np.sin(x / rng.uniform(3, 40)), 128 + 100 * np.cos(y / rng.uniform(3, 40)),
                                (x + y) % 256], -1) + rng.normal(0, 8, (h, w, 3)), 0, 255).astype(np.uint8)
    size = int(rng.choice([64, 100, 300]))
    assert np.array_equal(S.dense_gray(img, size), oracle_c.dense_gray(img, size)), "dense_gray"
    # BoF vector
    n = int(rng.integers(50, 3000))
    kxy = rng.uniform(0, 300, (n, 2)).astype(np.float32)
    desc = rng.integers(0, 256, (n, 61)).astype(np.float32)
    K = int(rng.integers(2, 120))
    if rng.uniform() < 0.6:
        n_pca = int(rng.integers(2, 61))
        mean = desc.mean(0).astype(np.float32)
        evec = np.linalg.qr(rng.normal(size=(61, 61)))[0][:n_pca].astype(np.float32)
        evals = np.sort(rng.uniform(50, 5000, n_pca))[::-1].astype(np.float32)
        proj = ((desc - mean) @ evec.T) / evals
        centers = proj[rng.choice(n, K, replace=K > n)].astype(np.float32)
        b = S.BofModel(centers, 61, pca_mean=mean, pca_eigvec=evec, pca_eigval=evals, n_pca=n_pca)
        exp = oracle_c.bof(desc, kxy, centers, pca_mean=mean, pca_eigvec=evec, pca_eigval=evals, n_pca=n_pca)
    else:
        centers = desc[rng.choice(n, K, replace=K > n)].copy()
        b = S.BofModel(centers, 61, use_pyramid=False, norm="L2")
        exp = oracle_c.bof(desc, kxy, centers, levels=1, norm_type=1)
    got = b.compute(desc, kxy)
    b.close()
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), "bof vector"
    # shortlist with ties
    nv = int(rng.integers(3, 4000))
    dim = int(rng.choice([8, 64, 500]))
    bow = rng.integers(0, 3, (nv, dim)).astype(np.float32) * 0.5           # few distinct values: many exact ties
    m = synth.make_map(seed % 1000, n_views=nv, desc_per_view=1, views_per_place=max(1, nv), landmarks_per_place=2,
                       obs_per_view=1)
    with S.Map(m.view_id, m.view_off, m.desc, bow=bow) as dm:
        for _ in range(3):
            q = rng.integers(0, 3, dim).astype(np.float32) * 0.5
            k = int(rng.integers(1, nv))
            cand = None
            if rng.uniform() < 0.4 and nv > 4:
                cand = np.sort(rng.choice(nv, int(rng.integers(2, nv)), replace=False)).astype(np.uint32)
                k = int(rng.integers(1, len(cand)))
            assert np.array_equal(dm.bow_select(q, k, cand), oracle_c.bow_select(bow, q, k, cand)), "bow_select"
    return 1


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    oracle_c.build()
    t0 = time.time()
    for s in range(first, first + n):
        try:
            one(s)
        except AssertionError as e:
            print(f"seed {s}: PARITY FAILURE: {e}", flush=True)
            raise
        if (s - first) % 20 == 19:
            print(f"{s - first + 1} cases, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {n} cases (dense gray image, BoF vector, three shortlists each), bit-exact ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
