"""CPU: the C oracle (oracle/sfm_oracle.c) against its NumPy twin and the golden fixture.

The reference holds no vectors for this path (parity unpinned); the fixture is minted by
tests/golden/make_golden.py from the NumPy twin.
"""
import os

import numpy as np
import pytest

from oracle import oracle_np as onp
import synthdata as synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hamming_planted.npz")
INT_MAX = 2**31 - 1


def test_golden_fixture_matches_c_oracle(oracle_c):
    g = np.load(GOLD)
    j0, d0, j1, d1 = oracle_c.hamming_2nn(g["query"], g["bank"])
    np.testing.assert_array_equal(j0, g["j0"])
    np.testing.assert_array_equal(d0, g["d0"])
    np.testing.assert_array_equal(d1, g["d1"])
    np.testing.assert_array_equal(j1, g["j1"])
    cnt, mi, mj, md = oracle_c.match_to_query(g["query"], g["bank"], g["view_off"], None, 0.6)
    np.testing.assert_array_equal(cnt, g["view_count"])
    np.testing.assert_array_equal(mi, g["match_i"])
    np.testing.assert_array_equal(mj, g["match_j"])
    np.testing.assert_array_equal(md, g["match_d"])


def test_golden_has_the_planted_cases():
    g = np.load(GOLD)
    d0, d1, j0 = g["d0"], g["d1"], g["j0"]
    # planted near-duplicates at known distances: first 64 rows, 8 per distance
    for k, dist in enumerate((0, 1, 5, 17, 40, 80, 120, 150)):
        assert (d0[8 * k:8 * k + 8] <= dist).all()
    assert (d0[:16] == np.repeat([0, 1], 8)).all()
    # tie rows (query rows 3, 10, 50 identical): nearest is the LOWEST index, d0 == d1
    assert list(j0[64:67]) == [3, 3, 3]
    assert list(g["j1"][64:67]) == [10, 10, 10]
    assert (d0[64:67] == d1[64:67]).all()
    assert d1[64] == 0 and not g["accept06"][64]  # 0/0 = NaN -> rejected (MatchUtils.cpp:347)
    # equidistant row between query 20 and 21
    assert d0[67] == 10 and d1[67] == 10 and j0[67] == 20 and g["j1"][67] == 21


@pytest.mark.parametrize("nq", [0, 1, 2, 3, 64, 65, 200])
def test_c_vs_numpy_random(oracle_c, nq):
    rng = np.random.Generator(np.random.PCG64(100 + nq))
    q = synth.random_descriptors(rng, nq)
    bank = synth.random_descriptors(rng, 333)
    if nq:
        bank[:40] = synth.flip_bits(rng, q[rng.integers(0, nq, 40)], 60)
    a = oracle_c.hamming_2nn(q, bank, threads=2)
    b = onp.hamming_2nn(q, bank)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    if nq == 1:
        assert (a[3] == INT_MAX).all() and (a[2] == -1).all()


def test_ratio_expression_float32(oracle_c):
    # the expression of MatchUtils.cpp:347-349, float32: boundary cases around 0.6
    for d0 in range(0, 513, 7):
        for d1 in list(range(0, 513, 5)) + [INT_MAX]:
            assert oracle_c.ratio_accept(d0, d1, 0.6) == bool(onp.ratio_accept(np.int32(d0), np.int32(d1), 0.6))
    assert not oracle_c.ratio_accept(0, 0, 0.6)          # NaN
    assert oracle_c.ratio_accept(0, 1, 0.6)
    assert not oracle_c.ratio_accept(3, 5, 0.6)          # 0.6f < 0.6f is false
    assert oracle_c.ratio_accept(299, 500, 0.6)
    assert not oracle_c.ratio_accept(5, INT_MAX, 0.6)    # no second neighbour


def test_match_to_query_views(oracle_c):
    rng = np.random.Generator(np.random.PCG64(7))
    m = synth.make_map(3, n_views=12, desc_per_view=150, views_per_place=4, landmarks_per_place=120,
                       obs_per_view=60, ragged=True)
    q = synth.make_query(m, 11, n_feat=180, n_copies=60)
    sel = np.array([0, 2, 3, 7, 11], dtype=np.uint32)
    for view_sel in (None, sel):
        a = oracle_c.match_to_query(q.desc, m.desc, m.view_off, view_sel, 0.6, threads=3)
        b = onp.match_to_query(q.desc, m.desc, m.view_off, view_sel, 0.6)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    cnt = a[0]
    assert cnt.sum() > 0 and (cnt[[1, 4, 5, 6, 8, 9, 10]] == 0).all()
    # lists are in ascending map-feature index (MatchUtils.cpp:346 loop order)
    for v in sel:
        off = int(m.view_off[v])
        lst = a[1][off:off + int(cnt[v])]
        assert (np.diff(lst.astype(np.int64)) > 0).all()
    del rng


def test_match_akaze_pair_one_to_one_and_last_row():
    """Literal restatement of MatchUtils.cpp:99-150 on a handcrafted pair: rows 1 and 2 hit the same train row
    (both dropped), the last row has a perfect match but is never emitted."""
    from oracle import oracle_c, pipeline as opipe
    oracle_c.build()
    rng = np.random.Generator(np.random.PCG64(3))
    d2 = rng.integers(0, 256, (6, 64), dtype=np.uint8)
    d2[:, 61:] = 0
    d1 = np.stack([d2[0], d2[1], d2[1], d2[3], rng.integers(0, 256, 64, dtype=np.uint8), d2[5]])
    d1[:, 61:] = 0
    mi, mj = opipe.match_akaze_pair(d1, d2)
    assert mi.tolist() == [0, 3] and mj.tolist() == [0, 3]
    assert opipe.match_akaze_pair(d1[:1], d2)[0].size == 0          # fewer than 2 rows: skipped (:101-103)
    tr = opipe.track_akaze([d1, d2, d2], 3)
    assert list(tr) == [(0, 1), (0, 2), (1, 2)]
    assert tr[(1, 2)][0].tolist() == [0, 1, 2, 3, 4] and tr[(0, 2)][0].tolist() == [0, 3]
