"""Map-side matching through the C ABI: sfmloc_match_pairs / sfmloc_track against the literal
restatements of hulo::matchAKAZE / trackAKAZE (MatchUtils.cpp:73-277) in oracle/pipeline.py.  Bit-exact."""
import numpy as np
import pytest

import sfmlocalization_amd as S
import synthdata as synth
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu


def video(seed, n_views=12, ragged=False):
    m = synth.make_map(seed, n_views=n_views, desc_per_view=260, views_per_place=n_views, landmarks_per_place=220,
                       obs_per_view=150, map_flips=6, ragged=ragged)
    desc = m.desc.copy()
    off = m.view_off.astype(np.int64)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    for v in range(n_views - 1):
        n = int(off[v + 1] - off[v])
        if n < 8:
            continue
        lm_rows = np.nonzero(m.row_landmark[off[v]:off[v + 1]] >= 0)[0]
        if len(lm_rows) < 6:
            continue
        # two rows of frame v that hit the same row of frame v+1 -> the one-to-one filter drops both
        a, b = rng.choice(lm_rows, 2, replace=False)
        desc[off[v] + b] = desc[off[v] + a]
        # the last row of frame v gets a real match: it must not be emitted, but it still takes part in the filter
        c = rng.choice(lm_rows)
        desc[off[v + 1] - 1] = desc[off[v] + c]
    descs = [desc[off[v]:off[v + 1]] for v in range(n_views)]
    return m, desc, descs


def assert_same(got, exp):
    assert list(got.keys()) == list(exp.keys())
    for k in exp:
        np.testing.assert_array_equal(got[k][0], exp[k][0], err_msg=f"pair {k}: i")
        np.testing.assert_array_equal(got[k][1], exp[k][1], err_msg=f"pair {k}: j")


@pytest.mark.parametrize("ragged", [False, True])
def test_track_equals_reference_restatement(oracle_c, ragged):
    m, desc, descs = video(41, ragged=ragged)
    with S.Map(m.view_id, m.view_off, desc) as dm:
        for dist in (2, 4, 100):
            got = dm.track(dist)
            exp = opipe.track_akaze(descs, dist, 0.6)
            assert_same(got, exp)
            assert sum(len(v[0]) for v in exp.values()) > 200       # the scene really tracks
        # the filter did something: plain putative lists are longer than the one-to-one ones
        q = dm.query_from_view(1)
        dm.match_putative(q, np.array([0], np.uint32))
        n_put = int(dm.putative_read()[0][0])
        dm.match_one_to_one(q, np.array([0], np.uint32))
        cnt, mi, mj, _ = dm.putative_read()
        assert int(cnt[0]) < n_put
        e_i, e_j = opipe.match_akaze_pair(descs[0], descs[1], 0.6)
        np.testing.assert_array_equal(mi[:cnt[0]], e_i)
        np.testing.assert_array_equal(mj[:cnt[0]], e_j)
        q.close()


def test_pair_list_equals_reference_restatement(oracle_c):
    m, desc, descs = video(42, n_views=10)
    pairs = [(0, 5), (3, 1), (2, 2), (7, 5), (0, 5), (9, 0), (4, 5), (1, 3)]
    with S.Map(m.view_id, m.view_off, desc) as dm:
        got = dm.match_pairs(pairs)
        exp = opipe.match_akaze(descs, sorted(set(pairs)), 0.6)
        assert_same(got, exp)
        assert (2, 2) in got and len(got[(2, 2)][0]) > 0


def test_geometric_match_of_map_pairs(oracle_c):
    """sfmloc_geometric_pairs = hulo::geometricMatch on tracked pairs (incl. chained ones and the minMatch filter
    applied by the caller), against the oracle's F-matrix AC-RANSAC on the same lists."""
    m, desc, descs = video(43, n_views=8)
    off = m.view_off.astype(np.int64)
    kp = synth.round6(m.kpt_xy)
    for rounds in (25, 300):
        p = S.default_params(ransac_round=rounds, geom_precision=4.0)
        with S.Map(m.view_id, m.view_off, desc, params=p, view_wh=m.view_wh, kpt_xy=kp) as dm:
            put = dm.track(3)
            put = {k: v for k, v in put.items() if len(v[0]) >= 20}         # minMatch
            assert len(put) >= 8 and any(b - a == 2 for a, b in put)
            got = dm.geometric_pairs(put)
        exp = opipe.geometric_match([kp[off[v]:off[v + 1]] for v in range(8)], m.view_wh, m.view_id, put,
                                    ransac_round=rounds)
        assert_same(got, exp)
        assert len(exp) >= 6 and all(len(v[0]) > 17 for v in exp.values())


def test_fmatrix_golden_scene_on_device():
    """The committed fundamental-matrix scene (tests/golden/geometry_scenes.npz) through sfmloc_geometric_pairs: the
    device reproduces the minted inlier list exactly."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry_scenes.npz"))
    x1, x2 = g["f_x1"].astype(np.float32), g["f_x2"].astype(np.float32)
    assert np.array_equal(x1.astype(np.float64), g["f_x1"])          # the scene is float32-exact by construction
    n = len(x1)
    rng = np.random.Generator(np.random.PCG64(2))
    desc = synth.random_descriptors(rng, 2 * n)
    w, h = (int(v) for v in g["wh"])
    p = S.default_params(ransac_round=200, geom_precision=4.0)
    with S.Map(np.array([7, 9], np.uint32), np.array([0, n, 2 * n], np.uint32), desc, params=p,
               view_wh=np.array([[w, h], [w, h]], np.uint32), kpt_xy=np.concatenate([x1, x2])) as dm:
        ident = np.arange(n, dtype=np.uint32)
        got = dm.geometric_pairs({(0, 1): (ident, ident)})
    assert list(got) == [(0, 1)]
    np.testing.assert_array_equal(got[(0, 1)][0], g["f_inliers"])
    np.testing.assert_array_equal(got[(0, 1)][1], g["f_inliers"])
