"""GPU: K1/K2 (exact Hamming 2-NN + Lowe ratio + per-view compaction) through the C ABI against the C
oracle -- bit-exact, as SURVEY.md 8a demands for index work."""
import os

import numpy as np
import pytest

import sfmlocalization_amd as S
import synthdata as synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hamming_planted.npz")
INT_MAX = 2**31 - 1


def keys_from_oracle(j, d):
    k = (d.astype(np.int64) << 16) | (j.astype(np.int64) & 0xFFFF)
    k[j < 0] = 0xFFFFFFFF
    return k.astype(np.uint32)


def check_against_oracle(oracle_c, q, bank, view_off, view_sel, ratio=0.6):
    nv = len(view_off) - 1
    # default path (rows the screening kernel proves rejected are not finished) ...
    with S.Map(np.arange(nv, dtype=np.uint32) * 2 + 1, view_off, bank, params=S.default_params(dist_ratio=ratio)) as m:
        qq = m.query(q)
        m.match_putative(qq, view_sel)
        got_screened = m.putative_read()
        s0, s1 = m.putative_read_rows()      # pairs exist only for the rows the screening scan could not reject
        # run a second, different query on the same map first: stale partial-result slots must not leak through
        if len(q) > 1:
            q2 = m.query(q[::-1].copy())
            m.match_putative(q2, view_sel)
            m.match_putative(qq, view_sel)
            again = m.putative_read()
            for a, b in zip(got_screened, again):
                np.testing.assert_array_equal(a, b, err_msg="same query after another one on the same context")
            s0b, s1b = m.putative_read_rows()
            np.testing.assert_array_equal(s0, s0b)
            np.testing.assert_array_equal(s1, s1b)
            q2.close()
        qq.close()
    # ... and the exact-rows path, whose per-row (nearest, second) keys are compared too
    params = S.default_params(dist_ratio=ratio, exact_rows=1)
    with S.Map(np.arange(nv, dtype=np.uint32) * 2 + 1, view_off, bank, params=params) as m:
        qq = m.query(q)
        m.match_putative(qq, view_sel)
        got = m.putative_read()
        b0, b1 = m.putative_read_rows()
        qq.close()
    for name, a, b in zip(("view_count", "match_i", "match_j", "match_d"), got_screened, got):
        np.testing.assert_array_equal(a, b, err_msg="screened vs exact-rows: " + name)
    exp = oracle_c.match_to_query(q, bank, view_off, view_sel, ratio, threads=4)
    for name, a, b in zip(("view_count", "match_i", "match_j", "match_d"), got, exp):
        np.testing.assert_array_equal(a, b, err_msg=name)
    # per-row keys on the searched rows
    j0, d0, j1, d1 = oracle_c.hamming_2nn(q, bank, threads=4)
    searched = np.zeros(len(bank), bool)
    views = range(nv) if view_sel is None else view_sel
    for v in views:
        searched[int(view_off[v]):int(view_off[v + 1])] = True
    if len(q) > 0:
        np.testing.assert_array_equal(b0[searched], keys_from_oracle(j0, d0)[searched])
        np.testing.assert_array_equal(b1[searched], keys_from_oracle(j1, d1)[searched])
        # default path: a row either carries its exact pair or is marked "proved rejected"; accepted rows carry it
        have = s0 != S.NOMATCH
        np.testing.assert_array_equal(s0[have], b0[have])
        np.testing.assert_array_equal(s1[have], b1[have])
        acc = np.zeros(len(bank), bool)
        for v in views:
            o = int(view_off[v])
            acc[o + got[1][o:o + int(got[0][v])].astype(np.int64)] = True
        assert have[acc].all()
    return got


def test_golden_planted(oracle_c):
    g = np.load(GOLD)
    got = check_against_oracle(oracle_c, g["query"], g["bank"], g["view_off"], None)
    np.testing.assert_array_equal(got[0], g["view_count"])
    np.testing.assert_array_equal(got[1], g["match_i"])
    np.testing.assert_array_equal(got[2], g["match_j"])
    np.testing.assert_array_equal(got[3], g["match_d"])


@pytest.mark.parametrize("nq", [0, 1, 2, 63, 64, 65, 257, 767, 768, 769, 2000, 2049, 2400])
def test_query_sizes(oracle_c, nq):
    rng = np.random.Generator(np.random.PCG64(nq + 5))
    bank = synth.random_descriptors(rng, 700)
    q = synth.random_descriptors(rng, nq)
    if nq:
        bank[:200] = synth.flip_bits(rng, q[rng.integers(0, nq, 200)], 50)
    view_off = np.array([0, 64, 64, 129, 130, 700], np.uint32)  # block-aligned, empty, 1-row and long views
    check_against_oracle(oracle_c, q, bank, view_off, None)


def test_view_selection_and_query_split(oracle_c):
    m = synth.make_map(5, n_views=40, desc_per_view=300, views_per_place=8, landmarks_per_place=200,
                       obs_per_view=80, ragged=True)
    q = synth.make_query(m, 9, n_feat=1200, n_copies=120)
    sel = np.array([1, 2, 5, 17, 18, 19, 33, 39], np.uint32)
    got = check_against_oracle(oracle_c, q.desc, m.desc, m.view_off, sel)      # short list -> query split > 1
    assert got[0].sum() > 0
    check_against_oracle(oracle_c, q.desc, m.desc, m.view_off, np.array([7], np.uint32))
    check_against_oracle(oracle_c, q.desc, m.desc, m.view_off, None)
    check_against_oracle(oracle_c, q.desc, m.desc, m.view_off, None, ratio=0.8)


def test_large_bank_geometry_paths(oracle_c):
    # enough 64-row blocks to take the <4,16> kernel geometry (needs >= 2*CUs*64 wave-blocks)
    rng = np.random.Generator(np.random.PCG64(42))
    nq = 96
    n_rows = 64 * 33000 + 17
    q = synth.random_descriptors(rng, nq)
    bank = synth.random_descriptors(rng, n_rows)
    idx = rng.choice(n_rows, 5000, replace=False)
    bank[idx] = synth.flip_bits(rng, q[rng.integers(0, nq, 5000)], 40)
    view_off = np.linspace(0, n_rows, 1001).astype(np.uint32)
    check_against_oracle(oracle_c, q, bank, view_off, None)


@pytest.mark.parametrize("ratio", [0.3, 0.6, 0.95, 1.5])
def test_screening_on_structured_descriptors(oracle_c, ratio):
    """The screening kernel's threshold logic away from uniform random bits: sparse descriptors (small distances,
    large thresholds, the wave vote fires all the time), clusters of near-duplicates in the query (d1 = 0 and tiny
    second distances), ratios from strict to > 1 (everything with a finite ratio is accepted)."""
    rng = np.random.Generator(np.random.PCG64(int(ratio * 100)))
    nq, n_rows = 1500, 6000
    dense = synth.random_descriptors(rng, nq + n_rows)
    sparse = dense & synth.random_descriptors(rng, nq + n_rows) & synth.random_descriptors(rng, nq + n_rows)  # ~1/8 ones
    for src in (dense, sparse):
        q, bank = src[:nq].copy(), src[nq:].copy()
        q[100:140] = q[100]                                   # 40 identical query rows
        q[200:260] = synth.flip_bits(rng, np.repeat(q[200:201], 60, 0), 3)
        pick = rng.integers(0, nq, 1500)
        bank[:1500] = synth.flip_bits(rng, q[pick], 30)
        bank[1500:1600] = q[rng.integers(0, nq, 100)]        # exact copies: d0 = 0
        view_off = np.array([0, 1000, 1000, 2500, 6000], np.uint32)
        got = check_against_oracle(oracle_c, q, bank, view_off, None, ratio=ratio)
        if ratio >= 0.6 and src is dense:
            assert got[0].sum() > 500


def test_long_unaligned_views_through_the_screening_path(oracle_c):
    """Views longer than 64 blocks and not aligned to the 64-row blocks: K2 slides its window of row-mask words."""
    rng = np.random.Generator(np.random.PCG64(77))
    nq = 900
    n_rows = 9000 + 4500 + 70 + 5000
    q = synth.random_descriptors(rng, nq)
    bank = synth.random_descriptors(rng, n_rows)
    idx = rng.choice(n_rows, 2500, replace=False)
    bank[idx] = synth.flip_bits(rng, q[rng.integers(0, nq, 2500)], 35)
    view_off = np.array([0, 9000, 13500, 13570, n_rows], np.uint32)       # 9000 rows = 141 blocks, starts at 9000 % 64 != 0
    got = check_against_oracle(oracle_c, q, bank, view_off, None)
    assert got[0].min() > 0


def test_flagged_row_pass_paths(oracle_c):
    """The exact pass over flagged rows: the sliced path with its last-arrival merge (many chunks, several rounds per
    workgroup slot), the walking path when a query slice does not fit LDS (nq > 8192), and partial last chunks."""
    rng = np.random.Generator(np.random.PCG64(91))
    # (a) many flagged rows: 20 000 planted matches -> ~313 chunks over 128 chunk slots
    nq, n_rows = 1200, 40000
    q = synth.random_descriptors(rng, nq)
    bank = synth.random_descriptors(rng, n_rows)
    idx = rng.choice(n_rows, 20000, replace=False)
    bank[idx] = synth.flip_bits(rng, q[rng.integers(0, nq, 20000)], 30)
    view_off = np.linspace(0, n_rows, 21).astype(np.uint32)
    got = check_against_oracle(oracle_c, q, bank, view_off, None)
    assert got[0].sum() > 15000
    # (b) a query too long for one LDS slice per workgroup
    nq, n_rows = 9000, 3000
    q = synth.random_descriptors(rng, nq)
    bank = synth.random_descriptors(rng, n_rows)
    idx = rng.choice(n_rows, 700, replace=False)
    bank[idx] = synth.flip_bits(rng, q[rng.integers(0, nq, 700)], 30)
    check_against_oracle(oracle_c, q, bank, np.array([0, 1500, 3000], np.uint32), None)


def test_argument_errors():
    bank = np.zeros((10, 64), np.uint8)
    with pytest.raises(S.SfmlocError):
        S.Map([0, 0], [0, 5, 10], bank)          # view ids not ascending
    with pytest.raises(S.SfmlocError):
        S.Map([0, 1], [0, 5, 9], bank)           # view_off does not end at n_rows
    with S.Map([0, 1], [0, 5, 10], bank) as m:
        q = m.query(np.zeros((4, 64), np.uint8))
        with pytest.raises(S.SfmlocError):
            m.match_putative(q, np.array([1, 0], np.uint32))   # not ascending
        with pytest.raises(S.SfmlocError):
            m.match_putative(q, np.array([2], np.uint32))      # out of range
        with pytest.raises(S.SfmlocError):
            m.query(np.zeros((70000, 64), np.uint8))           # > 65535 query rows
        q.close()
