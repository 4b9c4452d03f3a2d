"""NumPy twin of the integer stages of the oracle (independent second implementation).

TEST INFRASTRUCTURE ONLY.  Used to cross-check sfm_oracle.c and to mint the golden fixtures under
tests/golden/ (the reference holds none for this path -- parity unpinned).
"""
import numpy as np

_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint16)
INT_MAX = 2**31 - 1


def hamming_matrix(a, b):
    """[len(a), len(b)] Hamming distances over all 64 stored bytes (FileUtils.cpp:94-103 rows)."""
    a = np.asarray(a, dtype=np.uint8).reshape(-1, 64)
    b = np.asarray(b, dtype=np.uint8).reshape(-1, 64)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.int32)
    step = max(1, (1 << 22) // max(1, b.shape[0] * 64))
    for s in range(0, a.shape[0], step):
        x = a[s:s + step, None, :] ^ b[None, :, :]
        out[s:s + step] = _POP8[x].sum(axis=2, dtype=np.int32)
    return out


def hamming_2nn(query, bank):
    """Exact 2-NN of each bank row among the query rows; ties -> lowest query index
    (MatchUtils.cpp:339-340 made exact).  Missing neighbours: j = -1, d = INT_MAX."""
    query = np.asarray(query, dtype=np.uint8).reshape(-1, 64)
    bank = np.asarray(bank, dtype=np.uint8).reshape(-1, 64)
    n, nq = bank.shape[0], query.shape[0]
    j0 = np.full(n, -1, np.int32)
    d0 = np.full(n, INT_MAX, np.int32)
    j1 = np.full(n, -1, np.int32)
    d1 = np.full(n, INT_MAX, np.int32)
    if nq == 0 or n == 0:
        return j0, d0, j1, d1
    D = hamming_matrix(bank, query).astype(np.int64)
    key = D * 65536 + np.arange(nq, dtype=np.int64)[None, :]  # (distance, index) lexicographic
    order = np.sort(key, axis=1)
    j0[:] = order[:, 0] % 65536
    d0[:] = order[:, 0] // 65536
    if nq >= 2:
        j1[:] = order[:, 1] % 65536
        d1[:] = order[:, 1] // 65536
    return j0, d0, j1, d1


def ratio_accept(d0, d1, ratio):
    """(0.0f + d0) / d1 < ratio in float32, and d1 < INT_MAX (MatchUtils.cpp:347-349)."""
    d0 = np.asarray(d0)
    d1 = np.asarray(d1)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (np.float32(0.0) + d0.astype(np.float32)) / d1.astype(np.float32)
        return (q < np.float32(ratio)) & (d1 < INT_MAX)


def match_to_query(query, bank, view_off, view_sel=None, ratio=0.6):
    """matchAKAZEToQuery (MatchUtils.cpp:283-367): per selected view the ordered list of (i, j0, d0)."""
    bank = np.asarray(bank, dtype=np.uint8).reshape(-1, 64)
    view_off = np.asarray(view_off, dtype=np.int64)
    nv = len(view_off) - 1
    cnt = np.zeros(nv, np.uint32)
    n = bank.shape[0]
    mi = np.full(n, 0xFFFFFFFF, np.uint32)
    mj = np.full(n, 0xFFFFFFFF, np.uint32)
    md = np.full(n, 0xFFFFFFFF, np.uint32)
    nq = np.asarray(query).reshape(-1, 64).shape[0]
    if nq < 1:
        return cnt, mi, mj, md
    views = range(nv) if view_sel is None else [int(v) for v in view_sel]
    for v in views:
        off, end = int(view_off[v]), int(view_off[v + 1])
        if end == off:
            continue
        j0, d0, _, d1 = hamming_2nn(query, bank[off:end])
        acc = np.nonzero(ratio_accept(d0, d1, ratio))[0]
        c = len(acc)
        cnt[v] = c
        mi[off:off + c] = acc
        mj[off:off + c] = j0[acc]
        md[off:off + c] = d0[acc]
    return cnt, mi, mj, md
