#!/usr/bin/env python3
"""BASELINE configs[4] size on ONE MI355X: a 50 k-image / 100 M-descriptor bank (6.4 GB) -- capacity and the K1 scan
rate at that size, with planted matches checked exactly (a sample of views against the oracle).
usage: scale_test.py [n_views] [rows_per_view]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402
from oracle import oracle_c  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    nq = 2000
    rng = np.random.Generator(np.random.PCG64(5))
    t0 = time.time()
    n = V * per
    bank = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    bank[:, 61:] = 0
    bank[:, 60] &= 0x3F
    q = synth.random_descriptors(rng, nq)
    # planted matches in a few views spread over the bank (first, middle, last)
    probe = [0, 1, V // 3, V // 2, V - 2, V - 1]
    for v in probe:
        rows = v * per + rng.choice(per, 200, replace=False)
        bank[rows] = synth.flip_bits(rng, q[rng.integers(0, nq, 200)], 30)
    view_off = (np.arange(V + 1, dtype=np.uint64) * per).astype(np.uint32)
    print(json.dumps({"rows": n, "bank_GB": n * 64 / 1e9, "host_build_s": round(time.time() - t0, 1)}), flush=True)
    t0 = time.time()
    dm = S.Map(np.arange(V, dtype=np.uint32), view_off, bank, params=S.default_params(profile=1))
    dq = dm.query(q)
    print(json.dumps({"map_create_s": round(time.time() - t0, 1)}), flush=True)
    dm.match_putative(dq)
    dm.sync()
    dm.stats_reset()
    for _ in range(3):
        dm.match_putative(dq)
    dm.sync()
    st = dm.stats()
    k1 = st.total_ms[0] / st.launches[0]
    cnt, mi, mj, md = dm.putative_read()
    ok = True
    for v in probe:
        a, b = v * per, (v + 1) * per
        e = oracle_c.match_to_query(q, bank[a:b], np.array([0, per], np.uint32), None, 0.6, threads=8)
        c = int(cnt[v])
        ok &= c == int(e[0][0]) and np.array_equal(mi[a:a + c], e[1][:c]) and np.array_equal(mj[a:a + c], e[2][:c])
    print(json.dumps({"k1_ms": k1, "pairs_per_s": n * nq / (k1 * 1e-3), "bank_GBps": n * 64 / (k1 * 1e-3) / 1e9,
                      "lane_ops_per_pair": st.hamming_lane_ops / st.hamming_pairs, "matches": int(cnt.sum()),
                      "views_with_matches": int((cnt > 0).sum()), "probe_views_exact": bool(ok)}), flush=True)
    assert ok
    dq.close()
    dm.close()


if __name__ == "__main__":
    main()
