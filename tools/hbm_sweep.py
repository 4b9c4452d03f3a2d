"""K1 at small N_q, where it IS HBM-bound (SURVEY.md F7 / 8d item iii): achieved GB/s of the bank read against
the measured streaming-read bandwidth of tools/peaks.  Bank = 8 M rows = 512 MB (larger than the 256 MiB
Infinity Cache).  Prints one JSON line per N_q."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402

rows = int(os.environ.get("ROWS", 8_000_000))
rng = np.random.Generator(np.random.PCG64(1))
bank = rng.integers(0, 256, size=(rows, 64), dtype=np.uint8)
nv = 4000
view_off = np.linspace(0, rows, nv + 1).astype(np.uint32)
m = S.Map(np.arange(nv, dtype=np.uint32), view_off, bank, params=S.default_params(profile=1))
for nq in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    q = m.query(synth.random_descriptors(rng, nq))
    for _ in range(3):
        m.match_putative(q)
    m.sync()
    m.stats_reset()
    reps = 20
    for _ in range(reps):
        m.match_putative(q)
    st = m.stats()
    ms = st.total_ms[0] / st.launches[0]
    gbs = (rows * 64 + nq * 64) / (ms * 1e-3) / 1e9
    print(json.dumps({"nq": nq, "rows": rows, "k1_ms": round(ms, 4), "bank_GBps": round(gbs, 1),
                      "pairs_per_s_T": round(rows * nq / (ms * 1e-3) / 1e12, 3),
                      "frac_of_8TBps": round(gbs / 8000, 3)}), flush=True)
    q.close()
m.close()
