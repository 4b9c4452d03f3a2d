#!/usr/bin/env python3
"""How the exact-screening Hamming kernel behaves on M-LDB descriptors extracted from images (GPU AKAZE on synthetic
textured scenes) instead of uniform random bits: issued lane-ops per pair, fraction of screened pairs finished,
kernel time against the exact kernel on the same bank."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402


def main():
    q, big, out = synth.mldb_like_bank(S)
    view_off = np.arange(0, len(big) + 1, len(big) // 200, dtype=np.uint32)
    view_off[-1] = len(big)
    nw = int(os.environ.get("SFMLOC_K1_SCREEN_NW", "10"))    # read once per process by the library
    for exact in (0, 1):
        p = S.default_params(profile=1, exact_rows=exact)
        with S.Map(np.arange(len(view_off) - 1, dtype=np.uint32), view_off, big, params=p) as dm:
            dq = dm.query(q)
            dm.match_putative(dq)
            dm.sync()
            dm.stats_reset()
            for _ in range(10):
                dm.match_putative(dq)
            dm.sync()
            st = dm.stats()
            cnt = dm.putative_read()[0]
            key = "exact_kernel" if exact else f"screening_nw{nw}"
            out[key] = {"k1_ms": st.total_ms[0] / st.launches[0], "lane_ops_per_pair": st.hamming_lane_ops / st.hamming_pairs,
                        "pairs_finished_frac": st.hamming_pairs_finished / st.hamming_pairs,
                        "rows_flagged_frac": st.hamming_rows_flagged / st.launches[0] / len(big),
                        "matches": int(cnt.sum())}
            dq.close()
    assert out["exact_kernel"]["matches"] == out[f"screening_nw{nw}"]["matches"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
