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
from sfmlocalization_amd import synth  # noqa: E402


def main():
    ak = S.Akaze(640, 480)
    descs = []
    for seed in range(48):
        img = synth.texture_image(100 + seed, 480, 640, n_blobs=900, n_rects=160)
        kp, d = ak.detect_and_compute(img)
        descs.append(d)
    ak.close()
    n_per = [len(d) for d in descs]
    q = np.concatenate(descs[:8])[:2000]           # a query-sized set of real descriptors (8 images' worth)
    bank_imgs = descs[8:]
    bank = np.concatenate(bank_imgs)
    reps = max(1, 400000 // len(bank))
    rng = np.random.Generator(np.random.PCG64(1))
    big = np.concatenate([synth.flip_bits(rng, bank, 6 * r) if r else bank for r in range(reps)])
    view_off = np.arange(0, len(big) + 1, len(big) // 200, dtype=np.uint32)
    view_off[-1] = len(big)
    # distance statistics between unrelated descriptors
    a = np.unpackbits(q[:256], axis=1).astype(np.int32)
    b = np.unpackbits(bank[:2048], axis=1).astype(np.int32)
    dist = (a[:, None, :] != b[None, :, :]).sum(2)
    out = {"images": len(descs), "desc_per_image_mean": float(np.mean(n_per)), "nq": int(len(q)), "rows": int(len(big)),
           "pair_distance_mean": float(dist.mean()), "pair_distance_std": float(dist.std()),
           "bits_set_mean": float(np.unpackbits(bank, axis=1).sum(1).mean())}
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
