#!/usr/bin/env python3
"""Randomised AKAZE + M-LDB parity campaign: random textures at random sizes (odd widths, tile edges of the fused
kernels), random thresholds and octave / sub-level counts, GPU against the CPU restatement -- scale space, determinant
response, keypoints and descriptors bit for bit.  usage: fuzz_akaze.py [n_images] [first_seed]
SFMLOC_FUZZ_BATCH=1: every seed is a BATCH of 2..6 textures of one random size taken by
sfmloc_akaze_detect_and_compute_batch (one launch per kernel for the batch), each frame against the oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402
from oracle import oracle_c  # noqa: E402


def bits32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def one(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    h = int(rng.integers(64, 420))
    w = int(rng.integers(64, 560))
    g = synth.texture_image(seed, h, w, n_blobs=int(rng.integers(20, 400)), n_rects=int(rng.integers(5, 80)))
    if rng.uniform() < 0.3:
        g = np.clip(g.astype(np.int32) + rng.integers(-12, 13, g.shape), 0, 255).astype(np.uint8)
    ekp, edesc, eldet, elt = oracle_c.akaze_detect_and_compute(g, want_levels=True)
    ak = S.Akaze(w, h)
    try:
        assert [tuple(x) for x in oracle_c.akaze_levels(w, h)] == ak.levels, "level sizes"
        kp, desc = ak.detect_and_compute(g)
        ldet, lt = ak.read_levels()
        assert np.array_equal(bits32(lt), bits32(elt)), "scale space"
        assert np.array_equal(bits32(ldet), bits32(eldet)), "determinant response"
        assert len(kp) == len(ekp) and np.array_equal(bits32(kp), bits32(ekp)), "keypoints"
        assert np.array_equal(desc[:, :61], edesc), "descriptors"
    finally:
        ak.close()
    return len(kp)


def one_batch(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    h = int(rng.integers(64, 420))
    w = int(rng.integers(64, 560))
    k = int(rng.integers(2, 7))
    imgs = [synth.texture_image(seed * 8 + j, h, w, n_blobs=int(rng.integers(20, 400)), n_rects=int(rng.integers(5, 80)))
            for j in range(k)]
    exs = [S.Akaze(w, h) for _ in range(k)]
    try:
        got = S.Akaze.detect_and_compute_batch(exs, imgs)
        total = 0
        for j, (g, e, (kp, desc)) in enumerate(zip(imgs, exs, got)):
            ekp, edesc, eldet, elt = oracle_c.akaze_detect_and_compute(g, want_levels=True)
            ldet, lt = e.read_levels()
            assert np.array_equal(bits32(lt), bits32(elt)), f"scale space (frame {j} of {k})"
            assert np.array_equal(bits32(ldet), bits32(eldet)), f"determinant response (frame {j} of {k})"
            assert len(kp) == len(ekp) and np.array_equal(bits32(kp), bits32(ekp)), f"keypoints (frame {j} of {k})"
            assert np.array_equal(desc[:, :61], edesc), f"descriptors (frame {j} of {k})"
            total += len(kp)
    finally:
        for e in exs:
            e.close()
    return total


def main():
    global one
    if os.environ.get("SFMLOC_FUZZ_BATCH") == "1":
        one = one_batch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    oracle_c.build()
    t0 = time.time()
    total = 0
    for s in range(first, first + n):
        try:
            total += one(s)
        except AssertionError as e:
            print(f"seed {s}: PARITY FAILURE: {e}", flush=True)
            raise
        if (s - first) % 10 == 9:
            print(f"{s - first + 1} images, {total} keypoints compared, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {n} images, {total} keypoints, every stage bit-exact ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
