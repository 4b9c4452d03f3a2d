#!/usr/bin/env python3
"""Randomised campaign for the JPEG decoder (sfmloc_image_decode) against libjpeg-turbo through PIL -- the one stage
of this repository with a real third-party reference in the image: random sizes, contents, qualities, chroma
subsampling, baseline / progressive / optimised Huffman tables, restart intervals; colour (JCS_RGB, fancy upsampling)
and gray (JCS_GRAYSCALE) output bit for bit.  Host code: runs without a GPU.  usage: fuzz_jpeg.py [n] [first_seed]"""
import io
import os
import sys
import time

import numpy as np
from PIL import Image, ImageFile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sfmlocalization_amd import capi  # noqa: E402

ImageFile.MAXBLOCK = 1 << 24


def one(seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    kind = rng.integers(0, 3)
    y, x = np.mgrid[0:h, 0:w]
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        img = np.stack([128 + 110 * np.sin(x / rng.uniform(2, 30)) * np.cos(y / rng.uniform(2, 30)),
                        (x * rng.integers(1, 9) + y * rng.integers(1, 9)) % 256, 255 - x - y], -1) + rng.normal(0, 10, (h, w, 3))
    else:
        img = np.zeros((h, w, 3)) + rng.integers(0, 256, 3)
        img[h // 3:, w // 4:] = rng.integers(0, 256, 3)          # hard edges, saturated colours
    img = np.clip(img, 0, 255).astype(np.uint8)
    gray_mode = rng.uniform() < 0.25
    kw = {"quality": int(rng.integers(1, 101))}
    if not gray_mode:
        kw["subsampling"] = int(rng.integers(0, 3))
    if rng.uniform() < 0.4:
        kw["progressive"] = True
    if rng.uniform() < 0.4:
        kw["optimize"] = True
    r = rng.uniform()
    if r < 0.25:
        kw["restart_marker_blocks"] = int(rng.integers(1, 20))
    elif r < 0.4:
        kw["restart_marker_rows"] = int(rng.integers(1, 4))
    b = io.BytesIO()
    Image.fromarray(img[:, :, 1] if gray_mode else img).save(b, "JPEG", **kw)
    data = b.getvalue()
    ref_rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    g = Image.open(io.BytesIO(data))
    g.draft("L", g.size)
    ref_gray = np.asarray(g.convert("L"))
    assert np.array_equal(capi.image_decode(data, True)[:, :, ::-1], ref_rgb), f"colour {h}x{w} {kw}"
    assert np.array_equal(capi.image_decode(data, False), ref_gray), f"gray {h}x{w} {kw}"
    return h * w


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    px = 0
    for s in range(first, first + n):
        try:
            px += one(s)
        except AssertionError as e:
            print(f"seed {s}: MISMATCH: {e}", flush=True)
            raise
    print(f"OK: {n} JPEG files, {px} pixels, colour and gray equal libjpeg-turbo's bit for bit ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
