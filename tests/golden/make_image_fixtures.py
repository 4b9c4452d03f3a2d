"""Writes tests/golden/images/: small JPEG / PNG files and what libjpeg-turbo (through PIL) decodes them to.

    python tests/golden/make_image_fixtures.py

The expected arrays pin sfmloc_image_decode (sfmlocalization_amd/csrc/image_io.hip) to the library the reference's
cv::imread sits on: colour = JCS_RGB with libjpeg's defaults (islow IDCT, fancy upsampling), gray = JCS_GRAYSCALE
(PIL's draft mode), which is what imread(IMREAD_GRAYSCALE) asks for.  PNG is lossless, so PIL's RGB is the truth for
the colour read.  Needs Pillow; the fixtures are committed so the tests do not."""
import io
import os

import numpy as np
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 24
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "images")


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 5.0) * np.cos(y / 4.0), 128 + 90 * np.cos((x + y) / 3.0), 40 + 3.0 * y], -1)
    img += rng.normal(0, 10, img.shape)
    img[h // 3: h // 2, w // 4: w // 2] = (250, 10, 30)          # a saturated patch: exercises the range limits
    return np.clip(img, 0, 255).astype(np.uint8)


JPEGS = [  # name, (h, w), mode, save options
    ("base_420", (37, 53), "RGB", dict(quality=75, subsampling=2)),
    ("base_422", (24, 41), "RGB", dict(quality=90, subsampling=1)),
    ("base_444_rst", (48, 64), "RGB", dict(quality=85, subsampling=0, restart_marker_blocks=5)),
    ("prog_420", (45, 35), "RGB", dict(quality=60, subsampling=2, progressive=True)),
    ("prog_444_opt", (19, 23), "RGB", dict(quality=95, subsampling=0, progressive=True, optimize=True)),
    ("gray_base", (33, 17), "L", dict(quality=80)),
    ("gray_prog_rst", (40, 40), "L", dict(quality=50, progressive=True, restart_marker_rows=1)),
    ("narrow_420", (9, 3), "RGB", dict(quality=75, subsampling=2)),   # downsampled width 2: replication, not triangle
]


def main():
    os.makedirs(OUT, exist_ok=True)
    exp = {}
    for k, (name, (h, w), mode, opts) in enumerate(JPEGS):
        img = picture(h, w, k)
        im = Image.fromarray(img if mode == "RGB" else img[:, :, 0])
        b = io.BytesIO()
        im.save(b, "JPEG", **opts)
        data = b.getvalue()
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(data)
        exp[name + "_bgr"] = np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[:, :, ::-1])
        g = Image.open(io.BytesIO(data))
        g.draft("L", g.size)
        exp[name + "_gray"] = np.asarray(g.convert("L"))
    img = picture(21, 30, 99)
    rgba = np.dstack([img, np.full(img.shape[:2], 200, np.uint8)])
    pal = Image.fromarray(img).quantize(16)
    for name, im in (("rgb", Image.fromarray(img)), ("rgba", Image.fromarray(rgba)), ("gray", Image.fromarray(img[:, :, 1])),
                     ("palette", pal), ("gray_alpha", Image.fromarray(np.dstack([img[:, :, 2], rgba[:, :, 3]]), "LA"))):
        path = os.path.join(OUT, name + ".png")
        im.save(path, "PNG")
        exp["png_" + name + "_bgr"] = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **exp)
    print("wrote", len(exp), "arrays,", sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)), "bytes")


if __name__ == "__main__":
    main()
