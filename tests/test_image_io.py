"""sfmloc_image_decode / sfmloc_image_read = cv::imread of the query image (AKAZEOpenCV.cpp:60 IMREAD_GRAYSCALE,
DenseLocalFeatureWrapper.cpp:85 IMREAD_COLOR).  Host code of the C-ABI library, so these run without a GPU.

Pinned: JPEG against libjpeg-turbo (the committed fixtures of tests/golden/make_image_fixtures.py, and live through
PIL when Pillow is importable), bit for bit, in colour (JCS_RGB, fancy upsampling) and gray (JCS_GRAYSCALE = Y plane);
PNG against libpng 1.6 itself (libpng16.so.16 is in the image; driven through ctypes exactly as OpenCV's decoder drives
it), colour and gray, every accepted colour type and bit depth."""
import io
import os

import numpy as np
import pytest

import sfmlocalization_amd as S
from sfmlocalization_amd import capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "images")
JPEG_NAMES = ["base_420", "base_422", "base_444_rst", "prog_420", "prog_444_opt", "gray_base", "gray_prog_rst",
              "narrow_420"]
PNG_NAMES = ["rgb", "rgba", "gray", "palette", "gray_alpha"]


def png_gray(bgr):
    b, g, r = (bgr[:, :, i].astype(np.int64) for i in range(3))
    out = (9797 * r + 19234 * g + 3737 * b) >> 15          # libpng truncates (png_do_rgb_to_gray)
    same = (r == g) & (r == b)
    return np.where(same, r, out).astype(np.uint8)


def libpng_read(path, color):
    """What OpenCV's PNG decoder does (grfmt_png.cpp PngDecoder::readData), executed by libpng itself through ctypes:
    strip 16 -> 8 bits and alpha, palette -> RGB, low-bit gray -> 8 bits, then png_set_bgr (colour out of colour),
    png_set_gray_to_rgb (colour out of gray) or png_set_rgb_to_gray(1, 0.299, 0.587) (gray)."""
    import ctypes as C
    png = C.CDLL("libpng16.so.16")
    libc = C.CDLL(None)
    libc.fopen.restype, libc.fopen.argtypes = C.c_void_p, [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    png.png_create_read_struct.restype = C.c_void_p
    png.png_create_read_struct.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    png.png_create_info_struct.restype, png.png_create_info_struct.argtypes = C.c_void_p, [C.c_void_p]
    png.png_init_io.argtypes = [C.c_void_p, C.c_void_p]
    png.png_read_info.argtypes = [C.c_void_p, C.c_void_p]
    for f in ("png_get_image_width", "png_get_image_height"):
        getattr(png, f).restype, getattr(png, f).argtypes = C.c_uint32, [C.c_void_p, C.c_void_p]
    for f in ("png_get_color_type", "png_get_bit_depth"):
        getattr(png, f).restype, getattr(png, f).argtypes = C.c_ubyte, [C.c_void_p, C.c_void_p]
    png.png_set_rgb_to_gray.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    for f in ("png_set_strip_alpha", "png_set_palette_to_rgb", "png_set_expand_gray_1_2_4_to_8", "png_set_strip_16",
              "png_set_bgr", "png_set_gray_to_rgb"):
        getattr(png, f).argtypes = [C.c_void_p]
    png.png_read_update_info.argtypes = [C.c_void_p, C.c_void_p]
    png.png_get_rowbytes.restype, png.png_get_rowbytes.argtypes = C.c_size_t, [C.c_void_p, C.c_void_p]
    png.png_read_image.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    png.png_read_end.argtypes = [C.c_void_p, C.c_void_p]
    png.png_destroy_read_struct.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]
    fp = libc.fopen(os.fsencode(path), b"rb")
    assert fp
    p = C.c_void_p(png.png_create_read_struct(b"1.6.37", None, None, None))
    info = C.c_void_p(png.png_create_info_struct(p))
    png.png_init_io(p, fp)
    png.png_read_info(p, info)
    w, h = png.png_get_image_width(p, info), png.png_get_image_height(p, info)
    ct, bd = png.png_get_color_type(p, info), png.png_get_bit_depth(p, info)
    if bd == 16:
        png.png_set_strip_16(p)
    png.png_set_strip_alpha(p)
    if ct == 3:
        png.png_set_palette_to_rgb(p)
    if ct == 0 and bd < 8:
        png.png_set_expand_gray_1_2_4_to_8(p)
    if (ct & 2 or ct == 3) and color:
        png.png_set_bgr(p)
    elif color:
        png.png_set_gray_to_rgb(p)
    else:
        png.png_set_rgb_to_gray(p, 1, 0.299, 0.587)
    png.png_read_update_info(p, info)
    rb = png.png_get_rowbytes(p, info)
    buf = np.zeros((h, rb), np.uint8)
    rows = (C.c_void_p * h)(*[buf.ctypes.data + i * rb for i in range(h)])
    png.png_read_image(p, rows)
    png.png_read_end(p, None)
    png.png_destroy_read_struct(C.byref(p), C.byref(info), None)
    libc.fclose(fp)
    return buf.reshape(h, -1, 3)[:, :w] if color else buf[:, :w]


def test_png_equals_libpng_itself(tmp_path):
    """The PNG reader against libpng 1.6 (the library under OpenCV's decoder), driven the way grfmt_png.cpp drives it:
    every colour type and bit depth the reader accepts, gray and colour output, bit for bit -- including libpng's
    truncating RGB -> gray."""
    import ctypes
    try:
        ctypes.CDLL("libpng16.so.16")
    except OSError:
        pytest.skip("libpng16 is not in this image")
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    img[5:9, :, 1] = img[5:9, :, 0]
    img[5:9, :, 2] = img[5:9, :, 0]                                   # exact grays inside a colour image
    alpha = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    cases = {"rgb": Image.fromarray(img), "rgba": Image.fromarray(np.dstack([img, alpha])),
             "gray": Image.fromarray(img[:, :, 0]), "la": Image.fromarray(np.dstack([img[:, :, 0], alpha]), "LA"),
             "pal256": Image.fromarray(img).quantize(256), "pal16": Image.fromarray(img).quantize(16),
             "pal2": Image.fromarray(img).quantize(2), "bilevel": Image.fromarray(img[:, :, 0]).convert("1"),
             "gray16": Image.fromarray((img[:, :, 0].astype(np.uint16) * 257 + 13))}
    for name, im in cases.items():
        path = str(tmp_path / (name + ".png"))
        im.save(path, "PNG")
        for color in (False, True):
            np.testing.assert_array_equal(capi.image_read(path, color), libpng_read(path, color), err_msg=f"{name} {color}")
    for name in PNG_NAMES:
        path = os.path.join(GOLD, name + ".png")
        for color in (False, True):
            np.testing.assert_array_equal(capi.image_read(path, color), libpng_read(path, color), err_msg=name)


@pytest.mark.parametrize("name", JPEG_NAMES)
def test_jpeg_fixtures_equal_libjpeg(name):
    exp = np.load(os.path.join(GOLD, "expected.npz"))
    path = os.path.join(GOLD, name + ".jpg")
    bgr = capi.image_read(path, color=True)
    gray = capi.image_read(path, color=False)
    np.testing.assert_array_equal(bgr, exp[name + "_bgr"])
    np.testing.assert_array_equal(gray, exp[name + "_gray"])
    assert capi.image_size(path) == (gray.shape[1], gray.shape[0])
    with open(path, "rb") as f:
        np.testing.assert_array_equal(capi.image_decode(f.read(), color=True), bgr)


@pytest.mark.parametrize("name", PNG_NAMES)
def test_png_fixtures(name):
    exp = np.load(os.path.join(GOLD, "expected.npz"))
    path = os.path.join(GOLD, name + ".png")
    bgr = capi.image_read(path, color=True)
    np.testing.assert_array_equal(bgr, exp["png_" + name + "_bgr"])
    np.testing.assert_array_equal(capi.image_read(path, color=False), png_gray(bgr))
    assert capi.image_size(path) == (bgr.shape[1], bgr.shape[0])


def test_pnm_and_errors(tmp_path):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    p6 = tmp_path / "a.ppm"
    p6.write_bytes(b"P6\n# comment\n9 7\n255\n" + img.tobytes())
    np.testing.assert_array_equal(capi.image_read(str(p6), color=True), img[:, :, ::-1])
    r, g, b = (img[:, :, i].astype(np.int64) for i in range(3))
    np.testing.assert_array_equal(capi.image_read(str(p6), color=False),
                                  ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8))   # cvtColor BGR2GRAY
    p5 = tmp_path / "a.pgm"
    p5.write_bytes(b"P5 9 7 255\n" + img[:, :, 0].tobytes())
    np.testing.assert_array_equal(capi.image_read(str(p5)), img[:, :, 0])
    np.testing.assert_array_equal(capi.image_read(str(p5), color=True), np.repeat(img[:, :, :1], 3, axis=2))
    # errors: missing file, not an image, truncated JPEG header, arithmetic-coded frame, caller buffer too small
    with pytest.raises(S.SfmlocError) as e:
        capi.image_read(str(tmp_path / "nope.jpg"))
    assert e.value.code == capi.EIO
    with pytest.raises(S.SfmlocError):
        capi.image_decode(b"hello world, this is not an image at all....")
    with open(os.path.join(GOLD, "base_420.jpg"), "rb") as f:
        data = f.read()
    with pytest.raises(S.SfmlocError):
        capi.image_decode(data[:40])
    sof = data.index(b"\xff\xc0")
    with pytest.raises(S.SfmlocError) as e:
        capi.image_decode(data[:sof] + b"\xff\xc9" + data[sof + 2:])
    assert "arithmetic" in str(e.value)
    import ctypes as C
    L = capi._L()
    w, h = C.c_int32(0), C.c_int32(0)
    raw = np.frombuffer(data, np.uint8)
    small = np.zeros(10, np.uint8)
    rc = L.sfmloc_image_decode(capi._ptr(raw, C.c_uint8), raw.size, 0, capi._ptr(small, C.c_uint8), small.size,
                               C.byref(w), C.byref(h))
    assert rc == capi.ECAP and (w.value, h.value) == (53, 37)
    # a scan cut short still decodes (libjpeg pads with zeros and warns); the size must be right and nothing may crash
    cut = capi.image_decode(data[: len(data) * 2 // 3] + b"\xff\xd9")
    assert cut.shape == (37, 53)


def test_jpeg_live_against_pil():
    """Every mode PIL can write, at sizes around the MCU edges, against libjpeg-turbo itself."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 24
    rng = np.random.default_rng(11)
    n = 0
    for (h, w) in [(1, 1), (2, 3), (8, 8), (15, 17), (16, 16), (17, 33), (5, 70), (64, 2), (97, 131)]:
        y, x = np.mgrid[0:h, 0:w]
        img = np.clip(np.stack([128 + 110 * np.sin(x / 6.0) * np.cos(y / 5.0), (x * y) % 256, 255 - 2.0 * x + y], -1)
                      + rng.normal(0, 15, (h, w, 3)), 0, 255).astype(np.uint8)
        for mode, subs in (("RGB", (0, 1, 2)), ("L", (None,))):
            for sub in subs:
                for q in (35, 90, 100):
                    for opts in ({}, {"progressive": True}, {"optimize": True, "restart_marker_blocks": 2}):
                        kw = dict(quality=q, **opts)
                        if sub is not None:
                            kw["subsampling"] = sub
                        b = io.BytesIO()
                        Image.fromarray(img if mode == "RGB" else img[:, :, 1]).save(b, "JPEG", **kw)
                        data = b.getvalue()
                        ref_rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
                        g = Image.open(io.BytesIO(data))
                        g.draft("L", g.size)
                        ref_gray = np.asarray(g.convert("L"))
                        tag = f"{h}x{w} {mode} sub={sub} q={q} {opts}"
                        np.testing.assert_array_equal(capi.image_decode(data, True)[:, :, ::-1], ref_rgb, err_msg=tag)
                        np.testing.assert_array_equal(capi.image_decode(data, False), ref_gray, err_msg=tag)
                        n += 1
    assert n == 9 * 4 * 3 * 3
