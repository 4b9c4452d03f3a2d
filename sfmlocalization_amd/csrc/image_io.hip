// cv::imread for the query image (AKAZEOpenCV.cpp:60 IMREAD_GRAYSCALE; DenseLocalFeatureWrapper.cpp:85 and
// localizeImage.cc:463 IMREAD_COLOR), host code: sfmloc_image_decode / sfmloc_image_read of include/sfmloc.h.
//
// The reference gets its pixels from OpenCV's image codecs, i.e. from libjpeg and libpng; neither has headers in this
// image, so the containers the reference's tools accept (localization.cpp:204-206: jpg/jpeg/png) are decoded here:
//   JPEG  baseline, extended-sequential and progressive Huffman, 8 bit, 1 or 3 components, h/v sampling 1 or 2 for
//         the first component and 1x1 for the others (4:4:4, 4:2:2, 4:2:0), restart intervals.  The arithmetic is a
//         restatement of libjpeg's default decompression path so that the pixels are the ones imread returns:
//         jidctint.c (jpeg_idct_islow: 13-bit constants, two passes, range-limit table), jdsample.c (h2v1 / h2v2
//         "fancy" triangle upsampling incl. its edge rules and alternating rounding), jdcolor.c (16-bit fixed-point
//         YCbCr -> RGB tables).  IMREAD_GRAYSCALE asks libjpeg for JCS_GRAYSCALE (grfmt_jpeg.cpp), which is the
//         Y plane itself -- no colour conversion, no upsampling.  tests/test_image_io.py pins every mode against
//         libjpeg-turbo (through PIL) bit for bit.
//   PNG   8 bit, non-interlaced: gray, gray+alpha, RGB, RGBA, palette.  IMREAD_GRAYSCALE on a colour PNG is libpng's
//         png_set_rgb_to_gray(1, 0.299, 0.587) (grfmt_png.cpp): (9797 R + 19234 G + 3737 B) >> 15 (truncated), and the
//         pixel itself when R == G == B.
//   PGM/PPM binary, maxval 255 (cvtColor BGR2GRAY for colour, as grfmt_pxm.cpp does through icvCvt_BGR2Gray).
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sfmloc_internal.h"

namespace sfmloc {
namespace {

struct Image {
  int w = 0, h = 0, ch = 0;  // ch = channels of `px` (1 gray, 3 B G R)
  std::vector<uint8_t> px;
};

// ---------------------------------------------------------------------------------------------------------
// PNM
// ---------------------------------------------------------------------------------------------------------
inline uint8_t cv_bgr2gray(int r, int g, int b) { return (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14); }

bool decode_pnm(const uint8_t *raw, size_t n_raw, bool color, Image *im, std::string *err) {
  size_t p = 2;
  int vals[3], nv = 0;
  while (nv < 3 && p < n_raw) {
    while (p < n_raw && isspace(raw[p])) ++p;
    if (p < n_raw && raw[p] == '#') {
      while (p < n_raw && raw[p] != '\n') ++p;
      continue;
    }
    int v = 0;
    bool any = false;
    while (p < n_raw && isdigit(raw[p]) && v < 100000000) {
      v = v * 10 + (raw[p++] - '0');
      any = true;
    }
    if (!any) break;
    vals[nv++] = v;
  }
  if (nv < 3 || vals[2] != 255 || p >= n_raw || vals[0] <= 0 || vals[1] <= 0) {
    *err = "PNM: only binary P5/P6 with maxval 255";
    return false;
  }
  ++p;  // the single whitespace after maxval
  const size_t n = (size_t)vals[0] * vals[1], sch = raw[1] == '6' ? 3 : 1;
  if (n_raw < p + n * sch) {
    *err = "PNM: truncated";
    return false;
  }
  im->w = vals[0];
  im->h = vals[1];
  im->ch = color ? 3 : 1;
  im->px.resize(n * im->ch);
  for (size_t i = 0; i < n; ++i) {
    const int r = raw[p + sch * i], g = sch == 1 ? r : raw[p + 3 * i + 1], b = sch == 1 ? r : raw[p + 3 * i + 2];
    if (color) {
      im->px[3 * i] = (uint8_t)b;
      im->px[3 * i + 1] = (uint8_t)g;
      im->px[3 * i + 2] = (uint8_t)r;
    } else {
      im->px[i] = sch == 1 ? (uint8_t)r : cv_bgr2gray(r, g, b);
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------------------------------------
inline uint32_t be32(const uint8_t *p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

// libpng png_do_rgb_to_gray, 8 bit, no gamma; coefficients of png_set_rgb_to_gray_fixed(29900, 58700); libpng
// truncates ("the historical approach"), it does not round -- pinned against libpng 1.6 itself in tests/test_image_io.py
inline uint8_t png_rgb_to_gray(int r, int g, int b) {
  if (r == g && r == b) return (uint8_t)r;
  return (uint8_t)((9797 * r + 19234 * g + 3737 * b) >> 15);
}

bool decode_png(const uint8_t *raw, size_t n_raw, bool color, Image *im, std::string *err) {
  size_t p = 8;
  int bit_depth = 0, ctype = 0, interlace = 0, w = 0, h = 0;
  std::vector<uint8_t> idat, plte;
  bool have_hdr = false;
  while (p + 12 <= n_raw) {
    const uint32_t len = be32(raw + p);
    const char *type = reinterpret_cast<const char *>(raw + p + 4);
    if ((size_t)len > n_raw - p - 12) {
      *err = "PNG: truncated chunk";
      return false;
    }
    const uint8_t *data = raw + p + 8;
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      w = (int)be32(data);
      h = (int)be32(data + 4);
      bit_depth = data[8];
      ctype = data[9];
      interlace = data[12];
      have_hdr = true;
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    p += 12 + (size_t)len;
  }
  if (!have_hdr || w <= 0 || h <= 0 || w > 65535 || h > 65535) {
    *err = "PNG: bad header";
    return false;
  }
  if (interlace != 0) {
    *err = "PNG: interlaced files are not supported";
    return false;
  }
  const int sch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  const bool depth_ok = bit_depth == 8 || (bit_depth == 16 && ctype != 3) ||
                        ((bit_depth == 1 || bit_depth == 2 || bit_depth == 4) && (ctype == 0 || ctype == 3));
  if (!sch || !depth_ok) {
    *err = "PNG: unsupported colour type / bit depth";
    return false;
  }
  const size_t row_bytes = ((size_t)w * sch * bit_depth + 7) / 8;  // one filtered scanline
  const size_t fbpp = std::max<size_t>(1, (size_t)sch * bit_depth / 8);
  std::vector<uint8_t> buf((row_bytes + 1) * (size_t)h);
  uLongf out_len = buf.size();
  if (uncompress(buf.data(), &out_len, idat.data(), idat.size()) != Z_OK || out_len != buf.size()) {
    *err = "PNG: image data does not inflate to width x height";
    return false;
  }
  std::vector<uint8_t> lines(row_bytes * (size_t)h);
  for (int y = 0; y < h; ++y) {  // undo the scanline filters
    const uint8_t ft = buf[(row_bytes + 1) * y];
    const uint8_t *src = &buf[(row_bytes + 1) * y + 1];
    uint8_t *dst = &lines[row_bytes * y];
    const uint8_t *up = y ? &lines[row_bytes * (y - 1)] : nullptr;
    for (size_t x = 0; x < row_bytes; ++x) {
      const int a = x >= fbpp ? dst[x - fbpp] : 0, b = up ? up[x] : 0, c = (up && x >= fbpp) ? up[x - fbpp] : 0;
      int pred = 0;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        } break;
        default: *err = "PNG: bad filter type"; return false;
      }
      dst[x] = (uint8_t)(src[x] + pred);
    }
  }
  // 8 bits per sample: png_set_strip_16 keeps the high byte, png_set_expand_gray_1_2_4_to_8 scales to 0..255,
  // palette indices stay indices
  const size_t stride = (size_t)w * sch;
  std::vector<uint8_t> img(stride * (size_t)h);
  for (int y = 0; y < h; ++y) {
    const uint8_t *src = &lines[row_bytes * y];
    uint8_t *dst = &img[stride * y];
    if (bit_depth == 8) {
      memcpy(dst, src, stride);
    } else if (bit_depth == 16) {
      for (size_t x = 0; x < stride; ++x) dst[x] = src[2 * x];
    } else {
      const int mask = (1 << bit_depth) - 1, scale = ctype == 0 ? 255 / mask : 1;
      for (int x = 0; x < w; ++x) {
        const int bit = x * bit_depth, v = (src[bit >> 3] >> (8 - bit_depth - (bit & 7))) & mask;
        dst[x] = (uint8_t)(v * scale);
      }
    }
  }
  const size_t n = (size_t)w * h;
  im->w = w;
  im->h = h;
  im->ch = color ? 3 : 1;
  im->px.resize(n * im->ch);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t *px = &img[i * sch];
    int r, g, b;
    if (ctype == 0 || ctype == 4) {
      r = g = b = px[0];
    } else if (ctype == 3) {
      if ((size_t)px[0] * 3 + 2 >= plte.size()) {
        *err = "PNG: palette index out of range";
        return false;
      }
      r = plte[px[0] * 3], g = plte[px[0] * 3 + 1], b = plte[px[0] * 3 + 2];
    } else {
      r = px[0], g = px[1], b = px[2];
    }
    if (color) {
      im->px[3 * i] = (uint8_t)b;  // imread(IMREAD_COLOR) order
      im->px[3 * i + 1] = (uint8_t)g;
      im->px[3 * i + 2] = (uint8_t)r;
    } else {
      im->px[i] = png_rgb_to_gray(r, g, b);
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// JPEG
// ---------------------------------------------------------------------------------------------------------
const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33,
                                  40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                  29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                  47, 55, 62, 63,
                                  // a run that overshoots the block lands here (jutils.c jpeg_natural_order)
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {
  bool defined = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  // canonical decoding (ITU T.81 F.2.2.3): code lengths 1..16
  int32_t maxcode[18];
  int32_t valoffset[17];
  // 9-bit lookahead: (length << 8) | symbol, 0 when the code is longer
  uint16_t look[512];

  bool build() {
    int32_t code = 0;
    int k = 0;
    memset(look, 0, sizeof(look));
    for (int l = 1; l <= 16; ++l) {
      valoffset[l] = k - code;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (k >= 256) return false;
        if (l <= 9) {
          const int first = code << (9 - l), cnt = 1 << (9 - l);
          if (first + cnt > 512) return false;
          for (int j = 0; j < cnt; ++j) look[first + j] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      if (code > (1 << l)) return false;
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    return true;
  }
};

struct BitReader {
  const uint8_t *p, *end;
  uint64_t acc = 0;
  int n = 0;         // valid bits in acc (right-aligned)
  int marker = 0;    // a marker met in the entropy-coded data (0 = none)
  bool overrun = false;

  void fill() {
    while (n <= 56) {
      int byte = 0;
      if (!marker && p < end) {
        byte = *p++;
        if (byte == 0xFF) {
          while (p < end && *p == 0xFF) ++p;  // fill bytes
          const int nx = p < end ? *p++ : 0xD9;
          if (nx != 0) {
            marker = nx;
            byte = 0;
          }
        }
      } else if (!marker) {
        marker = 0xD9;
      }
      acc = (acc << 8) | (uint64_t)byte;
      n += 8;
      if (marker) break;  // zeros are fed one byte at a time past a marker
    }
  }
  inline int peek(int k) {
    if (n < k) {
      fill();
      if (n < k) {  // past a marker: pad with zeros (jdhuff.c does the same and warns)
        acc <<= (k - n) + 8;
        n += (k - n) + 8;
        overrun = true;
      }
    }
    return (int)((acc >> (n - k)) & ((1u << k) - 1));
  }
  inline void skip(int k) { n -= k; }
  inline int get(int k) {
    if (k == 0) return 0;
    const int v = peek(k);
    n -= k;
    return v;
  }
  void reset() {
    acc = 0;
    n = 0;
    marker = 0;
  }
};

inline int huff_decode(BitReader &br, const HuffTable &t) {
  const int look = br.peek(9);
  const uint16_t e = t.look[look];
  if (e) {
    br.skip(e >> 8);
    return e & 0xFF;
  }
  int32_t code = look;
  int l = 9;
  br.skip(9);
  while (l < 17 && code > t.maxcode[l]) {
    code = (code << 1) | br.get(1);
    ++l;
  }
  if (l > 16) return 0;  // corrupt data: libjpeg warns and uses 0
  return t.vals[(code + t.valoffset[l]) & 0xFF];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Component {
  int id = 0, hs = 1, vs = 1, tq = 0;
  int td = 0, ta = 0;           // current scan's tables
  int bw = 0, bh = 0;           // blocks allocated (padded to whole MCUs)
  int rw = 0, rh = 0;           // downsampled_width / _height: the real samples
  int last_dc = 0;
  std::vector<int16_t> coef;    // bw*bh*64, natural order
  std::vector<uint8_t> plane;   // (bw*8) x (bh*8)
};

// jidctint.c jpeg_idct_islow on one dequantised block -> 8x8 samples
void idct_islow(const int16_t *in, const uint16_t *q, uint8_t *out, int out_stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                 F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
  int ws[64];
  for (int c = 0; c < 8; ++c) {  // pass 1: columns
    if ((in[8 + c] | in[16 + c] | in[24 + c] | in[32 + c] | in[40 + c] | in[48 + c] | in[56 + c]) == 0) {
      // a column with only its DC term: the arithmetic below reduces to dc << PASS1_BITS in every row
      const int dcv = (int)((long)in[c] * (long)q[c] * (1L << P1));
      for (int r = 0; r < 8; ++r) ws[8 * r + c] = dcv;
      continue;
    }
    long d[8];
    for (int r = 0; r < 8; ++r) d[r] = (long)in[8 * r + c] * (long)q[8 * r + c];
    long z2 = d[2], z3 = d[6];
    long z1 = (z2 + z3) * F0_541;
    long tmp2 = z1 + z3 * (-F1_847);
    long tmp3 = z1 + z2 * F0_765;
    z2 = d[0];
    z3 = d[4];
    long tmp0 = (z2 + z3) * (1L << CB);
    long tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = d[7];
    tmp1 = d[5];
    tmp2 = d[3];
    tmp3 = d[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298;
    tmp1 *= F2_053;
    tmp2 *= F3_072;
    tmp3 *= F1_501;
    z1 *= -F0_899;
    z2 *= -F2_562;
    z3 *= -F1_961;
    z4 *= -F0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    ws[8 * 0 + c] = (int)descale(tmp10 + tmp3, CB - P1);
    ws[8 * 7 + c] = (int)descale(tmp10 - tmp3, CB - P1);
    ws[8 * 1 + c] = (int)descale(tmp11 + tmp2, CB - P1);
    ws[8 * 6 + c] = (int)descale(tmp11 - tmp2, CB - P1);
    ws[8 * 2 + c] = (int)descale(tmp12 + tmp1, CB - P1);
    ws[8 * 5 + c] = (int)descale(tmp12 - tmp1, CB - P1);
    ws[8 * 3 + c] = (int)descale(tmp13 + tmp0, CB - P1);
    ws[8 * 4 + c] = (int)descale(tmp13 - tmp0, CB - P1);
  }
  // the post-IDCT range-limit table (jdmaster.c prepare_range_limit_table), indexed with 10 bits
  auto limit = [](long v) -> uint8_t {
    const int i = (int)(v & 1023);
    return i < 128 ? (uint8_t)(128 + i) : i < 512 ? 255 : i < 896 ? 0 : (uint8_t)(i - 896);
  };
  for (int r = 0; r < 8; ++r) {  // pass 2: rows
    const int *w = ws + 8 * r;
    uint8_t *o = out + (size_t)r * out_stride;
    if ((w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0) {  // same value as the full computation
      const uint8_t v = limit(descale((long)w[0], P1 + 3));
      for (int x = 0; x < 8; ++x) o[x] = v;
      continue;
    }
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F0_541;
    long tmp2 = z1 + z3 * (-F1_847);
    long tmp3 = z1 + z2 * F0_765;
    long tmp0 = ((long)w[0] + (long)w[4]) * (1L << CB);
    long tmp1 = ((long)w[0] - (long)w[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7];
    tmp1 = w[5];
    tmp2 = w[3];
    tmp3 = w[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298;
    tmp1 *= F2_053;
    tmp2 *= F3_072;
    tmp3 *= F1_501;
    z1 *= -F0_899;
    z2 *= -F2_562;
    z3 *= -F1_961;
    z4 *= -F0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    constexpr int SH = CB + P1 + 3;
    o[0] = limit(descale(tmp10 + tmp3, SH));
    o[7] = limit(descale(tmp10 - tmp3, SH));
    o[1] = limit(descale(tmp11 + tmp2, SH));
    o[6] = limit(descale(tmp11 - tmp2, SH));
    o[2] = limit(descale(tmp12 + tmp1, SH));
    o[5] = limit(descale(tmp12 - tmp1, SH));
    o[3] = limit(descale(tmp13 + tmp0, SH));
    o[4] = limit(descale(tmp13 - tmp0, SH));
  }
}

struct Jpeg {
  int w = 0, h = 0, ncomp = 0;
  bool progressive = false, have_sof = false;
  bool jfif = false, adobe = false;
  int adobe_transform = 0;
  int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  int restart_interval = 0;
  uint16_t quant[4][64];
  bool quant_defined[4] = {false, false, false, false};
  HuffTable dc[4], ac[4];
  Component comp[3];
  std::string err;

  bool fail(const char *m) {
    err = std::string("JPEG: ") + m;
    return false;
  }

  bool read_dqt(const uint8_t *d, size_t len) {
    size_t p = 0;
    while (p < len) {
      const int pq = d[p] >> 4, tq = d[p] & 15;
      ++p;
      if (tq > 3 || pq > 1) return fail("bad quantisation table");
      if (p + (pq ? 128 : 64) > len) return fail("truncated quantisation table");
      for (int i = 0; i < 64; ++i) {
        const int v = pq ? (d[p] << 8) | d[p + 1] : d[p];
        p += pq ? 2 : 1;
        quant[tq][kZigzag[i]] = (uint16_t)v;
      }
      quant_defined[tq] = true;
    }
    return true;
  }

  bool read_dht(const uint8_t *d, size_t len) {
    size_t p = 0;
    while (p < len) {
      if (p + 17 > len) return fail("truncated Huffman table");
      const int tc = d[p] >> 4, th = d[p] & 15;
      if (tc > 1 || th > 3) return fail("bad Huffman table index");
      HuffTable &t = tc ? ac[th] : dc[th];
      int total = 0;
      t.bits[0] = 0;
      for (int i = 1; i <= 16; ++i) {
        t.bits[i] = d[p + i];
        total += t.bits[i];
      }
      p += 17;
      if (total > 256 || p + total > len) return fail("bad Huffman table");
      memset(t.vals, 0, sizeof(t.vals));
      memcpy(t.vals, d + p, total);
      p += total;
      if (!t.build()) return fail("bad Huffman code lengths");
      t.defined = true;
    }
    return true;
  }

  bool read_sof(const uint8_t *d, size_t len, bool allocate) {
    if (have_sof) return fail("more than one frame header");
    if (len < 6) return fail("truncated frame header");
    if (d[0] != 8) return fail("only 8-bit precision is supported");
    h = (d[1] << 8) | d[2];
    w = (d[3] << 8) | d[4];
    ncomp = d[5];
    if (w <= 0 || h <= 0) return fail("empty image");
    if (ncomp != 1 && ncomp != 3) return fail("only 1- and 3-component images are supported");
    if (len < 6 + 3 * (size_t)ncomp) return fail("truncated frame header");
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.id = d[6 + 3 * i];
      c.hs = d[7 + 3 * i] >> 4;
      c.vs = d[7 + 3 * i] & 15;
      c.tq = d[8 + 3 * i];
      if (c.tq > 3) return fail("bad quantisation table index");
    }
    if (ncomp == 1) comp[0].hs = comp[0].vs = 1;  // a single component is never subsampled (jdinput.c)
    const bool ok_first = (comp[0].hs == 1 || comp[0].hs == 2) && (comp[0].vs == 1 || comp[0].vs == 2);
    bool ok_rest = true;
    for (int i = 1; i < ncomp; ++i) ok_rest = ok_rest && comp[i].hs == 1 && comp[i].vs == 1;
    if (!ok_first || !ok_rest || (ncomp == 3 && comp[0].hs == 1 && comp[0].vs == 2))
      return fail("unsupported chroma subsampling (4:4:4, 4:2:2 and 4:2:0 are)");
    hmax = comp[0].hs;
    vmax = comp[0].vs;
    mcus_x = (w + 8 * hmax - 1) / (8 * hmax);
    mcus_y = (h + 8 * vmax - 1) / (8 * vmax);
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.bw = mcus_x * c.hs;
      c.bh = mcus_y * c.vs;
      c.rw = (w * c.hs + hmax - 1) / hmax;
      c.rh = (h * c.vs + vmax - 1) / vmax;
      if (allocate) c.coef.assign((size_t)c.bw * c.bh * 64, 0);
    }
    have_sof = true;
    return true;
  }

  // ----- entropy decoding of one scan into the coefficient arrays -----
  struct Scan {
    int n = 0;
    int ci[3];
    int ss = 0, se = 63, ah = 0, al = 0;
  };

  bool decode_block_sequential(BitReader &br, Component &c, int16_t *blk) {
    const HuffTable &dt = dc[c.td], &at = ac[c.ta];
    int s = huff_decode(br, dt);
    if (s) {
      if (s > 15) return fail("bad DC difference");
      s = extend(br.get(s), s);
    }
    c.last_dc = (int)((unsigned)c.last_dc + (unsigned)s);  // wraps on hostile data instead of overflowing
    blk[0] = (int16_t)c.last_dc;
    for (int k = 1; k < 64;) {
      const int rs = huff_decode(br, at), r = rs >> 4;
      s = rs & 15;
      if (s) {
        k += r;
        blk[kZigzag[k]] = (int16_t)extend(br.get(s), s);
        ++k;
      } else {
        if (r != 15) break;
        k += 16;
      }
    }
    return true;
  }

  // jdphuff.c decode_mcu_DC_first / _AC_first / _DC_refine / _AC_refine for one block
  bool decode_block_progressive(BitReader &br, Component &c, int16_t *blk, const Scan &sc, unsigned &eobrun) {
    if (sc.ss == 0) {
      if (sc.ah == 0) {
        int s = huff_decode(br, dc[c.td]);
        if (s) {
          if (s > 15) return fail("bad DC difference");
          s = extend(br.get(s), s);
        }
        c.last_dc = (int)((unsigned)c.last_dc + (unsigned)s);  // wraps on hostile data instead of overflowing
        blk[0] = (int16_t)((int64_t)c.last_dc * (1 << sc.al));
      } else if (br.get(1)) {
        blk[0] |= (int16_t)(1 << sc.al);
      }
      return true;
    }
    const HuffTable &at = ac[c.ta];
    if (sc.ah == 0) {
      if (eobrun > 0) {
        --eobrun;
        return true;
      }
      for (int k = sc.ss; k <= sc.se; ++k) {
        const int rs = huff_decode(br, at), r = rs >> 4;
        int s = rs & 15;
        if (s) {
          k += r;
          s = extend(br.get(s), s);
          blk[kZigzag[k]] = (int16_t)(s * (1 << sc.al));
        } else if (r == 15) {
          k += 15;
        } else {
          eobrun = 1u << r;
          if (r) eobrun += (unsigned)br.get(r);
          --eobrun;
          break;
        }
      }
      return true;
    }
    const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
    int k = sc.ss;
    if (eobrun == 0) {
      for (; k <= sc.se; ++k) {
        const int rs = huff_decode(br, at);
        int r = rs >> 4, s = rs & 15;
        if (s) {
          s = br.get(1) ? p1 : m1;  // the size must be 1
        } else if (r != 15) {
          eobrun = 1u << r;
          if (r) eobrun += (unsigned)br.get(r);
          break;  // force end-of-band
        }
        do {
          int16_t *co = blk + kZigzag[k];
          if (*co != 0) {
            if (br.get(1) && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
          } else if (--r < 0) {
            break;
          }
          ++k;
        } while (k <= sc.se);
        if (s) blk[kZigzag[k]] = (int16_t)s;
      }
    }
    if (eobrun > 0) {
      for (; k <= sc.se; ++k) {
        int16_t *co = blk + kZigzag[k];
        if (*co != 0 && br.get(1) && (*co & p1) == 0) *co = (int16_t)(*co + (*co >= 0 ? p1 : m1));
      }
      --eobrun;
    }
    return true;
  }

  bool decode_scan(const uint8_t *d, size_t len, const uint8_t *data, const uint8_t *data_end, const uint8_t **next) {
    if (!have_sof) return fail("scan before the frame header");
    if (len < 1) return fail("truncated scan header");
    Scan sc;
    sc.n = d[0];
    if (sc.n < 1 || sc.n > ncomp || len < 1 + 2 * (size_t)sc.n + 3) return fail("bad scan header");
    for (int i = 0; i < sc.n; ++i) {
      const int id = d[1 + 2 * i];
      int ci = -1;
      for (int j = 0; j < ncomp; ++j)
        if (comp[j].id == id) ci = j;
      if (ci < 0) return fail("scan names an unknown component");
      sc.ci[i] = ci;
      comp[ci].td = d[2 + 2 * i] >> 4;
      comp[ci].ta = d[2 + 2 * i] & 15;
      if (comp[ci].td > 3 || comp[ci].ta > 3) return fail("bad table selector");
    }
    sc.ss = d[1 + 2 * sc.n];
    sc.se = d[2 + 2 * sc.n];
    sc.ah = d[3 + 2 * sc.n] >> 4;
    sc.al = d[3 + 2 * sc.n] & 15;
    if (progressive) {
      if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || (sc.ss == 0 && sc.se != 0) || (sc.ss != 0 && sc.n != 1))
        return fail("bad progressive scan parameters");
    } else {
      sc.ss = 0;
      sc.se = 63;
      sc.ah = sc.al = 0;
    }
    for (int i = 0; i < sc.n; ++i) {
      const Component &c = comp[sc.ci[i]];
      const bool need_dc = sc.ss == 0 && sc.ah == 0, need_ac = progressive ? sc.ss != 0 : true;
      if ((need_dc && !dc[c.td].defined) || (need_ac && !ac[c.ta].defined)) return fail("scan uses an undefined Huffman table");
    }
    BitReader br;
    br.p = data;
    br.end = data_end;
    unsigned eobrun = 0;
    for (int i = 0; i < ncomp; ++i) comp[i].last_dc = 0;
    // a one-component scan walks that component's own blocks (ceil(real size / 8)), not the padded MCU grid
    int nx, ny;
    if (sc.n == 1) {
      const Component &c = comp[sc.ci[0]];
      nx = (c.rw + 7) / 8;
      ny = (c.rh + 7) / 8;
    } else {
      nx = mcus_x;
      ny = mcus_y;
    }
    long until_restart = restart_interval;
    int next_rst = 0;
    for (int my = 0; my < ny; ++my) {
      for (int mx = 0; mx < nx; ++mx) {
        if (restart_interval && until_restart == 0) {
          // RSTn: byte-align, expect the marker, reset the predictors (jdhuff.c process_restart)
          br.n = 0;
          br.acc = 0;
          if (!br.marker) {  // the marker may not have been reached yet when the last MCU ended on a byte boundary
            br.fill();
            br.n = 0;
            br.acc = 0;
          }
          if (br.marker != 0xD0 + next_rst) return fail("missing restart marker");
          br.marker = 0;
          next_rst = (next_rst + 1) & 7;
          for (int i = 0; i < ncomp; ++i) comp[i].last_dc = 0;
          eobrun = 0;
          until_restart = restart_interval;
        }
        if (sc.n == 1) {
          Component &c = comp[sc.ci[0]];
          int16_t *blk = &c.coef[((size_t)my * c.bw + mx) * 64];
          const bool ok = progressive ? decode_block_progressive(br, c, blk, sc, eobrun) : decode_block_sequential(br, c, blk);
          if (!ok) return false;
        } else {
          for (int i = 0; i < sc.n; ++i) {
            Component &c = comp[sc.ci[i]];
            for (int by = 0; by < c.vs; ++by)
              for (int bx = 0; bx < c.hs; ++bx) {
                int16_t *blk = &c.coef[((size_t)(my * c.vs + by) * c.bw + (mx * c.hs + bx)) * 64];
                const bool ok =
                    progressive ? decode_block_progressive(br, c, blk, sc, eobrun) : decode_block_sequential(br, c, blk);
                if (!ok) return false;
              }
          }
        }
        --until_restart;
      }
    }
    // position after the scan: the marker the bit reader stopped at, if any, else search for the next one
    if (br.marker) {
      *next = br.p - 2;
    } else {
      const uint8_t *q = br.p;
      while (q + 1 < data_end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
      *next = q;
    }
    return true;
  }

  bool parse(const uint8_t *raw, size_t n, bool header_only) {
    size_t p = 2;
    bool seen_scan = false;
    while (p + 4 <= n) {
      if (raw[p] != 0xFF) {
        ++p;  // garbage between segments: libjpeg skips it with a warning
        continue;
      }
      const int m = raw[p + 1];
      if (m == 0xFF) {
        ++p;
        continue;
      }
      if (m == 0xD9) break;
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) {
        p += 2;
        continue;
      }
      const size_t len = ((size_t)raw[p + 2] << 8) | raw[p + 3];
      if (len < 2 || p + 2 + len > n) return fail("truncated segment");
      const uint8_t *d = raw + p + 4;
      const size_t dl = len - 2;
      p += 2 + len;
      switch (m) {
        case 0xDB:
          if (!read_dqt(d, dl)) return false;
          break;
        case 0xC4:
          if (!read_dht(d, dl)) return false;
          break;
        case 0xC0:
        case 0xC1:
        case 0xC2:
          progressive = (m == 0xC2);
          if (!read_sof(d, dl, !header_only)) return false;
          if (header_only) return true;
          break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
          return fail("lossless, hierarchical and arithmetic-coded files are not supported");
        case 0xDD:
          if (dl < 2) return fail("truncated restart interval");
          restart_interval = (d[0] << 8) | d[1];
          break;
        case 0xE0:
          if (dl >= 5 && !memcmp(d, "JFIF", 5)) jfif = true;
          break;
        case 0xEE:
          if (dl >= 12 && !memcmp(d, "Adobe", 5)) {
            adobe = true;
            adobe_transform = d[11];
          }
          break;
        case 0xDA: {
          if (header_only) return have_sof ? true : fail("scan before the frame header");
          const uint8_t *next = nullptr;
          if (!decode_scan(d, dl, raw + p, raw + n, &next)) return false;
          seen_scan = true;
          p = (size_t)(next - raw);
        } break;
        default: break;  // APPn, COM, DNL ...
      }
    }
    if (!have_sof || !seen_scan) return fail("no image data");
    return true;
  }

  void reconstruct(int n_planes) {
    for (int i = 0; i < n_planes; ++i) {
      Component &c = comp[i];
      const uint16_t *q = quant[c.tq];
      const int stride = c.bw * 8;
      c.plane.resize((size_t)stride * c.bh * 8);
      for (int by = 0; by < c.bh; ++by)
        for (int bx = 0; bx < c.bw; ++bx)
          idct_islow(&c.coef[((size_t)by * c.bw + bx) * 64], q, &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
    }
  }

  // jdsample.c: one output row `y` (full resolution) of component c, `out` has room for 2*rw+2 samples
  void upsample_row(const Component &c, int y, uint8_t *out) const {
    const int stride = c.bw * 8;
    const int hx = hmax / c.hs, vx = vmax / c.vs;
    if (hx == 1 && vx == 1) {
      memcpy(out, &c.plane[(size_t)y * stride], (size_t)w);
      return;
    }
    const int n = c.rw;
    if (n <= 2) {  // jdsample.c jinit_upsampler: the triangle filters need downsampled_width > 2, else replication
      const uint8_t *in = &c.plane[(size_t)(vx == 2 ? y >> 1 : y) * stride];
      for (int x = 0; x < 2 * n; ++x) out[x] = in[x >> 1];
      return;
    }
    if (hx == 2 && vx == 1) {  // h2v1_fancy_upsample
      const uint8_t *in = &c.plane[(size_t)y * stride];
      out[0] = in[0];
      out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
      for (int i = 1; i < n - 1; ++i) {
        const int v = in[i] * 3;
        out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
        out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
      }
      out[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
      out[2 * n - 1] = in[n - 1];
      return;
    }
    // h2v2_fancy_upsample: the nearer input row counts 3/4, the farther 1/4; rows beyond the image replicate the edge
    const int r0 = y >> 1;
    int r1 = (y & 1) ? r0 + 1 : r0 - 1;
    r1 = std::max(0, std::min(c.rh - 1, r1));
    const uint8_t *in0 = &c.plane[(size_t)r0 * stride], *in1 = &c.plane[(size_t)r1 * stride];
    int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
    out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
    out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol;
    thiscol = nextcol;
    for (int i = 1; i < n - 1; ++i) {
      nextcol = in0[i + 1] * 3 + in1[i + 1];
      out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
      out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
      lastcol = thiscol;
      thiscol = nextcol;
    }
    out[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
    out[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
  }

  bool to_image(bool color, Image *im) {
    // jdapimin.c default_decompress_parms: the file's colour space
    bool ycc = true;
    if (ncomp == 3) {
      if (jfif) ycc = true;
      else if (adobe) ycc = adobe_transform != 0;
      else ycc = !(comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
    }
    im->w = w;
    im->h = h;
    im->ch = color ? 3 : 1;
    im->px.resize((size_t)w * h * im->ch);
    const bool y_only = ncomp == 1 || (!color && ycc);
    reconstruct(y_only ? 1 : ncomp);
    if (y_only) {  // JCS_GRAYSCALE out of a gray or YCbCr file is the first plane (jdcolor.c grayscale_convert)
      const int stride = comp[0].bw * 8;
      for (int y = 0; y < h; ++y) {
        const uint8_t *src = &comp[0].plane[(size_t)y * stride];
        uint8_t *dst = &im->px[(size_t)y * w * im->ch];
        if (!color) memcpy(dst, src, (size_t)w);
        else
          for (int x = 0; x < w; ++x) dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = src[x];
      }
      return true;
    }
    // jdcolor.c build_ycc_rgb_table
    struct YccTables {
      int cr_r[256], cb_b[256];
      long cr_g[256], cb_g[256];
      YccTables() {
        for (int i = 0; i < 256; ++i) {
          const long x = i - 128;
          cr_r[i] = (int)((91881L * x + 32768L) >> 16);
          cb_b[i] = (int)((116130L * x + 32768L) >> 16);
          cr_g[i] = -46802L * x;
          cb_g[i] = -22554L * x + 32768L;
        }
      }
    };
    static const YccTables ycc_tab;  // built once, safely, whichever thread decodes first (server workers decode concurrently)
    const int *const cr_r = ycc_tab.cr_r, *const cb_b = ycc_tab.cb_b;
    const long *const cr_g = ycc_tab.cr_g, *const cb_g = ycc_tab.cb_g;
    auto clamp = [](int v) -> uint8_t { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    std::vector<uint8_t> rows[3];
    for (int i = 0; i < 3; ++i) rows[i].resize((size_t)2 * comp[i].bw * 8 + 16);
    for (int y = 0; y < h; ++y) {
      for (int i = 0; i < 3; ++i) upsample_row(comp[i], y, rows[i].data());
      uint8_t *dst = &im->px[(size_t)y * w * im->ch];
      for (int x = 0; x < w; ++x) {
        const int a = rows[0][x], b = rows[1][x], c = rows[2][x];
        int R, G, B;
        if (ycc) {
          R = clamp(a + cr_r[c]);
          G = clamp(a + (int)((cb_g[b] + cr_g[c]) >> 16));
          B = clamp(a + cb_b[b]);
        } else {
          R = a, G = b, B = c;
        }
        if (color) {
          dst[3 * x] = (uint8_t)B;
          dst[3 * x + 1] = (uint8_t)G;
          dst[3 * x + 2] = (uint8_t)R;
        } else {  // jdcolor.c rgb_gray_convert
          dst[x] = (uint8_t)((19595L * R + 38470L * G + 7471L * B + 32768L) >> 16);
        }
      }
    }
    return true;
  }
};

bool decode_any(const uint8_t *raw, size_t n, bool color, bool size_only, Image *im, std::string *err) {
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (n >= 33 && !memcmp(raw, png_sig, 8)) {
    if (size_only && !memcmp(raw + 12, "IHDR", 4)) {
      im->w = (int)be32(raw + 16);
      im->h = (int)be32(raw + 20);
      if (im->w > 0 && im->h > 0 && im->w <= 65535 && im->h <= 65535) return true;
    }
    return decode_png(raw, n, color, im, err);
  }
  if (n >= 4 && raw[0] == 0xFF && raw[1] == 0xD8) {
    Jpeg j;
    if (size_only) {
      if (!j.parse(raw, n, true)) {
        *err = j.err;
        return false;
      }
      im->w = j.w;
      im->h = j.h;
      return true;
    }
    if (!j.parse(raw, n, false) || !j.to_image(color, im)) {
      *err = j.err;
      return false;
    }
    return true;
  }
  if (n >= 8 && raw[0] == 'P' && (raw[1] == '5' || raw[1] == '6')) return decode_pnm(raw, n, color, im, err);
  *err = "not a JPEG, PNG or binary PGM/PPM image";
  return false;
}

}  // namespace
}  // namespace sfmloc

using namespace sfmloc;

extern "C" int sfmloc_image_decode(const uint8_t *bytes, uint64_t n_bytes, int32_t color, uint8_t *out, uint64_t cap,
                                   int32_t *width, int32_t *height) {
  SFM_CHECK(bytes && width && height, SFMLOC_EINVAL, "sfmloc_image_decode: null argument");
  Image im;
  std::string err;
  try {
    if (!decode_any(bytes, (size_t)n_bytes, color != 0, out == nullptr, &im, &err)) {
      set_error("sfmloc_image_decode: %s", err.c_str());
      return SFMLOC_EIO;
    }
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_image_decode: out of host memory");
    return SFMLOC_ENOMEM;
  }
  *width = im.w;
  *height = im.h;
  if (!out) return SFMLOC_OK;  // size query
  SFM_CHECK(cap >= im.px.size(), SFMLOC_ECAP, "sfmloc_image_decode: buffer holds %llu bytes, the image needs %llu",
            (unsigned long long)cap, (unsigned long long)im.px.size());
  memcpy(out, im.px.data(), im.px.size());
  return SFMLOC_OK;
}

extern "C" int sfmloc_image_read(const char *path, int32_t color, uint8_t *out, uint64_t cap, int32_t *width,
                                 int32_t *height) {
  SFM_CHECK(path && width && height, SFMLOC_EINVAL, "sfmloc_image_read: null argument");
  FILE *f = fopen(path, "rb");
  SFM_CHECK(f != nullptr, SFMLOC_EIO, "sfmloc_image_read: cannot open %s", path);
  std::vector<uint8_t> raw;
  uint8_t buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof(buf), f)) > 0) raw.insert(raw.end(), buf, buf + got);
  fclose(f);
  return sfmloc_image_decode(raw.data(), raw.size(), color, out, cap, width, height);
}
