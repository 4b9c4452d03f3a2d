/*
 * sfm_oracle_akaze.c -- CPU restatement of AKAZE detection + full 486-bit M-LDB description,
 * the call `cv::AKAZE::create(DESCRIPTOR_MLDB, 0, 3, thres, nOct, nOctLay)->detectAndCompute(gray, ...)` of
 * VisionLocalizeCommon/src/AKAZEOpenCV.cpp:44-46,67 and `cv::AKAZE::create()->compute(...)` on given keypoints of
 * BoWCommon/src/DenseLocalFeatureWrapper.cpp:42,146.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: OpenCV 3.0 is not in /root/reference nor in this image; this file
 * restates the published algorithm (Alcantarilla, Nuevo, Bartoli, "Fast Explicit Diffusion for Accelerated Features
 * in Nonlinear Scale Spaces", BMVC 2013) in the structure of OpenCV 3.0's AKAZEFeatures as the builder knows it.
 * Build-defined points (float summation orders, border handling where OpenCV reads outside the image, fixed-order
 * atan / sin / cos, NaN-free angle for a zero gradient) are marked "BUILD-DEFINED"; the HIP kernels follow the same
 * operation order so that keypoints and descriptor bits can be compared exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AKZ_MAX_LEVELS 32
#define AKZ_PI_F 3.14159265358979323846f

typedef struct {
  int w, h, octave, sublevel, sigma_size, nsteps;
  float esigma, etime;
  float tsteps[64];
} akz_level;

typedef struct {
  int nlev;
  akz_level lev[AKZ_MAX_LEVELS];
  float g16[9];  /* gaussian sigma = soffset (1.6), ksize 9 */
  float g10[5];  /* gaussian sigma = 1.0, ksize 5 */
  float gauss25[7][7];
} akz_plan;

static int fround_(float f) { return (int)(f + 0.5f); }

/* ---- plan: everything OpenCV computes once per image size (Allocate_Memory_Evolution, fed_tau_by_process_time) */
static int is_prime(int n) {
  if (n <= 3) return n > 1;
  if (n % 2 == 0 || n % 3 == 0) return 0;
  for (int i = 5; i * i <= n; i += 6)
    if (n % i == 0 || n % (i + 2) == 0) return 0;
  return 1;
}

static int fed_tau(float T, float tau_max, float *tau) {
  const int n = (int)(ceilf(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);
  if (n <= 0) return 0;
  const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
  float tauh[64];
  const float c = 1.0f / (4.0f * (float)n + 2.0f);
  const float d = scale * tau_max / 2.0f;
  for (int k = 0; k < n; ++k) {
    const float hh = cosf(AKZ_PI_F * (2.0f * (float)k + 1.0f) * c);
    tauh[k] = d / (hh * hh);
  }
  const int kappa = n / 2;
  int prime = n + 1;
  while (!is_prime(prime)) prime++;
  for (int k = 0, l = 0; l < n; ++k, ++l) {
    int index;
    while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
    tau[l] = tauh[index];
  }
  return n;
}

static void gaussian_kernel(int ksize, float sigma, float *cf) {
  const double scale2x = -0.5 / ((double)sigma * sigma);
  double sum = 0;
  for (int i = 0; i < ksize; ++i) {
    const double x = i - (ksize - 1) * 0.5;
    const double t = exp(scale2x * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1.0 / sum;
  for (int i = 0; i < ksize; ++i) cf[i] = (float)(cf[i] * sum);
}

void orc_akaze_plan(int w, int h, int omax, int nsublevels, akz_plan *P) {
  const float soffset = 1.6f, derivative_factor = 1.5f;
  P->nlev = 0;
  for (int i = 0; i < omax; ++i) {
    const float rfactor = 1.0f / powf(2.0f, (float)i);
    const int lh = (int)(h * rfactor), lw = (int)(w * rfactor);
    if ((lw < 80 || lh < 40) && i != 0) break;
    for (int j = 0; j < nsublevels; ++j) {
      akz_level *L = &P->lev[P->nlev++];
      memset(L, 0, sizeof(*L));
      L->w = lw;
      L->h = lh;
      L->esigma = soffset * powf(2.0f, (float)j / (float)nsublevels + (float)i);
      L->sigma_size = fround_(L->esigma * derivative_factor / powf(2.0f, (float)i));
      L->etime = 0.5f * (L->esigma * L->esigma);
      L->octave = i;
      L->sublevel = j;
    }
  }
  for (int i = 1; i < P->nlev; ++i) {
    const float ttime = P->lev[i].etime - P->lev[i - 1].etime;
    P->lev[i].nsteps = fed_tau(ttime, 0.25f, P->lev[i].tsteps);
  }
  gaussian_kernel(9, 1.6f, P->g16);
  gaussian_kernel(5, 1.0f, P->g10);
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j)
      P->gauss25[i][j] = (float)(exp(-(double)(i * i + j * j) / 12.5) / (2.0 * 3.14159265358979323846 * 6.25));
}

/* ---- BUILD-DEFINED fixed-order float math -------------------------------------------------------------------- */
static float det_atanf(float x) { /* x >= 0 (Cephes atanf reduction) */
  float y;
  if (x > 2.414213562373095f) {
    y = 1.5707963267948966f;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) {
    y = 0.7853981633974483f;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  const float z = x * x;
  float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
  p = p * z + 1.99777106478e-1f;
  p = p * z - 3.33329491539e-1f;
  p = p * z;
  p = p * x + x;
  return y + p;
}

/* getAngle of OpenCV's KAZE utils; a zero vector gives 0 (BUILD-DEFINED: OpenCV would produce NaN from 0/0) */
static float get_angle(float x, float y) {
  if (x == 0.0f && y == 0.0f) return 0.0f;
  if (x >= 0 && y >= 0) return (x == 0.0f) ? 1.5707963267948966f : det_atanf(y / x);
  if (x < 0 && y >= 0) return AKZ_PI_F - det_atanf(-y / x);
  if (x < 0 && y < 0) return AKZ_PI_F + det_atanf(y / x);
  return (x == 0.0f) ? (2.0f * AKZ_PI_F - 1.5707963267948966f) : 2.0f * AKZ_PI_F - det_atanf(-y / x);
}

static void det_sincosf(float a, float *s, float *c) { /* a in [0, 2 pi) */
  const int k = (int)(a * 0.6366197723675814f + 0.5f);
  const float kf = (float)k;
  float r = a - kf * 1.5707963705062866f; /* pi/2 split in two floats */
  r = r + kf * 4.371139000186241e-08f;
  const float z = r * r;
  float sp = -1.9515295891e-4f * z + 8.3321608736e-3f;
  sp = sp * z - 1.6666654611e-1f;
  sp = sp * z;
  sp = sp * r + r;
  float cp = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
  cp = cp * z + 4.166664568298827e-2f;
  cp = cp * z;
  cp = cp * z;
  cp = cp - 0.5f * z;
  cp = cp + 1.0f;
  switch (k & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
  }
}
void orc_akaze_math(float x, float y, float *out3) {
  out3[0] = get_angle(x, y);
  det_sincosf(out3[0], &out3[1], &out3[2]);
}

/* ---- image operators ------------------------------------------------------------------------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int reflect101(int v, int n) {
  if (n == 1) return 0;
  while (v < 0 || v >= n) v = v < 0 ? -v : 2 * (n - 1) - v;
  return v;
}

/* separable gaussian, BORDER_REPLICATE; sums run k = 0..ksize-1 in order */
static void gauss_blur(const float *src, float *dst, float *tmp, int w, int h, const float *k, int ksize) {
  const int r = ksize / 2;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float acc = 0.0f;
      for (int i = 0; i < ksize; ++i) {
        const float v = src[(size_t)y * w + clampi(x + i - r, 0, w - 1)];
        acc = acc + k[i] * v;
      }
      tmp[(size_t)y * w + x] = acc;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float acc = 0.0f;
      for (int i = 0; i < ksize; ++i) {
        const float v = tmp[(size_t)clampi(y + i - r, 0, h - 1) * w + x];
        acc = acc + k[i] * v;
      }
      dst[(size_t)y * w + x] = acc;
    }
}

/* Scharr-type derivative at spacing `scale` (3 taps at -scale, 0, +scale), BORDER_REFLECT_101:
 *   x-derivative: d(y,x) = wm*(r(y)) + ws*(r(y-s) + r(y+s)),  r(y,x) = src(y,x+s) - src(y,x-s)
 * plain cv::Scharr is scale = 1 with ws = 3, wm = 10; the multiscale form has ws = norm, wm = (10/3)*norm,
 * norm = 1/(2*scale*(10/3+2)) (compute_derivative_kernels). */
static void scharr(const float *src, float *dst, int w, int h, int xorder, int scale, float ws, float wm) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float d;
      if (xorder) {
        const int xm = reflect101(x - scale, w), xp = reflect101(x + scale, w);
        const int ym = reflect101(y - scale, h), yp = reflect101(y + scale, h);
        const float r0 = src[(size_t)y * w + xp] - src[(size_t)y * w + xm];
        const float rm = src[(size_t)ym * w + xp] - src[(size_t)ym * w + xm];
        const float rp = src[(size_t)yp * w + xp] - src[(size_t)yp * w + xm];
        d = wm * r0 + ws * (rm + rp);
      } else {
        const int xm = reflect101(x - scale, w), xp = reflect101(x + scale, w);
        const int ym = reflect101(y - scale, h), yp = reflect101(y + scale, h);
        const float r0 = src[(size_t)yp * w + x] - src[(size_t)ym * w + x];
        const float rm = src[(size_t)yp * w + xm] - src[(size_t)ym * w + xm];
        const float rp = src[(size_t)yp * w + xp] - src[(size_t)ym * w + xp];
        d = wm * r0 + ws * (rm + rp);
      }
      dst[(size_t)y * w + x] = d;
    }
}

/* cv::resize(INTER_AREA) for the octave change.  BUILD-DEFINED for sizes that are not exact halves: source pixel
 * range of a destination pixel is [d*sx, (d+1)*sx) with fractional end weights, accumulated row-major. */
static void halfsample(const float *src, int sw, int sh, float *dst, int dw, int dh) {
  const double sx = (double)sw / dw, sy = (double)sh / dh;
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      const double fx0 = x * sx, fx1 = (x + 1) * sx, fy0 = y * sy, fy1 = (y + 1) * sy;
      const int ix0 = (int)floor(fx0), iy0 = (int)floor(fy0);
      const int ix1 = (int)ceil(fx1), iy1 = (int)ceil(fy1);
      float acc = 0.0f;
      for (int yy = iy0; yy < iy1 && yy < sh; ++yy) {
        const double wy = fmin(fy1, yy + 1.0) - fmax(fy0, (double)yy);
        for (int xx = ix0; xx < ix1 && xx < sw; ++xx) {
          const double wx = fmin(fx1, xx + 1.0) - fmax(fx0, (double)xx);
          const float wgt = (float)(wx * wy / (sx * sy));
          acc = acc + wgt * src[(size_t)yy * sw + xx];
        }
      }
      dst[(size_t)y * dw + x] = acc;
    }
}

/* compute_k_percentile(img, 0.7, gscale = 1, nbins = 300) */
static float k_percentile(const float *img, int w, int h, const akz_plan *P, float *b0, float *b1, float *b2, float *b3) {
  gauss_blur(img, b0, b3, w, h, P->g10, 5);
  scharr(b0, b1, w, h, 1, 1, 3.0f, 10.0f);
  scharr(b0, b2, w, h, 0, 1, 3.0f, 10.0f);
  float hmax = 0.0f;
  for (int y = 1; y < h - 1; ++y)
    for (int x = 1; x < w - 1; ++x) {
      const float lx = b1[(size_t)y * w + x], ly = b2[(size_t)y * w + x];
      const float m = sqrtf(lx * lx + ly * ly);
      if (m > hmax) hmax = m;
    }
  int hist[300];
  memset(hist, 0, sizeof(hist));
  int npoints = 0;
  for (int y = 1; y < h - 1; ++y)
    for (int x = 1; x < w - 1; ++x) {
      const float lx = b1[(size_t)y * w + x], ly = b2[(size_t)y * w + x];
      const float m = sqrtf(lx * lx + ly * ly);
      if (m != 0.0f) {
        int nbin = (int)floorf(300.0f * (m / hmax));
        if (nbin == 300) nbin--;
        hist[nbin]++;
        npoints++;
      }
    }
  const int nthreshold = (int)((float)npoints * 0.7f);
  int nelements = 0, k = 0;
  for (k = 0; nelements < nthreshold && k < 300; k++) nelements += hist[k];
  if (nelements < nthreshold) return 0.03f;
  return hmax * ((float)k / 300.0f);
}

typedef struct {
  float *Lt, *Lsmooth, *Lx, *Ly, *Lxx, *Lxy, *Lyy, *Ldet;
} akz_bufs;

static void build_scale_space(const uint8_t *gray, int w, int h, const akz_plan *P, akz_bufs *B) {
  const size_t n0 = (size_t)w * h;
  float *img = (float *)malloc(n0 * sizeof(float));
  float *t0 = (float *)malloc(n0 * sizeof(float)), *t1 = (float *)malloc(n0 * sizeof(float));
  float *t2 = (float *)malloc(n0 * sizeof(float)), *t3 = (float *)malloc(n0 * sizeof(float));
  for (size_t i = 0; i < n0; ++i) img[i] = (float)gray[i] / 255.0f; /* convertTo(CV_32F, 1/255) */
  for (int i = 0; i < P->nlev; ++i) {
    const size_t n = (size_t)P->lev[i].w * P->lev[i].h;
    B[i].Lt = (float *)malloc(n * sizeof(float));
    B[i].Lsmooth = (float *)malloc(n * sizeof(float));
    B[i].Lx = (float *)malloc(n * sizeof(float));
    B[i].Ly = (float *)malloc(n * sizeof(float));
    B[i].Lxx = (float *)malloc(n * sizeof(float));
    B[i].Lxy = (float *)malloc(n * sizeof(float));
    B[i].Lyy = (float *)malloc(n * sizeof(float));
    B[i].Ldet = (float *)malloc(n * sizeof(float));
  }
  gauss_blur(img, B[0].Lt, t3, w, h, P->g16, 9);
  memcpy(B[0].Lsmooth, B[0].Lt, n0 * sizeof(float));
  float kcontrast = k_percentile(img, w, h, P, t0, t1, t2, t3);
  for (int i = 1; i < P->nlev; ++i) {
    const akz_level *L = &P->lev[i], *Lp = &P->lev[i - 1];
    const int lw = L->w, lh = L->h;
    const size_t n = (size_t)lw * lh;
    if (L->octave > Lp->octave) {
      halfsample(B[i - 1].Lt, Lp->w, Lp->h, B[i].Lt, lw, lh);
      kcontrast = kcontrast * 0.75f;
    } else {
      memcpy(B[i].Lt, B[i - 1].Lt, n * sizeof(float));
    }
    gauss_blur(B[i].Lt, B[i].Lsmooth, t3, lw, lh, P->g10, 5);
    scharr(B[i].Lsmooth, t0, lw, lh, 1, 1, 3.0f, 10.0f);
    scharr(B[i].Lsmooth, t1, lw, lh, 0, 1, 3.0f, 10.0f);
    const float inv_k = 1.0f / (kcontrast * kcontrast);
    for (size_t p = 0; p < n; ++p) t2[p] = 1.0f / (1.0f + inv_k * (t0[p] * t0[p] + t1[p] * t1[p])); /* pm_g2 */
    float *Ld = B[i].Lt;
    for (int sidx = 0; sidx < L->nsteps; ++sidx) {
      const float step = 0.5f * L->tsteps[sidx];
      /* nld_step_scalar, zero flux across the image border (BUILD-DEFINED at the four corners) */
      for (int y = 0; y < lh; ++y)
        for (int x = 0; x < lw; ++x) {
          const size_t p = (size_t)y * lw + x;
          const float c = t2[p], v = Ld[p];
          float xpos = 0.0f, xneg = 0.0f, ypos = 0.0f, yneg = 0.0f;
          if (x + 1 < lw) xpos = (c + t2[p + 1]) * (Ld[p + 1] - v);
          if (x > 0) xneg = (t2[p - 1] + c) * (v - Ld[p - 1]);
          if (y + 1 < lh) ypos = (c + t2[p + lw]) * (Ld[p + lw] - v);
          if (y > 0) yneg = (t2[p - lw] + c) * (v - Ld[p - lw]);
          t3[p] = step * (((xpos - xneg) + ypos) - yneg);
        }
      for (size_t p = 0; p < n; ++p) Ld[p] = Ld[p] + t3[p];
    }
  }
  /* Compute_Multiscale_Derivatives + Compute_Determinant_Hessian_Response */
  for (int i = 0; i < P->nlev; ++i) {
    const akz_level *L = &P->lev[i];
    const int lw = L->w, lh = L->h, s = L->sigma_size;
    const size_t n = (size_t)lw * lh;
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (float)s * (wgt + 2.0f));
    const float ws = norm, wm = wgt * norm;
    scharr(B[i].Lsmooth, B[i].Lx, lw, lh, 1, s, ws, wm);
    scharr(B[i].Lsmooth, B[i].Ly, lw, lh, 0, s, ws, wm);
    scharr(B[i].Lx, B[i].Lxx, lw, lh, 1, s, ws, wm);
    scharr(B[i].Ly, B[i].Lyy, lw, lh, 0, s, ws, wm);
    scharr(B[i].Lx, B[i].Lxy, lw, lh, 0, s, ws, wm);
    const float sf = (float)s, sf2 = (float)(s * s);
    for (size_t p = 0; p < n; ++p) {
      B[i].Lx[p] = B[i].Lx[p] * sf;
      B[i].Ly[p] = B[i].Ly[p] * sf;
      B[i].Lxx[p] = B[i].Lxx[p] * sf2;
      B[i].Lxy[p] = B[i].Lxy[p] * sf2;
      B[i].Lyy[p] = B[i].Lyy[p] * sf2;
      B[i].Ldet[p] = B[i].Lxx[p] * B[i].Lyy[p] - B[i].Lxy[p] * B[i].Lxy[p];
    }
  }
  free(img);
  free(t0);
  free(t1);
  free(t2);
  free(t3);
}

static void free_bufs(const akz_plan *P, akz_bufs *B) {
  for (int i = 0; i < P->nlev; ++i) {
    free(B[i].Lt);
    free(B[i].Lsmooth);
    free(B[i].Lx);
    free(B[i].Ly);
    free(B[i].Lxx);
    free(B[i].Lxy);
    free(B[i].Lyy);
    free(B[i].Ldet);
  }
}

typedef struct {
  float x, y, size, angle, response;
  int octave, class_id;
} akz_kpt;

/* Find_Scale_Space_Extrema: candidates in (level, row, column) order, then OpenCV's two sequential duplicate passes */
static int find_extrema(const akz_plan *P, const akz_bufs *B, float dthreshold, akz_kpt **out) {
  const float smax = 10.0f * sqrtf(2.0f);
  int cap = 4096, n = 0;
  akz_kpt *aux = (akz_kpt *)malloc((size_t)cap * sizeof(akz_kpt));
  for (int i = 0; i < P->nlev; ++i) {
    const akz_level *L = &P->lev[i];
    const int lw = L->w, lh = L->h;
    const float *D = B[i].Ldet;
    for (int iy = 1; iy < lh - 1; ++iy)
      for (int jx = 1; jx < lw - 1; ++jx) {
        const float v = D[(size_t)iy * lw + jx];
        if (!(v > dthreshold && v >= 0.00001f)) continue;
        if (!(v > D[(size_t)iy * lw + jx - 1] && v > D[(size_t)iy * lw + jx + 1] &&
              v > D[(size_t)(iy - 1) * lw + jx - 1] && v > D[(size_t)(iy - 1) * lw + jx] &&
              v > D[(size_t)(iy - 1) * lw + jx + 1] && v > D[(size_t)(iy + 1) * lw + jx - 1] &&
              v > D[(size_t)(iy + 1) * lw + jx] && v > D[(size_t)(iy + 1) * lw + jx + 1]))
          continue;
        akz_kpt pt;
        pt.response = fabsf(v);
        pt.size = L->esigma * 1.5f;
        pt.octave = L->octave;
        pt.class_id = i;
        pt.angle = 0.0f;
        const float ratio = (float)(1 << L->octave);
        const int sigma_size_ = fround_(pt.size / ratio);
        pt.x = (float)jx;
        pt.y = (float)iy;
        int is_extremum = 1, is_repeated = 0, id_repeated = 0;
        for (int ik = 0; ik < n; ++ik) {
          if (pt.class_id - 1 == aux[ik].class_id || pt.class_id == aux[ik].class_id) {
            const float dx = pt.x * ratio - aux[ik].x, dy = pt.y * ratio - aux[ik].y;
            const float dist = dx * dx + dy * dy;
            if (dist <= pt.size * pt.size) {
              if (pt.response > aux[ik].response) {
                id_repeated = ik;
                is_repeated = 1;
              } else {
                is_extremum = 0;
              }
              break;
            }
          }
        }
        if (!is_extremum) continue;
        const int left_x = fround_(pt.x - smax * sigma_size_) - 1, right_x = fround_(pt.x + smax * sigma_size_) + 1;
        const int up_y = fround_(pt.y - smax * sigma_size_) - 1, down_y = fround_(pt.y + smax * sigma_size_) + 1;
        if (left_x < 0 || right_x >= lw || up_y < 0 || down_y >= lh) continue;
        pt.x = pt.x * ratio;
        pt.y = pt.y * ratio;
        if (!is_repeated) {
          if (n == cap) {
            cap *= 2;
            aux = (akz_kpt *)realloc(aux, (size_t)cap * sizeof(akz_kpt));
          }
          aux[n++] = pt;
        } else {
          aux[id_repeated] = pt;
        }
      }
  }
  akz_kpt *kp = (akz_kpt *)malloc((size_t)(n ? n : 1) * sizeof(akz_kpt));
  int m = 0;
  for (int i = 0; i < n; ++i) {
    int rep = 0;
    for (int j = i + 1; j < n; ++j)
      if (aux[i].class_id + 1 == aux[j].class_id) {
        const float dx = aux[i].x - aux[j].x, dy = aux[i].y - aux[j].y;
        if (dx * dx + dy * dy <= aux[i].size * aux[i].size && aux[i].response < aux[j].response) {
          rep = 1;
          break;
        }
      }
    if (!rep) kp[m++] = aux[i];
  }
  free(aux);
  *out = kp;
  return m;
}

/* Do_Subpixel_Refinement; returns 0 when the point is dropped */
static int subpixel(const akz_plan *P, const akz_bufs *B, akz_kpt *k) {
  const akz_level *L = &P->lev[k->class_id];
  const float ratio = (float)(1 << k->octave);
  const int x = fround_(k->x / ratio), y = fround_(k->y / ratio), lw = L->w;
  const float *D = B[k->class_id].Ldet;
#define DD(yy, xx) D[(size_t)(yy) * lw + (xx)]
  const float Dx = 0.5f * (DD(y, x + 1) - DD(y, x - 1));
  const float Dy = 0.5f * (DD(y + 1, x) - DD(y - 1, x));
  const float Dxx = (DD(y, x + 1) + DD(y, x - 1)) - 2.0f * DD(y, x);
  const float Dyy = (DD(y + 1, x) + DD(y - 1, x)) - 2.0f * DD(y, x);
  const float Dxy = 0.25f * (DD(y + 1, x + 1) + DD(y - 1, x - 1)) - 0.25f * (DD(y - 1, x + 1) + DD(y + 1, x - 1));
#undef DD
  /* solve [Dxx Dxy; Dxy Dyy] d = -[Dx Dy] (BUILD-DEFINED: Cramer's rule instead of cv::solve's LU) */
  const float det = Dxx * Dyy - Dxy * Dxy;
  if (det == 0.0f) return 0;
  const float d0 = (-Dx * Dyy + Dy * Dxy) / det;
  const float d1 = (-Dy * Dxx + Dx * Dxy) / det;
  if (fabsf(d0) <= 1.0f && fabsf(d1) <= 1.0f) {
    k->x = ((float)x + d0) * ratio;
    k->y = ((float)y + d1) * ratio;
    k->angle = 0.0f;
    k->size = k->size * 2.0f; /* OpenCV keypoint size is a diameter */
    return 1;
  }
  return 0;
}

static inline float at_clamped(const float *img, int w, int h, int y, int x) {
  /* BUILD-DEFINED: OpenCV 3.0 reads unchecked; samples outside the level are clamped to its border */
  return img[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}

static void main_orientation(const akz_plan *P, const akz_bufs *B, akz_kpt *k) {
  static const int id[13] = {6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6};
  float resX[109], resY[109], Ang[109];
  const int level = k->class_id;
  const akz_level *L = &P->lev[level];
  const float ratio = (float)(1 << L->octave);
  const int s = fround_(0.5f * k->size / ratio);
  const float xf = k->x / ratio, yf = k->y / ratio;
  int idx = 0;
  for (int i = -6; i <= 6; ++i)
    for (int j = -6; j <= 6; ++j)
      if (i * i + j * j < 36) {
        const int iy = fround_(yf + (float)(j * s)), ix = fround_(xf + (float)(i * s));
        const float g = P->gauss25[id[i + 6]][id[j + 6]];
        resX[idx] = g * at_clamped(B[level].Lx, L->w, L->h, iy, ix);
        resY[idx] = g * at_clamped(B[level].Ly, L->w, L->h, iy, ix);
        Ang[idx] = get_angle(resX[idx], resY[idx]);
        ++idx;
      }
  float maxv = 0.0f;
  const float two_pi = 2.0f * AKZ_PI_F;
  for (float ang1 = 0.0f; ang1 < two_pi; ang1 += 0.15f) {
    const float ang2 = (ang1 + AKZ_PI_F / 3.0f > two_pi) ? ang1 - 5.0f * AKZ_PI_F / 3.0f : ang1 + AKZ_PI_F / 3.0f;
    float sumX = 0.0f, sumY = 0.0f;
    for (int q = 0; q < 109; ++q) {
      const float ang = Ang[q];
      if (ang1 < ang2 && ang1 < ang && ang < ang2) {
        sumX += resX[q];
        sumY += resY[q];
      } else if (ang2 < ang1 && ((ang > 0 && ang < ang2) || (ang > ang1 && ang < two_pi))) {
        sumX += resX[q];
        sumY += resY[q];
      }
    }
    const float mag = sumX * sumX + sumY * sumY;
    if (mag > maxv) {
      maxv = mag;
      k->angle = get_angle(sumX, sumY);
    }
  }
}

/* Get_MLDB_Full_Descriptor, 3 channels, pattern size 10 -> 486 bits in 61 bytes */
static void mldb_descriptor(const akz_plan *P, const akz_bufs *B, const akz_kpt *k, uint8_t *desc) {
  const int level = k->class_id;
  const akz_level *L = &P->lev[level];
  const int lw = L->w, lh = L->h;
  const float ratio = (float)(1 << L->octave);
  const int scale = fround_(0.5f * k->size / ratio);
  const float xf = k->x / ratio, yf = k->y / ratio;
  float si, co;
  det_sincosf(k->angle, &si, &co);
  const int pattern = 10;
  const int sample_step[3] = {pattern, (pattern * 2 + 2) / 3, pattern / 2}; /* 10, ceil(20/3)=7, 5 */
  memset(desc, 0, 61);
  int dpos = 0;
  for (int lvl = 0; lvl < 3; ++lvl) {
    const int step = sample_step[lvl];
    float values[16][3];
    int count = 0;
    for (int i = -pattern; i < pattern; i += step)
      for (int j = -pattern; j < pattern; j += step) {
        float di = 0.0f, dx = 0.0f, dy = 0.0f;
        int ns = 0;
        for (int kk = i; kk < i + step; ++kk)
          for (int l = j; l < j + step; ++l) {
            const float sample_y = yf + ((float)l * co * (float)scale + (float)kk * si * (float)scale);
            const float sample_x = xf + (-(float)l * si * (float)scale + (float)kk * co * (float)scale);
            const int y1 = fround_(sample_y), x1 = fround_(sample_x);
            const float ri = at_clamped(B[level].Lt, lw, lh, y1, x1);
            const float rx = at_clamped(B[level].Lx, lw, lh, y1, x1);
            const float ry = at_clamped(B[level].Ly, lw, lh, y1, x1);
            di += ri;
            const float rry = rx * co + ry * si;
            const float rrx = -rx * si + ry * co;
            dx += rrx;
            dy += rry;
            ns++;
          }
        values[count][0] = di / (float)ns;
        values[count][1] = dx / (float)ns;
        values[count][2] = dy / (float)ns;
        count++;
      }
    for (int pos = 0; pos < 3; ++pos)
      for (int a = 0; a < count; ++a)
        for (int b = a + 1; b < count; ++b) {
          if (values[a][pos] > values[b][pos]) desc[dpos >> 3] |= (uint8_t)(1 << (dpos & 7));
          dpos++;
        }
  }
}

/*
 * detectAndCompute.  kpts_out [cap x 6] = x, y, size, angle (radians), response, class_id ; desc_out [cap x 61].
 * Returns the number of keypoints (<= cap), or -needed when cap is too small.
 * levels_out (optional, may be NULL): concatenated Ldet of every level, for stage-wise comparisons.
 */
int orc_akaze_detect_and_compute(const uint8_t *gray, int w, int h, int omax, int nsublevels, float dthreshold,
                                 float *kpts_out, uint8_t *desc_out, int cap, float *ldet_out, float *lt_out) {
  akz_plan P;
  orc_akaze_plan(w, h, omax, nsublevels, &P);
  akz_bufs B[AKZ_MAX_LEVELS];
  build_scale_space(gray, w, h, &P, B);
  if (ldet_out || lt_out) {
    size_t off = 0;
    for (int i = 0; i < P.nlev; ++i) {
      const size_t n = (size_t)P.lev[i].w * P.lev[i].h;
      if (ldet_out) memcpy(ldet_out + off, B[i].Ldet, n * sizeof(float));
      if (lt_out) memcpy(lt_out + off, B[i].Lt, n * sizeof(float));
      off += n;
    }
  }
  akz_kpt *kp = NULL;
  const int n = find_extrema(&P, B, dthreshold, &kp);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    akz_kpt k = kp[i];
    if (!subpixel(&P, B, &k)) continue;
    if (m < cap) {
      main_orientation(&P, B, &k);
      mldb_descriptor(&P, B, &k, desc_out + (size_t)m * 61);
      float *o = kpts_out + (size_t)m * 6;
      o[0] = k.x;
      o[1] = k.y;
      o[2] = k.size;
      o[3] = k.angle;
      o[4] = k.response;
      o[5] = (float)k.class_id;
    }
    ++m;
  }
  free(kp);
  free_bufs(&P, B);
  return m <= cap ? m : -m;
}

/* compute() on given keypoints (dense BoW features): kin [n x 4] = x, y, size, class_id (octave 0) */
int orc_akaze_compute(const uint8_t *gray, int w, int h, int omax, int nsublevels, const float *kin, int n,
                      uint8_t *desc_out, float *angle_out) {
  akz_plan P;
  orc_akaze_plan(w, h, omax, nsublevels, &P);
  akz_bufs B[AKZ_MAX_LEVELS];
  build_scale_space(gray, w, h, &P, B);
  for (int i = 0; i < n; ++i) {
    akz_kpt k;
    k.x = kin[4 * i];
    k.y = kin[4 * i + 1];
    k.size = kin[4 * i + 2];
    k.class_id = clampi((int)kin[4 * i + 3], 0, P.nlev - 1);
    k.octave = P.lev[k.class_id].octave;
    k.angle = 0.0f;
    k.response = 0.0f;
    main_orientation(&P, B, &k);
    mldb_descriptor(&P, B, &k, desc_out + (size_t)i * 61);
    if (angle_out) angle_out[i] = k.angle;
  }
  free_bufs(&P, B);
  return n;
}

int orc_akaze_levels(int w, int h, int omax, int nsublevels, int *wh_out /*[32*2]*/) {
  akz_plan P;
  orc_akaze_plan(w, h, omax, nsublevels, &P);
  for (int i = 0; i < P.nlev; ++i) {
    wh_out[2 * i] = P.lev[i].w;
    wh_out[2 * i + 1] = P.lev[i].h;
  }
  return P.nlev;
}
