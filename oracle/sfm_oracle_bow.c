/*
 * sfm_oracle_bow.c -- CPU restatement of the bag-of-words view shortlist.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see sfm_oracle.c): OpenCV's FLANN KD-tree searches
 * (BoFSpatialPyramids.cpp:118-119 nearest centre, BoFUtils.cpp:46-59 k-NN over the views' .bow vectors) are
 * approximate; the oracle is the exact search they approximate, with ties to the lowest index, and float32
 * sums evaluated in ONE fixed order (written down below) that the HIP kernels reproduce.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Squared L2 distance in float32, the order both sides use: 64 partial sums, partial l takes elements
 * l, l+64, l+128, ... in index order; then a butterfly over the partials with strides 32,16,8,4,2,1
 * (partial[l] += partial[l ^ stride], both partners get the same sum). */
float orc_bow_dist(const float *a, const float *b, int dim) {
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float s = 0.0f;
    for (int i = l; i < dim; i += 64) {
      const float d = a[i] - b[i];
      const float d2 = d * d;
      s = s + d2;
    }
    part[l] = s;
  }
  for (int stride = 32; stride >= 1; stride >>= 1) {
    float nxt[64];
    for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ stride];
    memcpy(part, nxt, sizeof(part));
  }
  return part[0];
}

typedef struct {
  uint32_t bits;
  uint32_t pos;
} dist_pos;
static int cmp_dist_pos(const void *pa, const void *pb) {
  const dist_pos *a = (const dist_pos *)pa, *b = (const dist_pos *)pb;
  if (a->bits != b->bits) return a->bits < b->bits ? -1 : 1;
  return (a->pos > b->pos) - (a->pos < b->pos);
}

/* selectViewByBoF (BoFUtils.cpp:27-68) made exact: the k candidate views nearest (L2) to the query's BoW vector;
 * cand = ascending view indices (the std::set order the reference maps trainIdx back through, :62-67) or NULL
 * for all views.  Output: selected view indices, ascending (the reference returns a std::set). */
void orc_bow_select(const float *bow, int dim, uint32_t n_views, const uint32_t *cand, uint32_t n_cand,
                    const float *query, uint32_t k, uint32_t *out_sel) {
  if (!cand) n_cand = n_views;
  dist_pos *d = (dist_pos *)malloc((size_t)n_cand * sizeof(dist_pos));
  for (uint32_t p = 0; p < n_cand; ++p) {
    const uint32_t v = cand ? cand[p] : p;
    const float x = orc_bow_dist(bow + (size_t)v * dim, query, dim);
    uint32_t bits;
    memcpy(&bits, &x, 4);
    d[p].bits = bits; /* non-negative floats order like their bit patterns */
    d[p].pos = p;
  }
  qsort(d, n_cand, sizeof(dist_pos), cmp_dist_pos);
  if (k > n_cand) k = n_cand;
  uint32_t *sel = (uint32_t *)malloc((size_t)(k ? k : 1) * sizeof(uint32_t));
  for (uint32_t i = 0; i < k; ++i) sel[i] = d[i].pos;
  /* ascending position = ascending view index */
  for (uint32_t i = 1; i < k; ++i) {
    uint32_t v = sel[i];
    int j = (int)i - 1;
    while (j >= 0 && sel[j] > v) {
      sel[j + 1] = sel[j];
      --j;
    }
    sel[j + 1] = v;
  }
  for (uint32_t i = 0; i < k; ++i) out_sel[i] = cand ? cand[sel[i]] : sel[i];
  free(sel);
  free(d);
}

/*
 * The query's BoW vector from its dense local features:
 *   PcaWrapper::calcPcaProject (PcaWrapper.cpp:67-89): (x - mean) . eigvec_d, first n_pca components, each
 *     DIVIDED BY ITS EIGENVALUE (not its root);  float32, sequential over the input dimension;
 *   BoFSpatialPyramids::calcBoF (BoFSpatialPyramids.cpp:108-302): nearest centre (exact here), spatial-pyramid
 *     counts (level 0: whole image; level 1: 2x2 cells of edge R/2, cell = 1 + cy*2 + cx; level 2: 3 stripes),
 *     divide by the number of descriptors, then per cell L1-sqrt (norm 2), L2 (norm 1) or nothing (norm 0).
 * n_pca = 0 skips the projection (bofPyramids->calcBoF(descriptors, ...), localization.cpp:356-361).
 * out: K * cells doubles.
 */
int orc_bof_cells(int levels) {
  int c = 0;
  for (int l = 0; l < levels; ++l) c += (l == 0) ? 1 : (l == 2 ? 3 : (l + 1) * (l + 1));
  return c;
}

void orc_bof(const float *desc, const float *kxy, int n, int in_dim, const float *pca_mean, const float *pca_evec,
             const float *pca_eval, int n_pca, const float *centers, int K, int resized, int levels, int norm_type,
             double *out) {
  const int cdim = n_pca > 0 ? n_pca : in_dim;
  const int cells = orc_bof_cells(levels);
  for (int i = 0; i < K * cells; ++i) out[i] = 0.0;
  float *y = (float *)malloc((size_t)cdim * sizeof(float));
  for (int r = 0; r < n; ++r) {
    const float *x = desc + (size_t)r * in_dim;
    if (n_pca > 0) {
      for (int d = 0; d < n_pca; ++d) {
        float acc = 0.0f;
        for (int i = 0; i < in_dim; ++i) {
          const float c = x[i] - pca_mean[i];
          const float pr = c * pca_evec[(size_t)d * in_dim + i];
          acc = acc + pr;
        }
        y[d] = acc / pca_eval[d];
      }
    } else {
      memcpy(y, x, (size_t)in_dim * sizeof(float));
    }
    int best = 0;
    float bestd = INFINITY;
    for (int c = 0; c < K; ++c) {
      float s = 0.0f;
      for (int i = 0; i < cdim; ++i) {
        const float d = y[i] - centers[(size_t)c * cdim + i];
        const float d2 = d * d;
        s = s + d2;
      }
      if (s < bestd) {
        bestd = s;
        best = c;
      }
    }
    const float px = kxy[2 * r], py = kxy[2 * r + 1];
    int cell0 = 0;
    for (int level = 0; level < levels; ++level) {
      const int len = level + 1;
      const int edge = resized / len;
      if (level == 2) {
        for (int cy = 0; cy < len; ++cy)
          if (px >= 0 && px < (float)resized && py >= (float)(edge * cy) && py < (float)(edge * (cy + 1)))
            out[(size_t)K * (cell0 + cy) + best] += 1.0;
        cell0 += 3;
      } else {
        for (int cx = 0; cx < len; ++cx)
          for (int cy = 0; cy < len; ++cy)
            if (px >= (float)(edge * cx) && px < (float)(edge * (cx + 1)) && py >= (float)(edge * cy) &&
                py < (float)(edge * (cy + 1)))
              out[(size_t)K * (cell0 + cy * len + cx) + best] += 1.0;
        cell0 += len * len;
      }
    }
  }
  free(y);
  for (int i = 0; i < K * cells; ++i) out[i] = out[i] / (double)n;
  for (int c = 0; c < cells; ++c) {
    double *h = out + (size_t)K * c;
    if (norm_type == 2) { /* L1_NORM_SQUARE_ROOT */
      double norm = 0.0;
      for (int i = 0; i < K; ++i) norm += h[i];
      if (norm > 0.0)
        for (int i = 0; i < K; ++i) h[i] = sqrt(h[i] / norm);
    } else if (norm_type == 1) { /* L2_NORM */
      double norm = 0.0;
      for (int i = 0; i < K; ++i) norm += h[i] * h[i];
      norm = sqrt(norm);
      if (norm > 0.0)
        for (int i = 0; i < K; ++i) h[i] = h[i] / norm;
    }
  }
}

/* ---------------------------------------------------------------------------------------------------------
 * Dense-BoW front end (DenseLocalFeatureWrapper.cpp:89-99): restatement of OpenCV 3.0's 8-bit
 * cv::resize(INTER_CUBIC) as its two passes (rows to an int buffer, then columns), BGR2GRAY and
 * normalize(0, 255, NORM_MINMAX).  OpenCV is not available here: parity with it is unpinned.
 * ------------------------------------------------------------------------------------------------------- */
static void orc_cubic_coeffs(float x, float *c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

static short orc_sat_short(float v) {
  const float r = nearbyintf(v);
  return (short)(r > 32767.f ? 32767.f : (r < -32768.f ? -32768.f : r));
}

static int orc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_dense_gray(const uint8_t *bgr, int w, int h, int size, uint8_t *gray) {
  int *xofs = (int *)malloc(sizeof(int) * size), *yofs = (int *)malloc(sizeof(int) * size);
  short *xa = (short *)malloc(sizeof(short) * 4 * size), *ya = (short *)malloc(sizeof(short) * 4 * size);
  const double scale_x = 1.0 / ((double)size / (double)w), scale_y = 1.0 / ((double)size / (double)h);
  for (int d = 0; d < size; ++d) {
    float c[4];
    float fx = (float)((d + 0.5) * scale_x - 0.5);
    int s = (int)floorf(fx);
    fx -= (float)s;
    orc_cubic_coeffs(fx, c);
    xofs[d] = s;
    for (int k = 0; k < 4; ++k) xa[4 * d + k] = orc_sat_short(c[k] * 2048.f);
    float fy = (float)((d + 0.5) * scale_y - 0.5);
    s = (int)floorf(fy);
    fy -= (float)s;
    orc_cubic_coeffs(fy, c);
    yofs[d] = s;
    for (int k = 0; k < 4; ++k) ya[4 * d + k] = orc_sat_short(c[k] * 2048.f);
  }
  /* pass 1: every source row resampled horizontally (int, unshifted) */
  int *rows = (int *)malloc(sizeof(int) * (size_t)h * size * 3);
  for (int y = 0; y < h; ++y)
    for (int d = 0; d < size; ++d)
      for (int c = 0; c < 3; ++c) {
        int acc = 0;
        for (int k = 0; k < 4; ++k) {
          const int x = orc_clampi(xofs[d] - 1 + k, 0, w - 1);
          acc += (int)bgr[((size_t)y * w + x) * 3 + c] * (int)xa[4 * d + k];
        }
        rows[((size_t)y * size + d) * 3 + c] = acc;
      }
  /* pass 2: columns, rounding shift by 22, saturate; then gray */
  int gmin = 255, gmax = 0;
  for (int dy = 0; dy < size; ++dy)
    for (int dx = 0; dx < size; ++dx) {
      int ch[3];
      for (int c = 0; c < 3; ++c) {
        int acc = 0;
        for (int k = 0; k < 4; ++k) {
          const int y = orc_clampi(yofs[dy] - 1 + k, 0, h - 1);
          acc += rows[((size_t)y * size + dx) * 3 + c] * (int)ya[4 * dy + k];
        }
        ch[c] = orc_clampi((acc + (1 << 21)) >> 22, 0, 255);
      }
      const int g = (ch[0] * 1868 + ch[1] * 9617 + ch[2] * 4899 + 8192) >> 14;
      gray[(size_t)dy * size + dx] = (uint8_t)g;
      if (g < gmin) gmin = g;
      if (g > gmax) gmax = g;
    }
  const double smin = gmin, smax = gmax;
  const double scale_d = 255.0 * ((smax - smin > 2.220446049250313e-16) ? 1.0 / (smax - smin) : 0.0);
  const float scale = (float)scale_d, shift = (float)(0.0 - smin * scale_d);
  for (size_t i = 0; i < (size_t)size * size; ++i) {
    const float v = (float)gray[i] * scale + shift;
    const float r = nearbyintf(v);
    gray[i] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
  }
  free(rows);
  free(xofs);
  free(yofs);
  free(xa);
  free(ya);
}
