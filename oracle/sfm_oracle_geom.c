/*
 * sfm_oracle_geom.c -- CPU restatement of the geometric half of the query path:
 *   A8  geometricMatch  -> OpenMVG GeometricFilter_FMatrix_AC (7-point + AC-RANSAC)   MatchUtils.cpp:372-420
 *   A9  matchProviderToMatchSet                                                       SfMDataUtils.cpp:59-125
 *   A11 SfM_Localizer::Localize (Kneip P3P + AC-RANSAC)                               localization.cpp:504-509
 *   A12 KRt_From_P, t_out = -R^T t                                                    localization.cpp:544-547
 *
 * TEST INFRASTRUCTURE ONLY (see sfm_oracle.c).  PARITY UNPINNED: OpenMVG 1.1 is not in /root/reference nor
 * in this image.  What follows restates the published algorithms -- AC-RANSAC (Moisan, Moulon, Monasse,
 * IPOL 2012), the 7-point fundamental solver (Hartley & Zisserman 11.1.2), P3P (Kneip, Scaramuzza,
 * Siegwart, CVPR 2011), RQ by Givens rotations (H&Z A4.1.1) -- in the shape of the reference's call
 * sites.  Three things are build-defined because the reference leaves them undefined or irreproducible:
 *   1. random samples come from a counter-based generator (Philox4x32-10) keyed by (seed, stage, stream,
 *      iteration), not from the unseeded rand() OpenMVG 1.1 uses;
 *   2. transcendental-free numerics: log10 is a fixed-order series, polynomial roots come from safeguarded
 *      Newton + deflation / Ferrari with a Newton resolvent root, so that the HIP kernels (compiled with
 *      -ffp-contract=off) can reproduce every double bit for bit;
 *   3. NaN residuals (degenerate models) sort as +inf.
 * The sequential semantics of OpenMVG's ACRANSAC loop (first meaningful model switches sampling to its
 * inliers and cuts the budget to the reserved 10 %) are kept exactly.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* deterministic scalar helpers                                                                      */
/* ------------------------------------------------------------------------------------------------ */
static inline uint64_t d2u(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
static inline double u2d(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}
static inline int is_nan(double x) { return (d2u(x) & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull; }
static inline int is_inf(double x) { return (d2u(x) & 0x7FFFFFFFFFFFFFFFull) == 0x7FF0000000000000ull; }
static inline double pos_inf(void) { return u2d(0x7FF0000000000000ull); }
static inline double q_nan(void) { return u2d(0x7FF8000000000000ull); }

/* log10 by a fixed operation order: x = m 2^e, m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m-1)/(m+1)). */
double orc_det_log10(double x) {
  if (is_nan(x) || x < 0.0) return q_nan();
  if (x == 0.0) return -pos_inf();
  if (is_inf(x)) return pos_inf();
  uint64_t u = d2u(x);
  int e = (int)((u >> 52) & 0x7FF);
  if (e == 0) { /* subnormal */
    x = x * 18014398509481984.0; /* 2^54 */
    u = d2u(x);
    e = (int)((u >> 52) & 0x7FF) - 54;
  }
  e -= 1023;
  double m = u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    e += 1;
  }
  const double z = (m - 1.0) / (m + 1.0);
  const double z2 = z * z;
  double p = 1.0 / 23.0;
  p = p * z2 + 1.0 / 21.0;
  p = p * z2 + 1.0 / 19.0;
  p = p * z2 + 1.0 / 17.0;
  p = p * z2 + 1.0 / 15.0;
  p = p * z2 + 1.0 / 13.0;
  p = p * z2 + 1.0 / 11.0;
  p = p * z2 + 1.0 / 9.0;
  p = p * z2 + 1.0 / 7.0;
  p = p * z2 + 1.0 / 5.0;
  p = p * z2 + 1.0 / 3.0;
  p = p * z2 + 1.0;
  const double lnm = (2.0 * z) * p;
  const double ln = (double)e * 0.6931471805599453 + lnm;
  return ln * 0.4342944819032518;
}

/* Philox4x32-10 (Salmon et al., SC'11) */
void orc_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

enum { STAGE_FMATRIX = 1, STAGE_P3P = 2 };

/* i-th 32-bit draw of (stage, stream, iteration) */
static uint32_t ac_draw(uint64_t seed, uint32_t stage, uint32_t stream, uint32_t iter, uint32_t i) {
  uint32_t c[4] = {iter, stream, i >> 2, stage};
  orc_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return c[i & 3];
}

/* OpenMVG UniformSample/random_sample: X distinct positions in [0,n), kept sorted, then mapped through
 * vec_index.  rand()>>3 is replaced by the Philox draw. */
static void ac_sample(int X, const int32_t *vec_index, int n, uint64_t seed, uint32_t stage, uint32_t stream,
                      uint32_t iter, int32_t *samples) {
  int32_t s[8];
  for (int i = 0; i < X; ++i) {
    int32_t r = (int32_t)(ac_draw(seed, stage, stream, iter, (uint32_t)i) % (uint32_t)(n - i));
    int j;
    for (j = 0; j < i && r >= s[j]; ++j) ++r;
    for (int k = i; k > j; --k) s[k] = s[k - 1];
    s[j] = r;
  }
  for (int i = 0; i < X; ++i) samples[i] = vec_index[s[i]];
}
void orc_ac_sample(int X, const int32_t *vec_index, int n, uint64_t seed, uint32_t stage, uint32_t stream,
                   uint32_t iter, int32_t *samples) {
  ac_sample(X, vec_index, n, seed, stage, stream, iter, samples);
}

static inline double dabs(double x) { return x < 0.0 ? -x : x; }
static inline double dmax(double a, double b) { return a > b ? a : b; }

/* One real root of the monic cubic x^3 + b x^2 + c x + d inside its Cauchy bracket:
 * Newton safeguarded by bisection (rtsafe), fixed operation order. */
static double cubic_one_root(double b, double c, double d) {
  const double B = 1.0 + dmax(dabs(b), dmax(dabs(c), dabs(d)));
  double xl = -B, xh = B;
  double x = -b / 3.0;
  if (!(x > xl && x < xh)) x = 0.0;
  double f = ((x + b) * x + c) * x + d;
  if (f == 0.0) return x;
  if (f < 0.0)
    xl = x;
  else
    xh = x;
  double dxold = xh - xl, dx = dxold;
  for (int it = 0; it < 200; ++it) {
    const double df = (3.0 * x + 2.0 * b) * x + c;
    if ((((x - xh) * df - f) * ((x - xl) * df - f) > 0.0) || (dabs(2.0 * f) > dabs(dxold * df))) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      x = xl + dx;
      // a bracket one ulp wide cannot shrink: its midpoint rounds to either end (to xh on a round-to-even tie,
      // which the classic `xl == x` test alone misses and then spins to the iteration cap on the same x)
      if (xl == x || xh == x) return x;
    } else {
      dxold = dx;
      dx = f / df;
      const double t = x;
      x = x - dx;
      if (t == x) return x;
    }
    f = ((x + b) * x + c) * x + d;
    if (f == 0.0) return x;
    if (f < 0.0)
      xl = x;
    else
      xh = x;
  }
  return x;
}

static double cubic_polish(double a3, double a2, double a1, double a0, double x) {
  for (int it = 0; it < 2; ++it) {
    const double f = ((a3 * x + a2) * x + a1) * x + a0;
    const double df = (3.0 * a3 * x + 2.0 * a2) * x + a1;
    if (df == 0.0) break;
    x = x - f / df;
  }
  return x;
}

/* real roots of a3 x^3 + a2 x^2 + a1 x + a0, ascending; returns their number */
int orc_solve_cubic(double a3, double a2, double a1, double a0, double r[3]) {
  int n = 0;
  if (a3 == 0.0) {
    if (a2 == 0.0) {
      if (a1 == 0.0) return 0;
      r[0] = -a0 / a1;
      return 1;
    }
    const double disc = a1 * a1 - 4.0 * a2 * a0;
    if (disc < 0.0) return 0;
    const double sq = sqrt(disc);
    const double q = -0.5 * (a1 + (a1 < 0.0 ? -sq : sq));
    r[0] = q / a2;
    r[1] = (q != 0.0) ? a0 / q : r[0];
    n = 2;
  } else {
    const double b = a2 / a3, c = a1 / a3, d = a0 / a3;
    const double x1 = cubic_one_root(b, c, d);
    r[0] = cubic_polish(a3, a2, a1, a0, x1);
    n = 1;
    const double B1 = b + x1;
    const double C1 = c + x1 * B1;
    const double disc = B1 * B1 - 4.0 * C1;
    if (disc >= 0.0) {
      const double sq = sqrt(disc);
      const double q = -0.5 * (B1 + (B1 < 0.0 ? -sq : sq));
      const double r2 = q;
      const double r3 = (q != 0.0) ? C1 / q : 0.0;
      r[1] = cubic_polish(a3, a2, a1, a0, r2);
      r[2] = cubic_polish(a3, a2, a1, a0, r3);
      n = 3;
    }
  }
  for (int i = 1; i < n; ++i) { /* insertion sort */
    const double v = r[i];
    int j = i - 1;
    while (j >= 0 && r[j] > v) {
      r[j + 1] = r[j];
      --j;
    }
    r[j + 1] = v;
  }
  return n;
}

static double quartic_polish(const double a[5], double x) {
  for (int it = 0; it < 2; ++it) {
    const double f = (((a[0] * x + a[1]) * x + a[2]) * x + a[3]) * x + a[4];
    const double df = ((4.0 * a[0] * x + 3.0 * a[1]) * x + 2.0 * a[2]) * x + a[3];
    if (df == 0.0 || is_nan(df)) break;
    const double xn = x - f / df;
    if (is_nan(xn) || is_inf(xn)) break;
    x = xn;
  }
  return x;
}

/* Real parts of the four roots of a[0] x^4 + a[1] x^3 + a[2] x^2 + a[3] x + a[4] (Kneip's solveQuartic returns
 * the real parts too).  Ferrari with a Newton resolvent root; real roots get two Newton polish steps. */
void orc_solve_quartic_real(const double a[5], double out[4]) {
  const double b = a[1] / a[0], c = a[2] / a[0], d = a[3] / a[0], e = a[4] / a[0];
  const double b2 = b * b;
  const double p = c - 0.375 * b2;
  const double q = d - 0.5 * b * c + 0.125 * b2 * b;
  const double r = e - 0.25 * b * d + 0.0625 * b2 * c - 0.01171875 * b2 * b2;
  const double shift = -0.25 * b;
  double y[4];
  int real[4] = {0, 0, 0, 0};
  if (is_nan(p) || is_nan(q) || is_nan(r) || is_inf(p) || is_inf(q) || is_inf(r)) {
    for (int i = 0; i < 4; ++i) out[i] = q_nan();
    return;
  }
  if (q == 0.0) {
    /* biquadratic: z^2 + p z + r = 0, y = +-sqrt(z) */
    const double disc = p * p - 4.0 * r;
    if (disc >= 0.0) {
      const double sq = sqrt(disc);
      const double z1 = 0.5 * (-p + sq), z2 = 0.5 * (-p - sq);
      if (z1 >= 0.0) {
        y[0] = sqrt(z1);
        y[1] = -y[0];
        real[0] = real[1] = 1;
      } else {
        y[0] = y[1] = 0.0;
      }
      if (z2 >= 0.0) {
        y[2] = sqrt(z2);
        y[3] = -y[2];
        real[2] = real[3] = 1;
      } else {
        y[2] = y[3] = 0.0;
      }
    } else {
      /* z complex, |z| = sqrt(r): Re sqrt(z) = sqrt((|z| + Re z) / 2) */
      const double mod = sqrt(r);
      const double re = sqrt(0.5 * (mod - 0.5 * p));
      y[0] = re;
      y[1] = -re;
      y[2] = re;
      y[3] = -re;
    }
  } else {
    /* resolvent m^3 + p m^2 + (p^2/4 - r) m - q^2/8 = 0 has a root m > 0 */
    const double rb = p, rc = 0.25 * p * p - r, rd = -0.125 * q * q;
    const double B = 1.0 + dmax(dabs(rb), dmax(dabs(rc), dabs(rd)));
    double xl = 0.0, xh = B;
    double m = B;
    double f = ((m + rb) * m + rc) * m + rd;
    double dxold = xh - xl, dx = dxold;
    for (int it = 0; it < 200 && f != 0.0; ++it) {
      const double df = (3.0 * m + 2.0 * rb) * m + rc;
      if ((((m - xh) * df - f) * ((m - xl) * df - f) > 0.0) || (dabs(2.0 * f) > dabs(dxold * df))) {
        dxold = dx;
        dx = 0.5 * (xh - xl);
        m = xl + dx;
        if (xl == m || xh == m) break;  // one-ulp bracket (see cubic_one_root)
      } else {
        dxold = dx;
        dx = f / df;
        const double t = m;
        m = m - dx;
        if (t == m) break;
      }
      f = ((m + rb) * m + rc) * m + rd;
      if (f < 0.0)
        xl = m;
      else
        xh = m;
    }
    const double s = sqrt(2.0 * m);
    const double h = 0.5 * p + m;
    const double g = q / (2.0 * s);
    /* y^2 - s y + (h + g) = 0  and  y^2 + s y + (h - g) = 0 */
    const double beta[2] = {-s, s};
    const double gamma[2] = {h + g, h - g};
    for (int k = 0; k < 2; ++k) {
      const double disc = beta[k] * beta[k] - 4.0 * gamma[k];
      if (disc >= 0.0) {
        const double sq = sqrt(disc);
        const double t = -0.5 * (beta[k] + (beta[k] < 0.0 ? -sq : sq));
        y[2 * k] = t;
        y[2 * k + 1] = (t != 0.0) ? gamma[k] / t : 0.0;
        real[2 * k] = real[2 * k + 1] = 1;
      } else {
        y[2 * k] = y[2 * k + 1] = -0.5 * beta[k];
      }
    }
  }
  for (int i = 0; i < 4; ++i) {
    double x = y[i] + shift;
    if (real[i]) x = quartic_polish(a, x);
    out[i] = x;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* 7-point fundamental matrix (OpenMVG SevenPointSolver::Solve; H&Z 11.1.2)                            */
/* x1, x2: 7 x 2 (normalised) points; F: up to 3 row-major 3x3 with x2^T F x1 = 0; returns their number   */
/* ------------------------------------------------------------------------------------------------ */
static double det3c(const double *a, const double *b, const double *c) {
  /* determinant of the matrix whose COLUMNS are a, b, c */
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - b[0] * (a[1] * c[2] - a[2] * c[1]) + c[0] * (a[1] * b[2] - a[2] * b[1]);
}

int orc_seven_point(const double *x1, const double *x2, double *F) {
  double A[7][9];
  for (int i = 0; i < 7; ++i) {
    const double ax = x1[2 * i], ay = x1[2 * i + 1], bx = x2[2 * i], by = x2[2 * i + 1];
    A[i][0] = bx * ax;
    A[i][1] = bx * ay;
    A[i][2] = bx;
    A[i][3] = by * ax;
    A[i][4] = by * ay;
    A[i][5] = by;
    A[i][6] = ax;
    A[i][7] = ay;
    A[i][8] = 1.0;
  }
  int perm[9];
  for (int j = 0; j < 9; ++j) perm[j] = j;
  /* Gauss-Jordan with full pivoting -> [I7 | C] in permuted columns */
  for (int k = 0; k < 7; ++k) {
    int pi = k, pj = k;
    double best = -1.0;
    for (int i = k; i < 7; ++i)
      for (int j = k; j < 9; ++j) {
        const double v = dabs(A[i][j]);
        if (v > best) {
          best = v;
          pi = i;
          pj = j;
        }
      }
    if (!(best > 0.0)) return 0; /* rank deficient (or NaN) */
    if (pi != k)
      for (int j = 0; j < 9; ++j) {
        const double t = A[k][j];
        A[k][j] = A[pi][j];
        A[pi][j] = t;
      }
    if (pj != k) {
      for (int i = 0; i < 7; ++i) {
        const double t = A[i][k];
        A[i][k] = A[i][pj];
        A[i][pj] = t;
      }
      const int t = perm[k];
      perm[k] = perm[pj];
      perm[pj] = t;
    }
    const double piv = A[k][k];
    for (int j = k; j < 9; ++j) A[k][j] = A[k][j] / piv;
    for (int i = 0; i < 7; ++i) {
      if (i == k) continue;
      const double fct = A[i][k];
      if (fct == 0.0) continue;
      for (int j = k; j < 9; ++j) A[i][j] = A[i][j] - fct * A[k][j];
    }
  }
  double f1[9], f2[9];
  for (int i = 0; i < 7; ++i) {
    f1[perm[i]] = -A[i][7];
    f2[perm[i]] = -A[i][8];
  }
  f1[perm[7]] = 1.0;
  f1[perm[8]] = 0.0;
  f2[perm[7]] = 0.0;
  f2[perm[8]] = 1.0;
  /* det(F1 + l F2) = c0 + c1 l + c2 l^2 + c3 l^3, by multilinearity over columns */
  double a0[3] = {f1[0], f1[3], f1[6]}, a1[3] = {f1[1], f1[4], f1[7]}, a2[3] = {f1[2], f1[5], f1[8]};
  double b0[3] = {f2[0], f2[3], f2[6]}, b1[3] = {f2[1], f2[4], f2[7]}, b2[3] = {f2[2], f2[5], f2[8]};
  const double c0 = det3c(a0, a1, a2);
  const double c1 = (det3c(b0, a1, a2) + det3c(a0, b1, a2)) + det3c(a0, a1, b2);
  const double c2 = (det3c(a0, b1, b2) + det3c(b0, a1, b2)) + det3c(b0, b1, a2);
  const double c3 = det3c(b0, b1, b2);
  double roots[3];
  const int n = orc_solve_cubic(c3, c2, c1, c0, roots);
  for (int s = 0; s < n; ++s)
    for (int k = 0; k < 9; ++k) F[9 * s + k] = f1[k] + roots[s] * f2[k];
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* P3P (Kneip et al. 2011), as OpenMVG's P3PSolver::Solve wraps it: input 3 normalised image points   */
/* (K^-1 x) and 3 world points; output 4 [R|t] (3x4 row-major), always 4 (real parts of the roots).    */
/* Returns 0 when the world points are collinear, else 4.                                              */
/* ------------------------------------------------------------------------------------------------ */
static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static double norm3(const double a[3]) { return sqrt(dot3(a, a)); }
static void normalize3(double a[3]) {
  const double n = norm3(a);
  a[0] = a[0] / n;
  a[1] = a[1] / n;
  a[2] = a[2] / n;
}
static void matvec3(const double M[9], const double v[3], double o[3]) {
  o[0] = (M[0] * v[0] + M[1] * v[1]) + M[2] * v[2];
  o[1] = (M[3] * v[0] + M[4] * v[1]) + M[5] * v[2];
  o[2] = (M[6] * v[0] + M[7] * v[1]) + M[8] * v[2];
}
static void matmul3(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
static void transpose3(const double A[9], double T[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
}

int orc_p3p_kneip(const double *x2d /*3x2*/, const double *X /*3x3, row = point*/, double *models /*4x12*/) {
  double P1[3] = {X[0], X[1], X[2]}, P2[3] = {X[3], X[4], X[5]}, P3[3] = {X[6], X[7], X[8]};
  double f1[3] = {x2d[0], x2d[1], 1.0}, f2[3] = {x2d[2], x2d[3], 1.0}, f3[3] = {x2d[4], x2d[5], 1.0};
  normalize3(f1);
  normalize3(f2);
  normalize3(f3);
  double d21[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]};
  double d31[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]};
  double cr[3];
  cross3(d21, d31, cr);
  if (norm3(cr) == 0.0) return 0;

  double e1[3], e2[3], e3[3], T[9], f3t[3];
  for (int pass = 0; pass < 2; ++pass) {
    e1[0] = f1[0];
    e1[1] = f1[1];
    e1[2] = f1[2];
    cross3(f1, f2, e3);
    normalize3(e3);
    cross3(e3, e1, e2);
    for (int k = 0; k < 3; ++k) {
      T[k] = e1[k];
      T[3 + k] = e2[k];
      T[6 + k] = e3[k];
    }
    matvec3(T, f3, f3t);
    if (pass == 0 && f3t[2] > 0.0) {
      /* enforce f3[2] <= 0 so that theta lies in [0, pi]: swap the roles of points 1 and 2 */
      for (int k = 0; k < 3; ++k) {
        double t = f1[k];
        f1[k] = f2[k];
        f2[k] = t;
        t = P1[k];
        P1[k] = P2[k];
        P2[k] = t;
      }
      continue;
    }
    break;
  }
  double n1[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]};
  double p31[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]};
  const double d_12 = norm3(n1);
  normalize3(n1);
  double n3[3], n2[3], N[9];
  cross3(n1, p31, n3);
  normalize3(n3);
  cross3(n3, n1, n2);
  for (int k = 0; k < 3; ++k) {
    N[k] = n1[k];
    N[3 + k] = n2[k];
    N[6 + k] = n3[k];
  }
  double P3n[3];
  matvec3(N, p31, P3n);
  const double f_1 = f3t[0] / f3t[2];
  const double f_2 = f3t[1] / f3t[2];
  const double p_1 = P3n[0];
  const double p_2 = P3n[1];
  const double cos_beta = dot3(f1, f2);
  double b = 1.0 / (1.0 - cos_beta * cos_beta) - 1.0;
  b = (cos_beta < 0.0) ? -sqrt(b) : sqrt(b);

  const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
  const double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
  const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
  const double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;

  double fac[5];
  fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
  fac[1] = 2.0 * p_2_pw3 * d_12 * b + 2.0 * f_2_pw2 * p_2_pw3 * d_12 * b - 2.0 * f_2 * p_2_pw3 * f_1 * d_12;
  fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 +
           f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2.0 * p_1 * p_2_pw2 * d_12 +
           2.0 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2 +
           2.0 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2.0 * p_1_pw2 * p_2_pw2;
  fac[3] = 2.0 * p_1_pw2 * p_2 * d_12 * b + 2.0 * f_2 * p_2_pw3 * f_1 * d_12 - 2.0 * f_2_pw2 * p_2_pw3 * d_12 * b -
           2.0 * p_1 * p_2 * d_12_pw2 * b;
  fac[4] = -2.0 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2.0 * p_1_pw3 * d_12 -
           p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2.0 * f_2_pw2 * p_2_pw2 * p_1 * d_12 +
           p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;

  double roots[4];
  orc_solve_quartic_real(fac, roots);

  double NT[9];
  transpose3(N, NT);
  for (int i = 0; i < 4; ++i) {
    const double cos_theta = roots[i];
    const double cot_alpha =
        (-f_1 * p_1 / f_2 - cos_theta * p_2 + d_12 * b) / (-f_1 * cos_theta * p_2 / f_2 + p_1 - d_12);
    const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    const double sin_alpha = sqrt(1.0 / (cot_alpha * cot_alpha + 1.0));
    double cos_alpha = sqrt(1.0 - sin_alpha * sin_alpha);
    if (cot_alpha < 0.0) cos_alpha = -cos_alpha;
    const double kk = d_12 * (sin_alpha * b + cos_alpha);
    double Ce[3] = {cos_alpha * kk, cos_theta * sin_alpha * kk, sin_theta * sin_alpha * kk};
    double C[3];
    matvec3(NT, Ce, C);
    C[0] = P1[0] + C[0];
    C[1] = P1[1] + C[1];
    C[2] = P1[2] + C[2];
    const double Q[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta,
                         sin_alpha,  -cos_alpha * cos_theta, -cos_alpha * sin_theta,
                         0.0,        -sin_theta,             cos_theta};
    /* world -> camera rotation R = T^T Q N ; t = -R C */
    double QN[9], TT[9], R[9], t[3];
    matmul3(Q, N, QN);
    transpose3(T, TT);
    matmul3(TT, QN, R);
    matvec3(R, C, t);
    double *M = models + 12 * i;
    for (int r = 0; r < 3; ++r) {
      M[4 * r + 0] = R[3 * r + 0];
      M[4 * r + 1] = R[3 * r + 1];
      M[4 * r + 2] = R[3 * r + 2];
      M[4 * r + 3] = -t[r];
    }
  }
  return 4;
}

/* ------------------------------------------------------------------------------------------------ */
/* AC-RANSAC (OpenMVG robust_estimator_ACRansac.hpp)                                                   */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  int min_samples;  /* Kernel::MINIMUM_SAMPLES */
  int max_models;   /* Kernel::MAX_MODELS */
  int model_size;   /* doubles per model */
  int n;            /* NumSamples */
  double logalpha0, mult_error;
  uint32_t stage;
  /* data */
  const double *a; /* F: normalised x1 [n*2]      | P3P: normalised x2d [n*2] */
  const double *b; /* F: normalised x2 [n*2]      | P3P: X [n*3]             */
} ac_kernel;

static int kernel_fit(const ac_kernel *k, const int32_t *s, double *models) {
  if (k->stage == STAGE_FMATRIX) {
    double x1[14], x2[14];
    for (int i = 0; i < 7; ++i) {
      x1[2 * i] = k->a[2 * s[i]];
      x1[2 * i + 1] = k->a[2 * s[i] + 1];
      x2[2 * i] = k->b[2 * s[i]];
      x2[2 * i + 1] = k->b[2 * s[i] + 1];
    }
    return orc_seven_point(x1, x2, models);
  }
  double x[6], X[9];
  for (int i = 0; i < 3; ++i) {
    x[2 * i] = k->a[2 * s[i]];
    x[2 * i + 1] = k->a[2 * s[i] + 1];
    X[3 * i] = k->b[3 * s[i]];
    X[3 * i + 1] = k->b[3 * s[i] + 1];
    X[3 * i + 2] = k->b[3 * s[i] + 2];
  }
  return orc_p3p_kneip(x, X, models);
}

static double kernel_error(const ac_kernel *k, const double *M, int i) {
  double e;
  if (k->stage == STAGE_FMATRIX) {
    /* EpipolarDistanceError: squared distance of x2 to the epipolar line F x1 */
    const double x = k->a[2 * i], y = k->a[2 * i + 1], u = k->b[2 * i], v = k->b[2 * i + 1];
    const double l0 = (M[0] * x + M[1] * y) + M[2];
    const double l1 = (M[3] * x + M[4] * y) + M[5];
    const double l2 = (M[6] * x + M[7] * y) + M[8];
    const double num = (l0 * u + l1 * v) + l2;
    e = (num * num) / (l0 * l0 + l1 * l1);
  } else {
    /* ResectionSquaredResidualError: ||Project(P, X) - x||^2 */
    const double X = k->b[3 * i], Y = k->b[3 * i + 1], Z = k->b[3 * i + 2];
    const double p0 = ((M[0] * X + M[1] * Y) + M[2] * Z) + M[3];
    const double p1 = ((M[4] * X + M[5] * Y) + M[6] * Z) + M[7];
    const double p2 = ((M[8] * X + M[9] * Y) + M[10] * Z) + M[11];
    const double dx = p0 / p2 - k->a[2 * i];
    const double dy = p1 / p2 - k->a[2 * i + 1];
    e = dx * dx + dy * dy;
  }
  if (is_nan(e)) e = pos_inf();
  return e;
}

typedef struct {
  double err;
  int32_t idx;
} err_idx;

static int cmp_err_idx(const void *pa, const void *pb) {
  const err_idx *a = (const err_idx *)pa, *b = (const err_idx *)pb;
  if (a->err < b->err) return -1;
  if (a->err > b->err) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

/* logcombi(k, n) = log10 C(n, k) as OpenMVG computes it (sum of log10 differences, then float) */
static float logcombi(int k, int n, const double *L10) {
  if (k >= n || k <= 0) return 0.0f;
  if (n - k < k) k = n - k;
  double r = 0.0;
  for (int i = 1; i <= k; ++i) r += L10[n - i + 1] - L10[i];
  return (float)r;
}

void orc_logcombi_tables(int s, int n, float *logc_n, float *logc_k) {
  double *L10 = (double *)malloc((size_t)(n + 2) * sizeof(double));
  L10[0] = 0.0;
  for (int i = 1; i <= n + 1; ++i) L10[i] = orc_det_log10((double)i);
  for (int k = 0; k <= n; ++k) logc_n[k] = logcombi(k, n, L10);
  for (int m = 0; m <= n; ++m) logc_k[m] = logcombi(s, m, L10);
  free(L10);
}

/* Optional trace of one acransac() run (tools/k5_policy_sim.py: which iterations improved the model decides what a
 * round-by-round schedule of the device costs; the schedule never changes the result).  Entry = iteration << 2 |
 * (index set changed) << 1 | (model improved); not thread-safe: set, run one estimation, read. */
static int32_t *g_ac_trace = NULL;
static int g_ac_trace_cap = 0, g_ac_trace_n = 0;
void orc_acransac_trace(int32_t *buf, int cap) {
  g_ac_trace = buf;
  g_ac_trace_cap = cap;
  g_ac_trace_n = 0;
}
int orc_acransac_trace_count(void) { return g_ac_trace_n; }

/* returns {errorMax (normalised frame), minNFA}; inliers in vec_inliers order (ascending residual);
 * model = best model in the normalised frame */
static void acransac(const ac_kernel *K, int n_iter_in, double max_threshold, uint64_t seed, uint32_t stream,
                     int32_t *vec_inliers, int *n_inliers, double *model, double *out_errmax, double *out_nfa,
                     int *out_iters) {
  const int s = K->min_samples, n = K->n;
  *n_inliers = 0;
  *out_errmax = pos_inf();
  *out_nfa = pos_inf();
  if (out_iters) *out_iters = 0;
  if (n <= s) return;
  err_idx *res = (err_idx *)malloc((size_t)n * sizeof(err_idx));
  int32_t *vec_index = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  float *logc_n = (float *)malloc((size_t)(n + 1) * sizeof(float));
  float *logc_k = (float *)malloc((size_t)(n + 1) * sizeof(float));
  double *models = (double *)malloc((size_t)K->max_models * K->model_size * sizeof(double));
  int n_index = n;
  for (int i = 0; i < n; ++i) vec_index[i] = i;
  const double loge0 = orc_det_log10((double)K->max_models * (double)(n - s));
  orc_logcombi_tables(s, n, logc_n, logc_k);

  double min_nfa = pos_inf(), error_max = pos_inf();
  int n_in = 0;
  long n_iter = n_iter_in;
  long n_reserve = n_iter / 10;
  n_iter -= n_reserve;
  long iter;
  for (iter = 0; iter < n_iter; ++iter) {
    int32_t sample[8];
    ac_sample(s, vec_index, n_index, seed, K->stage, stream, (uint32_t)iter, sample);
    const int nm = kernel_fit(K, sample, models);
    int better = 0;
    for (int k = 0; k < nm; ++k) {
      const double *M = models + (size_t)k * K->model_size;
      for (int i = 0; i < n; ++i) {
        res[i].err = kernel_error(K, M, i);
        res[i].idx = i;
      }
      qsort(res, (size_t)n, sizeof(err_idx), cmp_err_idx);
      /* bestNFA */
      double best_nfa = pos_inf();
      int best_k = s;
      for (int kk = s + 1; kk <= n && res[kk - 1].err <= max_threshold; ++kk) {
        const double logalpha = K->logalpha0 + K->mult_error * orc_det_log10(res[kk - 1].err + (double)FLT_EPSILON);
        const double nfa = loge0 + logalpha * (double)(kk - s) + (double)logc_n[kk] + (double)logc_k[kk];
        if (nfa < best_nfa) {
          best_nfa = nfa;
          best_k = kk;
        }
      }
      if (best_nfa < min_nfa) {
        better = 1;
        min_nfa = best_nfa;
        n_in = best_k;
        for (int i = 0; i < best_k; ++i) vec_inliers[i] = res[i].idx;
        error_max = res[best_k - 1].err;
        memcpy(model, M, (size_t)K->model_size * sizeof(double));
      }
    }
    int changed = 0;
    if ((better && min_nfa < 0.0) || (iter + 1 == n_iter && n_reserve)) {
      if (n_in == 0) {
        n_iter++;
        n_reserve--;
      } else {
        changed = 1;
        memcpy(vec_index, vec_inliers, (size_t)n_in * sizeof(int32_t));
        n_index = n_in;
        if (n_reserve) {
          n_iter = iter + 1 + n_reserve;
          n_reserve = 0;
        }
      }
    }
    if ((better || changed) && g_ac_trace && g_ac_trace_n < g_ac_trace_cap)
      g_ac_trace[g_ac_trace_n++] = (int32_t)((iter << 2) | (changed << 1) | better);
  }
  if (out_iters) *out_iters = (int)iter;
  if (min_nfa >= 0.0) n_in = 0;
  *n_inliers = n_in;
  *out_errmax = error_max;
  *out_nfa = min_nfa;
  free(res);
  free(vec_index);
  free(logc_n);
  free(logc_k);
  free(models);
}

/*
 * GeometricFilter_FMatrix_AC::Robust_estimation for one (view I, query J) pair, as called from
 * hulo::geometricMatch (MatchUtils.cpp:412-416) with guided matching off:
 *   x1/x2 raw pixel coordinates of the m putative matches (no undistortion: MatchUtils.cpp:381-410 builds an
 *   SfM_Data without intrinsics), image sizes for the normalisation and logalpha0 (point-to-line form),
 *   precision = geomPrec (upper bound geomPrec^2 px^2), n_iter = ransacRound.
 * Returns the number of inliers kept (0 unless > 2.5*7); inliers = putative indices in vec_inliers order.
 */
int orc_fmatrix_filter(const double *x1, int w1, int h1, const double *x2, int w2, int h2, int m, double precision,
                       int n_iter, uint64_t seed, uint32_t stream, int32_t *inliers, double *out_F,
                       double *out_nfa, double *out_errmax, int *out_iters) {
  if (m <= 0) return 0;
  double *a = (double *)malloc((size_t)m * 2 * sizeof(double));
  double *b = (double *)malloc((size_t)m * 2 * sizeof(double));
  /* NormalizePoints(x, w, h): N = [s 0 -w s/2; 0 s -h s/2; 0 0 1], s = 1/sqrt(w*h) */
  const double s1 = 1.0 / sqrt((double)(w1 * h1)), s2 = 1.0 / sqrt((double)(w2 * h2));
  const double t1x = -0.5 * (double)w1 * s1, t1y = -0.5 * (double)h1 * s1;
  const double t2x = -0.5 * (double)w2 * s2, t2y = -0.5 * (double)h2 * s2;
  for (int i = 0; i < m; ++i) {
    a[2 * i] = s1 * x1[2 * i] + t1x;
    a[2 * i + 1] = s1 * x1[2 * i + 1] + t1y;
    b[2 * i] = s2 * x2[2 * i] + t2x;
    b[2 * i + 1] = s2 * x2[2 * i + 1] + t2y;
  }
  ac_kernel K;
  K.min_samples = 7;
  K.max_models = 3;
  K.model_size = 9;
  K.n = m;
  K.stage = STAGE_FMATRIX;
  K.a = a;
  K.b = b;
  const double D = sqrt((double)w2 * (double)w2 + (double)h2 * (double)h2);
  const double A = (double)w2 * (double)h2;
  K.logalpha0 = orc_det_log10(2.0 * D / A / s2);
  K.mult_error = 0.5;
  const double max_threshold = (precision * precision) * s2 * s2;
  int n_in = 0;
  double F[9] = {0}, errmax, nfa;
  acransac(&K, n_iter, max_threshold, seed, stream, inliers, &n_in, F, &errmax, &nfa, out_iters);
  if (out_F) memcpy(out_F, F, sizeof(F));
  if (out_nfa) *out_nfa = nfa;
  if (out_errmax) *out_errmax = errmax;
  free(a);
  free(b);
  if ((double)n_in > 7 * 2.5) return n_in;
  return 0;
}

/*
 * GeometricFilter_FMatrix_AC::Geometry_guided_matching (OpenMVG 1.1 matching_image_collection/F_ACRobust.hpp) as
 * ImageCollectionGeometricFilter::Robust_model_estimation calls it when b_guided_matching is set
 * (MatchUtils.cpp:412-416 forwards hulo's -gm; d_distance_ratio keeps its default 0.6):
 *   m_F                 = the AC-RANSAC model un-normalised, N2^T F N1 (ACKernelAdaptor::Unnormalize)
 *   m_dPrecision_robust = sqrt(errorMax) / N2(0,0)            (ACKernelAdaptor::unormalizeError)
 *   geometry_aware::GuidedMatching<Mat3, EpipolarDistanceError>(m_F, camI = NULL, regions I, camJ = NULL, regions J,
 *                                                               Square(m_dPrecision_robust), Square(0.6), matches)
 * (the SfM_Data hulo::geometricMatch hands over holds views only, MatchUtils.cpp:381-410, so no intrinsic is found
 * and the positions are the raw .feat ones): for every feature i of image I, over ALL features j of image J in index
 * order, those with EpipolarDistanceError(F, x_i, x_j) < threshold compete by Binary_Regions::
 * SquaredDescriptorDistance (the Hamming distance, squared); distanceRatio<double> keeps best / second best
 * (strict <, so the lowest j wins ties) and the match (i, best j) is kept iff a second candidate exists and
 * best < Square(ratio) * second.  One entry per i, ascending i (IndMatch::getDeduplicated sorts by (i, j)).
 * OpenMVG's source is not in this image: restated from its published algorithm -- PARITY UNPINNED.
 *   F_norm, errmax_norm: orc_fmatrix_filter's out_F / out_errmax.  xy: float pairs as .feat stores them.
 * Returns the number of matches written to out_i / out_j.
 */
static int hamming64(const uint8_t *a, const uint8_t *b) {
  int d = 0;
  for (int k = 0; k < 64; ++k) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
  return d;
}

void orc_unnormalize_f(const double *F, int w1, int h1, int w2, int h2, double *Fp) {
  const double s1 = 1.0 / sqrt((double)(w1 * h1)), s2 = 1.0 / sqrt((double)(w2 * h2));
  const double N1[9] = {s1, 0.0, -0.5 * (double)w1 * s1, 0.0, s1, -0.5 * (double)h1 * s1, 0.0, 0.0, 1.0};
  const double N2[9] = {s2, 0.0, -0.5 * (double)w2 * s2, 0.0, s2, -0.5 * (double)h2 * s2, 0.0, 0.0, 1.0};
  double T[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T[3 * r + c] = (N2[r] * F[c] + N2[3 + r] * F[3 + c]) + N2[6 + r] * F[6 + c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Fp[3 * r + c] = (T[3 * r] * N1[c] + T[3 * r + 1] * N1[3 + c]) + T[3 * r + 2] * N1[6 + c];
}

int orc_guided_match(const double *F_norm, double errmax_norm, int w1, int h1, int w2, int h2, const float *xy1,
                     const uint8_t *desc1, int n1, const float *xy2, const uint8_t *desc2, int n2, double dist_ratio,
                     int32_t *out_i, int32_t *out_j) {
  double F[9];
  orc_unnormalize_f(F_norm, w1, h1, w2, h2, F);
  const double s2 = 1.0 / sqrt((double)(w2 * h2));
  const double pr = sqrt(errmax_norm) / s2;
  const double err_th = pr * pr;
  const double ratio2 = dist_ratio * dist_ratio;
  int n = 0;
  for (int i = 0; i < n1; ++i) {
    const double x = (double)xy1[2 * i], y = (double)xy1[2 * i + 1];
    const double l0 = (F[0] * x + F[1] * y) + F[2];
    const double l1 = (F[3] * x + F[4] * y) + F[5];
    const double l2 = (F[6] * x + F[7] * y) + F[8];
    const double den = l0 * l0 + l1 * l1;
    double bd = DBL_MAX, sbd = DBL_MAX;
    int idx = 0;
    for (int j = 0; j < n2; ++j) {
      const double u = (double)xy2[2 * j], v = (double)xy2[2 * j + 1];
      const double num = (l0 * u + l1 * v) + l2;
      const double e = (num * num) / den;
      if (e < err_th) {
        const int d = hamming64(desc1 + 64 * (size_t)i, desc2 + 64 * (size_t)j);
        const double dd = (double)(d * d);
        if (dd < bd) {
          idx = j;
          sbd = bd;
          bd = dd;
        } else if (dd < sbd) {
          sbd = dd;
        }
      }
    }
    if (sbd != DBL_MAX && bd < ratio2 * sbd) {
      out_i[n] = i;
      out_j[n] = idx;
      ++n;
    }
  }
  return n;
}

/*
 * SfM_Localizer::Localize with a valid pinhole intrinsic (P3P branch) on resection_data.pt2D / pt3D
 * (localization.cpp:479-509): AC-RANSAC on K^-1-normalised points, logalpha0 = log10(pi), no upper bound on
 * the precision, max_iteration iterations.  Returns the number of inliers if it exceeds 2.5*3, else 0.
 * P = K [R|t] (row-major 3x4) of the best model; errmax is un-normalised (pixels) as OpenMVG reports it.
 */
int orc_p3p_localize(const double *pt2d, const double *pt3d, int n, double focal, double ppx, double ppy,
                     int max_iteration, uint64_t seed, uint32_t stream, int32_t *inliers, double *P,
                     double *out_errmax, double *out_nfa, int *out_iters) {
  if (n <= 0) return 0;
  double *a = (double *)malloc((size_t)n * 2 * sizeof(double));
  const double inv_f = 1.0 / focal;
  const double cx = -ppx * inv_f, cy = -ppy * inv_f;
  for (int i = 0; i < n; ++i) {
    a[2 * i] = pt2d[2 * i] * inv_f + cx;
    a[2 * i + 1] = pt2d[2 * i + 1] * inv_f + cy;
  }
  ac_kernel K;
  K.min_samples = 3;
  K.max_models = 4;
  K.model_size = 12;
  K.n = n;
  K.stage = STAGE_P3P;
  K.a = a;
  K.b = pt3d;
  K.logalpha0 = orc_det_log10(3.14159265358979323846);
  K.mult_error = 1.0;
  int n_in = 0;
  double M[12] = {0}, errmax, nfa;
  acransac(&K, max_iteration, pos_inf(), seed, stream, inliers, &n_in, M, &errmax, &nfa, out_iters);
  free(a);
  if (n_in > 0) {
    /* Unnormalize: P = K * [R|t] ; errorMax = sqrt(e) / N(0,0) = sqrt(e) * f */
    for (int j = 0; j < 4; ++j) {
      P[j] = focal * M[j] + ppx * M[8 + j];
      P[4 + j] = focal * M[4 + j] + ppy * M[8 + j];
      P[8 + j] = M[8 + j];
    }
    errmax = sqrt(errmax) / inv_f;
  } else {
    for (int j = 0; j < 12; ++j) P[j] = 0.0;
  }
  if (out_errmax) *out_errmax = errmax;
  if (out_nfa) *out_nfa = nfa;
  if ((double)n_in > 2.5 * 3) return n_in;
  return 0;
}

/* KRt_From_P (OpenMVG multiview/projection.cpp, after libmv): RQ by Givens rotations, positive diagonal,
 * det R = +1, K(2,2) = 1.  All matrices row-major. */
void orc_krt_from_p(const double *P, double *Kout, double *Rout, double *tout) {
  double K[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
  double Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double T1[9], T2[9], G[9], GT[9];
  if (K[7] != 0.0) { /* K(2,1) */
    double c = -K[8], s = K[7];
    const double l = sqrt(c * c + s * s);
    c = c / l;
    s = s / l;
    const double Qx[9] = {1, 0, 0, 0, c, -s, 0, s, c};
    memcpy(G, Qx, sizeof(G));
    matmul3(K, G, T1);
    memcpy(K, T1, sizeof(K));
    transpose3(G, GT);
    matmul3(GT, Q, T2);
    memcpy(Q, T2, sizeof(Q));
  }
  if (K[6] != 0.0) { /* K(2,0) */
    double c = K[8], s = K[6];
    const double l = sqrt(c * c + s * s);
    c = c / l;
    s = s / l;
    const double Qy[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
    memcpy(G, Qy, sizeof(G));
    matmul3(K, G, T1);
    memcpy(K, T1, sizeof(K));
    transpose3(G, GT);
    matmul3(GT, Q, T2);
    memcpy(Q, T2, sizeof(Q));
  }
  if (K[3] != 0.0) { /* K(1,0) */
    double c = -K[4], s = K[3];
    const double l = sqrt(c * c + s * s);
    c = c / l;
    s = s / l;
    const double Qz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    memcpy(G, Qz, sizeof(G));
    matmul3(K, G, T1);
    memcpy(K, T1, sizeof(K));
    transpose3(G, GT);
    matmul3(GT, Q, T2);
    memcpy(Q, T2, sizeof(Q));
  }
  double R[9];
  memcpy(R, Q, sizeof(R));
  if (K[8] < 0.0)
    for (int i = 0; i < 9; ++i) {
      K[i] = -K[i];
      R[i] = -R[i];
    }
  if (K[4] < 0.0) { /* K = K S, R = S R, S = diag(1,-1,1) */
    for (int i = 0; i < 3; ++i) K[3 * i + 1] = -K[3 * i + 1];
    for (int j = 0; j < 3; ++j) R[3 + j] = -R[3 + j];
  }
  if (K[0] < 0.0) { /* S = diag(-1,1,1) */
    for (int i = 0; i < 3; ++i) K[3 * i] = -K[3 * i];
    for (int j = 0; j < 3; ++j) R[j] = -R[j];
  }
  /* t = K^-1 p4 by back substitution (K is upper triangular here) */
  double t[3];
  t[2] = P[11] / K[8];
  t[1] = (P[7] - K[5] * t[2]) / K[4];
  t[0] = ((P[3] - K[1] * t[1]) - K[2] * t[2]) / K[0];
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                     R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0.0) {
    for (int i = 0; i < 9; ++i) R[i] = -R[i];
    for (int i = 0; i < 3; ++i) t[i] = -t[i];
  }
  const double k22 = K[8];
  for (int i = 0; i < 9; ++i) Kout[i] = K[i] / k22;
  memcpy(Rout, R, sizeof(R));
  memcpy(tout, t, sizeof(t));
}

/* camera centre t_out = -R^T t (localization.cpp:547) */
void orc_center_from_rt(const double *R, const double *t, double *c) {
  for (int i = 0; i < 3; ++i) c[i] = -((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]);
}

/*
 * matchProviderToMatchSet (SfMDataUtils.cpp:59-125) over the geometric matches, which std::map iterates in
 * ascending view id, each list in its stored order (= vec_inliers order).
 *   geo_view[g], geo_i[g], geo_j[g]   flattened geometric matches in that order (view = index into the view table)
 *   put_*                              the putative lists of orc_match_to_query (featDist source)
 *   row_landmark                       (view, feat) -> landmark slot or -1 (structureToMapViewFeatTo3D)
 * featDist[(v,q)][j] = d0 of the LAST putative match of view v whose nearest query feature is j
 * (MatchUtils.cpp:351: later i overwrites).  For each query feature the candidate with the strictly
 * smallest distance wins, the first one on ties (SfMDataUtils.cpp:109).  Output sorted by query feature.
 */
int orc_match_set(const uint32_t *geo_view, const uint32_t *geo_i, const uint32_t *geo_j, int n_geo,
                  const uint32_t *view_off, const uint32_t *put_count, const uint32_t *put_i, const uint32_t *put_j,
                  const uint32_t *put_d, const int32_t *row_landmark, uint32_t nq, uint32_t *out_qfeat,
                  int32_t *out_landmark) {
  (void)put_i;
  int32_t *best_lm = (int32_t *)malloc((size_t)nq * sizeof(int32_t));
  float *best_d = (float *)malloc((size_t)nq * sizeof(float));
  for (uint32_t j = 0; j < nq; ++j) best_lm[j] = -1;
  for (int g = 0; g < n_geo; ++g) {
    const uint32_t v = geo_view[g], i = geo_i[g], j = geo_j[g];
    const int32_t lm = row_landmark[view_off[v] + i];
    if (lm < 0) continue; /* (view, feat) has no landmark */
    /* featDist[(v,q)].find(j) */
    int found = 0;
    uint32_t dist = 0;
    for (uint32_t k = 0; k < put_count[v]; ++k)
      if (put_j[view_off[v] + k] == j) {
        found = 1;
        dist = put_d[view_off[v] + k]; /* keep overwriting: last one wins */
      }
    if (!found) continue;
    if (best_lm[j] < 0 || best_d[j] > (float)(int)dist) {
      best_lm[j] = lm;
      best_d[j] = (float)(int)dist;
    }
  }
  int n = 0;
  for (uint32_t j = 0; j < nq; ++j)
    if (best_lm[j] >= 0) {
      out_qfeat[n] = j;
      out_landmark[n] = best_lm[j];
      ++n;
    }
  free(best_lm);
  free(best_d);
  return n;
}

/*
 * A13 (north-star extension; NOT in the reference, whose localiser never calls RefinePose -- SURVEY.md F4):
 * Levenberg-Marquardt refinement of [R|t] on the inliers, intrinsics fixed, squared reprojection error, left
 * rotation increment Exp(w) R.  Same algorithm and schedule as the HIP kernel (acransac.hip refine_pose_block);
 * parity unpinned by definition, compared to the kernel within a tolerance (the kernel accumulates the normal
 * equations with fused multiply-adds on the matrix cores).  Returns the final cost.
 */
static void normal_equations(const double *pt2d, const double *pt3d, const int32_t *inl, int n, double f, double ppx,
                             double ppy, const double *R, const double *t, double H[7][7]) {
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) H[i][j] = 0.0;
  for (int row = 0; row < 2 * n; ++row) {
    const int p = inl[row >> 1];
    const double X = pt3d[3 * p], Y = pt3d[3 * p + 1], Z = pt3d[3 * p + 2];
    const double rx = (R[0] * X + R[1] * Y) + R[2] * Z;
    const double ry = (R[3] * X + R[4] * Y) + R[5] * Z;
    const double rz = (R[6] * X + R[7] * Y) + R[8] * Z;
    const double xc = rx + t[0], yc = ry + t[1], zc = rz + t[2];
    const double iz = 1.0 / zc;
    double g0, g1, g2, res;
    if ((row & 1) == 0) {
      g0 = f * iz;
      g1 = 0.0;
      g2 = -f * xc * iz * iz;
      res = (f * xc * iz + ppx) - pt2d[2 * p];
    } else {
      g0 = 0.0;
      g1 = f * iz;
      g2 = -f * yc * iz * iz;
      res = (f * yc * iz + ppy) - pt2d[2 * p + 1];
    }
    const double m[7] = {ry * g2 - rz * g1, rz * g0 - rx * g2, rx * g1 - ry * g0, g0, g1, g2, res};
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) H[i][j] += m[i] * m[j];
  }
}

static void rotate_left(const double *w, const double *R, double *out) {
  const double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
  const double th = sqrt(th2);
  double a, b;
  if (th < 1e-8) {
    a = 1.0 - th2 / 6.0;
    b = 0.5 - th2 / 24.0;
  } else {
    a = sin(th) / th;
    b = (1.0 - cos(th)) / th2;
  }
  const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  double W2[9], E[9];
  matmul3(W, W, W2);
  for (int i = 0; i < 9; ++i) E[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * W[i] + b * W2[i];
  matmul3(E, R, out);
}

double orc_refine_pose(const double *pt2d, const double *pt3d, const int32_t *inl, int n, double f, double ppx,
                       double ppy, double *R, double *t, int max_iter, int *iters_out, double *cost0_out) {
  double S[7][7];
  normal_equations(pt2d, pt3d, inl, n, f, ppx, ppy, R, t, S);
  double H[6][6], g[6], cost = S[6][6];
  if (cost0_out) *cost0_out = cost;
  for (int i = 0; i < 6; ++i) {
    g[i] = S[i][6];
    for (int j = 0; j < 6; ++j) H[i][j] = S[i][j];
  }
  double lambda = 1e-4;
  int it = 0;
  for (; it < max_iter; ++it) {
    double A[6][7];
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 6; ++j) A[i][j] = H[i][j];
      A[i][i] = H[i][i] * (1.0 + lambda);
      A[i][6] = -g[i];
    }
    int singular = 0;
    for (int c = 0; c < 6; ++c) {
      int piv = c;
      for (int r = c + 1; r < 6; ++r)
        if (dabs(A[r][c]) > dabs(A[piv][c])) piv = r;
      if (!(dabs(A[piv][c]) > 0.0)) {
        singular = 1;
        break;
      }
      if (piv != c)
        for (int j = c; j < 7; ++j) {
          const double tt = A[c][j];
          A[c][j] = A[piv][j];
          A[piv][j] = tt;
        }
      for (int r = c + 1; r < 6; ++r) {
        const double fct = A[r][c] / A[c][c];
        for (int j = c; j < 7; ++j) A[r][j] = A[r][j] - fct * A[c][j];
      }
    }
    if (singular) break;
    double d[6];
    for (int i = 5; i >= 0; --i) {
      double acc = A[i][6];
      for (int j = i + 1; j < 6; ++j) acc = acc - A[i][j] * d[j];
      d[i] = acc / A[i][i];
    }
    double Rn[9], tn[3];
    rotate_left(d, R, Rn);
    tn[0] = t[0] + d[3];
    tn[1] = t[1] + d[4];
    tn[2] = t[2] + d[5];
    normal_equations(pt2d, pt3d, inl, n, f, ppx, ppy, Rn, tn, S);
    const double c_new = S[6][6];
    if (c_new < cost) {
      const double rel = (cost - c_new) / cost;
      memcpy(R, Rn, sizeof(Rn));
      memcpy(t, tn, sizeof(tn));
      cost = c_new;
      for (int i = 0; i < 6; ++i) {
        g[i] = S[i][6];
        for (int j = 0; j < 6; ++j) H[i][j] = S[i][j];
      }
      lambda = lambda * 0.1;
      if (lambda < 1e-12) lambda = 1e-12;
      if (rel < 1e-10) {
        ++it;
        break;
      }
    } else {
      lambda = lambda * 10.0;
      if (lambda > 1e10) break;
    }
  }
  if (iters_out) *iters_out = it;
  return cost;
}


/* Pinhole_Intrinsic_Radial_K3::get_ud_pixel (OpenMVG 1.1, restated from its published source; parity unpinned):
 * cam2ima(remove_disto(ima2cam(p))), remove_disto = p * sqrt(bisection_Radius_Solve(r2) / r2) with
 * distoFunctor(r2) = r2 (1 + r2 (k1 + r2 (k2 + r2 k3)))^2 inverted by bisection to 1e-8 (A10, localization.cpp:484-487).
 * Loop caps as on the device. */
static double orc_disto_k3(double k1, double k2, double k3, double r2) {
  const double t = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
  return r2 * (t * t);
}

void orc_ud_pixel_k3(double f, double ppx, double ppy, double k1, double k2, double k3, const double *xy, int n,
                     double *out) {
  for (int i = 0; i < n; ++i) {
    const double px = (xy[2 * i] - ppx) / f, py = (xy[2 * i + 1] - ppy) / f;
    const double r2 = px * px + py * py;
    double radius = 1.0;
    if (r2 != 0.0) {
      double lo = r2, up = r2;
      for (int it = 0; it < 4096 && orc_disto_k3(k1, k2, k3, lo) > r2; ++it) lo = lo / 1.05;
      for (int it = 0; it < 4096 && orc_disto_k3(k1, k2, k3, up) < r2; ++it) up = up * 1.05;
      for (int it = 0; it < 4096 && 1e-8 < up - lo; ++it) {
        const double mid = 0.5 * (lo + up);
        if (orc_disto_k3(k1, k2, k3, mid) > r2)
          up = mid;
        else
          lo = mid;
      }
      radius = sqrt((0.5 * (lo + up)) / r2);
    }
    out[2 * i] = f * (radius * px) + ppx;
    out[2 * i + 1] = f * (radius * py) + ppy;
  }
}
