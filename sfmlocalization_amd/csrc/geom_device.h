// Device-side f64 geometry used by the AC-RANSAC kernels (acransac.hip).
//
// Every function here evaluates the SAME sequence of IEEE-754 double operations as its CPU restatement in
// the test oracle, so that inlier sets can be compared bit for bit: only + - * / sqrt and comparisons, no
// library transcendentals, and the translation unit is compiled with -ffp-contract=off (no FMA fusion).
// Algorithms: 7-point fundamental matrix (H&Z 11.1.2; OpenMVG SevenPointSolver), P3P (Kneip et al. CVPR
// 2011; OpenMVG P3PSolver), RQ by Givens (OpenMVG KRt_From_P), Philox4x32-10 sampling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfmloc {
namespace geom {

#define GD __device__ __forceinline__
#define GDN __device__ __noinline__

GD uint64_t d2u(double x) { return (uint64_t)__double_as_longlong(x); }
GD double u2d(uint64_t u) { return __longlong_as_double((long long)u); }
GD bool is_nan(double x) { return (d2u(x) & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull; }
GD bool is_inf(double x) { return (d2u(x) & 0x7FFFFFFFFFFFFFFFull) == 0x7FF0000000000000ull; }
GD double pos_inf() { return u2d(0x7FF0000000000000ull); }
GD double q_nan() { return u2d(0x7FF8000000000000ull); }
GD double dabs(double x) { return x < 0.0 ? -x : x; }
GD double dmax(double a, double b) { return a > b ? a : b; }

// The .feat text round trip of a keypoint coordinate on the device: what `snprintf("%.6g", (double)v)` followed by
// `strtof` returns (sfmloc_feat_round_trip, AKAZEOpenCV.cpp:80-81 / :106-111: the query's features are written to a
// .feat file at the default stream precision and read back).  v is exact in double; it is scaled by an exact power of
// ten to 6 significant digits before the decimal point, rounded half-to-even (printf rounds the exact binary value), and
// the decimal r * 10^(e-5) is taken back to float through one correctly rounded division -- no float midpoint lies within
// 2^-53 of such a quotient for |v| >= 1, so the double rounding is harmless there.  Values of 10^6 and more are rounded
// in integer arithmetic.  Exact for 1e-10 <= |v| < 1e15 (a keypoint coordinate is a pixel position); outside that range
// the value is returned as it is.
GD float round6_dev(float vf) {
  const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
  if (!(vf == vf) || vf == 0.0f) return vf;
  const bool neg = vf < 0.0f;
  const double v = neg ? -(double)vf : (double)vf;
  if (!(v >= 1e-10 && v < 1e15)) return vf;
  double res;
  if (v >= 1e6) {  // 6 digits end left of the decimal point: integer arithmetic
    int e = 6;
    while (e < 14 && v >= p10[e + 1]) ++e;
    const long long D = (long long)p10[e - 5];
    const long long iv = (long long)v;         // (truncation; v < 2^53)
    const double frac = v - (double)iv;        // exact
    long long q = iv / D;
    const long long rem = iv % D;
    const double twice = 2.0 * ((double)rem + frac);  // exact: rem < 1e9, frac a few bits
    if (twice > (double)D || (twice == (double)D && (q & 1))) ++q;
    res = (double)(q * D);
  } else if (v >= 1.0) {
    int e = 0;
    while (e < 5 && v >= p10[e + 1]) ++e;
    const double r = rint(v * p10[5 - e]);     // exact product, half-to-even
    res = r / p10[5 - e];
  } else {
    int k = 1;
    while (k < 10 && v * p10[k] < 1.0) ++k;    // 10^-k <= v < 10^-(k-1): e = -k
    const double r = rint(v * p10[5 + k]);
    res = r / p10[5 + k];
  }
  return (float)(neg ? -res : res);
}

// log10 by a fixed operation order: x = m 2^e, m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m-1)/(m+1))
GDN double det_log10(double x) {
  if (is_nan(x) || x < 0.0) return q_nan();
  if (x == 0.0) return -pos_inf();
  if (is_inf(x)) return pos_inf();
  uint64_t u = d2u(x);
  int e = (int)((u >> 52) & 0x7FF);
  if (e == 0) {
    x = x * 18014398509481984.0;
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

// The same function without branches (same operations in the same order on the main path, so the same bits; the
// special cases are selected at the end): several calls in a row can then be interleaved by the compiler, which a
// chain of calls to the out-of-line version cannot (one call is ~40 dependent f64 operations).
__device__ __forceinline__ double det_log10_inline(double x) {
  const bool bad = is_nan(x) || x < 0.0;
  const bool zero = x == 0.0;
  const bool inf = is_inf(x);
  uint64_t u = d2u(x);
  const bool sub = ((u >> 52) & 0x7FF) == 0;
  const double xs = sub ? x * 18014398509481984.0 : x;
  u = d2u(xs);
  int e = (int)((u >> 52) & 0x7FF) - (sub ? 54 : 0) - 1023;
  double m = u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  const bool big = m > 1.4142135623730951;
  m = big ? m * 0.5 : m;
  e += big ? 1 : 0;
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
  const double r = ln * 0.4342944819032518;
  return bad ? q_nan() : zero ? -pos_inf() : inf ? pos_inf() : r;
}

GD void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
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

GD uint32_t ac_draw(uint64_t seed, uint32_t stage, uint32_t stream, uint32_t iter, uint32_t i) {
  uint32_t c[4] = {iter, stream, i >> 2, stage};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return c[i & 3];
}

// OpenMVG UniformSample: X distinct sorted positions in [0,n) mapped through vec_index (identity if null)
template <int X>
GD void ac_sample(const int32_t *vec_index, int n, uint64_t seed, uint32_t stage, uint32_t stream, uint32_t iter,
                  int32_t *samples) {
  int32_t s[X];
#pragma unroll
  for (int i = 0; i < X; ++i) s[i] = 0;
  for (int i = 0; i < X; ++i) {
    int32_t r = (int32_t)(ac_draw(seed, stage, stream, iter, (uint32_t)i) % (uint32_t)(n - i));
    int j;
    for (j = 0; j < i && r >= s[j]; ++j) ++r;
    for (int k = i; k > j; --k) s[k] = s[k - 1];
    s[j] = r;
  }
  for (int i = 0; i < X; ++i) samples[i] = vec_index ? vec_index[s[i]] : s[i];
}

GDN double cubic_one_root(double b, double c, double d) {
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

GD double cubic_polish(double a3, double a2, double a1, double a0, double x) {
  for (int it = 0; it < 2; ++it) {
    const double f = ((a3 * x + a2) * x + a1) * x + a0;
    const double df = (3.0 * a3 * x + 2.0 * a2) * x + a1;
    if (df == 0.0) break;
    x = x - f / df;
  }
  return x;
}

GDN int solve_cubic(double a3, double a2, double a1, double a0, double r[3]) {
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
  for (int i = 1; i < n; ++i) {
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

GD double quartic_polish(const double a[5], double x) {
  #pragma unroll
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

// the four candidates before their Newton polish (out[i], and whether root i is real and gets polished)
GD void solve_quartic_raw(const double a[5], double out[4], int real[4]) {
  const double b = a[1] / a[0], c = a[2] / a[0], d = a[3] / a[0], e = a[4] / a[0];
  const double b2 = b * b;
  const double p = c - 0.375 * b2;
  const double q = d - 0.5 * b * c + 0.125 * b2 * b;
  const double r = e - 0.25 * b * d + 0.0625 * b2 * c - 0.01171875 * b2 * b2;
  const double shift = -0.25 * b;
  double y[4];
  #pragma unroll
  for (int i = 0; i < 4; ++i) real[i] = 0;
  if (is_nan(p) || is_nan(q) || is_nan(r) || is_inf(p) || is_inf(q) || is_inf(r)) {
    #pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = q_nan();
    return;
  }
  if (q == 0.0) {
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
      const double mod = sqrt(r);
      const double re = sqrt(0.5 * (mod - 0.5 * p));
      y[0] = re;
      y[1] = -re;
      y[2] = re;
      y[3] = -re;
    }
  } else {
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
    const double beta[2] = {-s, s};
    const double gamma[2] = {h + g, h - g};
    #pragma unroll
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
  #pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = y[i] + shift;
}

GD void solve_quartic_real(const double a[5], double out[4]) {
  int real[4];
  solve_quartic_raw(a, out, real);
  #pragma unroll
  for (int i = 0; i < 4; ++i)
    if (real[i]) out[i] = quartic_polish(a, out[i]);
}

GD double det3c(const double *a, const double *b, const double *c) {
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - b[0] * (a[1] * c[2] - a[2] * c[1]) + c[0] * (a[1] * b[2] - a[2] * b[1]);
}

// x1, x2: 7 x 2 normalised points. F: up to 3 row-major 3x3. Returns the number of solutions.
GDN int seven_point(const double *x1, const double *x2, double *F) {
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
    if (!(best > 0.0)) return 0;
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
  double a0[3] = {f1[0], f1[3], f1[6]}, a1[3] = {f1[1], f1[4], f1[7]}, a2[3] = {f1[2], f1[5], f1[8]};
  double b0[3] = {f2[0], f2[3], f2[6]}, b1[3] = {f2[1], f2[4], f2[7]}, b2[3] = {f2[2], f2[5], f2[8]};
  const double c0 = det3c(a0, a1, a2);
  const double c1 = (det3c(b0, a1, a2) + det3c(a0, b1, a2)) + det3c(a0, a1, b2);
  const double c2 = (det3c(a0, b1, b2) + det3c(b0, a1, b2)) + det3c(b0, b1, a2);
  const double c3 = det3c(b0, b1, b2);
  double roots[3];
  const int n = solve_cubic(c3, c2, c1, c0, roots);
  for (int s = 0; s < n; ++s)
    for (int k = 0; k < 9; ++k) F[9 * s + k] = f1[k] + roots[s] * f2[k];
  return n;
}

// The same solver spread over one wave: lane l < 63 owns A[l / 9][l % 9] in a register, the pivot search is a
// butterfly reduction, row/column swaps, the pivot-row broadcast and the elimination factors are shuffles.  Every
// value goes through exactly the operations seven_point applies to it, in the same order, so the results are
// bit-identical; what changes is that nothing is dynamically indexed (no scratch) and the 7x9 eliminations run
// 63-wide.  (ax, ay, bx, by): the coordinates of point l / 9 (any value in lane 63).  All lanes return the
// number of solutions; lane l < 9 * n returns F[l] (solution l / 9, entry l % 9) in *f_out.
// wave64 maximum of a 32-bit unsigned value by DPP (quad permutes, row shifts, row broadcasts: ~8 cycles a step
// instead of a ~100-cycle ds_bpermute); every lane returns the result
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define SFM_DPP_MAX(ctrl, rmask)                                                                          \
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false))
  SFM_DPP_MAX(0xB1, 0xF);   // quad_perm:[1,0,3,2]
  SFM_DPP_MAX(0x4E, 0xF);   // quad_perm:[2,3,0,1]
  SFM_DPP_MAX(0x114, 0xF);  // row_shr:4
  SFM_DPP_MAX(0x118, 0xF);  // row_shr:8   -> lanes 12..15 of a row hold the row maximum
  SFM_DPP_MAX(0x142, 0xA);  // row_bcast:15 into rows 1 and 3
  SFM_DPP_MAX(0x143, 0xC);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
#undef SFM_DPP_MAX
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// a double held by lane `src` (wave-uniform index), in every lane: two v_readlane instead of two ds_bpermute
__device__ __forceinline__ double wave_read_f64(double v, int src) {
  const uint64_t u = d2u(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
  return u2d(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int wave_seven_point(double ax, double ay, double bx, double by, double *f_out) {
  const int lane = (int)(threadIdx.x & 63u);
  const int row = lane / 9, col = lane - 9 * row;  // lane 63: row 7 (inactive)
  const bool live = lane < 63;
  const double pa = (col < 3) ? bx : (col < 6) ? by : 1.0;
  const int c3 = col % 3;
  const double pb = (c3 == 0) ? ax : (c3 == 1) ? ay : 1.0;
  double a = (col < 6) ? ((c3 == 2) ? pa : pa * pb) : pb;  // bx*ax bx*ay bx by*ax by*ay by ax ay 1
  uint64_t perm = 0x876543210ull;  // nibble j = perm[j]
  for (int k = 0; k < 7; ++k) {
    // complete pivoting: largest |A[i][j]|, i >= k, j >= k; first in row-major order on ties.  |x| >= 0 orders like
    // its bit pattern, so the arg-max is two 32-bit wave maxima (high word, then low word among the lanes that hold
    // the high maximum) and the lowest lane of the ballot of the winners; not eligible / NaN = key 0, and a maximum
    // of 0 is the reference's "best <= 0 -> no solution".
    const double aa = dabs(a);
    // (dabs keeps the sign of -0.0, which compares equal to +0.0 as a double: clear it in the bit key)
    const uint64_t key = (live && row >= k && col >= k && aa == aa) ? (d2u(aa) & 0x7FFFFFFFFFFFFFFFull) : 0ull;
    const uint32_t khi = (uint32_t)(key >> 32), klo = (uint32_t)key;
    const uint32_t mhi = wave_max_u32(khi);
    const uint32_t mlo = wave_max_u32(khi == mhi ? klo : 0u);
    if ((mhi | mlo) == 0u) return 0;
    const unsigned long long win = __ballot(khi == mhi && klo == mlo);
    const int bi = __builtin_ctzll(win);
    const int pi = bi / 9, pj = bi - 9 * pi;
    const int sr = (row == k) ? pi : (row == pi) ? k : row;
    const int sc = (col == k) ? pj : (col == pj) ? k : col;
    a = __shfl(a, live ? sr * 9 + sc : lane, 64);
    if (pj != k) {
      const uint64_t nk = (perm >> (4 * k)) & 0xFull, nj = (perm >> (4 * pj)) & 0xFull;
      perm &= ~((0xFull << (4 * k)) | (0xFull << (4 * pj)));
      perm |= (nj << (4 * k)) | (nk << (4 * pj));
    }
    const double piv = wave_read_f64(a, k * 9 + k);
    if (live && row == k && col >= k) a = a / piv;
    const double rk = __shfl(a, live ? k * 9 + col : lane, 64);   // normalised pivot row at my column
    const double fct = __shfl(a, live ? row * 9 + k : lane, 64);  // my row's entry in the pivot column
    if (live && row != k && col >= k && fct != 0.0) a = a - fct * rk;
  }
  // null-space basis: f1[perm[i]] = -A[i][7], f2[perm[i]] = -A[i][8] (i < 7); unit entries at perm[7], perm[8]
  double f1[9], f2[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    int pos = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q)
      if ((int)((perm >> (4 * q)) & 0xFull) == c) pos = q;
    const double v1 = wave_read_f64(a, pos < 7 ? pos * 9 + 7 : 0);
    const double v2 = wave_read_f64(a, pos < 7 ? pos * 9 + 8 : 0);
    f1[c] = (pos < 7) ? -v1 : (pos == 7 ? 1.0 : 0.0);
    f2[c] = (pos < 7) ? -v2 : (pos == 8 ? 1.0 : 0.0);
  }
  double a0[3] = {f1[0], f1[3], f1[6]}, a1[3] = {f1[1], f1[4], f1[7]}, a2[3] = {f1[2], f1[5], f1[8]};
  double b0[3] = {f2[0], f2[3], f2[6]}, b1[3] = {f2[1], f2[4], f2[7]}, b2[3] = {f2[2], f2[5], f2[8]};
  const double c0 = det3c(a0, a1, a2);
  const double c1 = (det3c(b0, a1, a2) + det3c(a0, b1, a2)) + det3c(a0, a1, b2);
  const double c2 = (det3c(a0, b1, b2) + det3c(b0, a1, b2)) + det3c(b0, b1, a2);
  const double c3c = det3c(b0, b1, b2);
  double roots[3] = {0.0, 0.0, 0.0};
  const int n = solve_cubic(c3c, c2, c1, c0, roots);
  const int sol = lane / 9, ent = lane - 9 * sol;
  double g1 = 0.0, g2 = 0.0;
#pragma unroll
  for (int c = 0; c < 9; ++c)
    if (ent == c) {
      g1 = f1[c];
      g2 = f2[c];
    }
  const double rt = (sol == 0) ? roots[0] : (sol == 1) ? roots[1] : roots[2];
  *f_out = g1 + rt * g2;
  return n;
}

GD void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
GD double dot3(const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
GD double norm3(const double a[3]) { return sqrt(dot3(a, a)); }
GD void normalize3(double a[3]) {
  const double n = norm3(a);
  a[0] = a[0] / n;
  a[1] = a[1] / n;
  a[2] = a[2] / n;
}
GD void matvec3(const double M[9], const double v[3], double o[3]) {
  o[0] = (M[0] * v[0] + M[1] * v[1]) + M[2] * v[2];
  o[1] = (M[3] * v[0] + M[4] * v[1]) + M[5] * v[2];
  o[2] = (M[6] * v[0] + M[7] * v[1]) + M[8] * v[2];
}
GD void matmul3(const double A[9], const double B[9], double C[9]) {
  #pragma unroll
  for (int i = 0; i < 3; ++i)
    #pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
GD void transpose3(const double A[9], double T[9]) {
  #pragma unroll
  for (int i = 0; i < 3; ++i)
    #pragma unroll
    for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
}

// x2d: 3 x 2 normalised image points, X: 3 x 3 world points (row = point); models: 4 x 12 [R|t] row-major.
// Kneip's P3P in two parts: everything up to the quartic's four roots (one lane), then one model per root (the four
// are independent, so k_p3p_eval gives them to four lanes); p3p_kneip below is the two put together.
struct P3pPrep {
  double f_1, f_2, p_1, p_2, d_12, b;
  double N[9], T[9], P1[3];
  double fac[5], roots[4];  // the quartic and its roots BEFORE the Newton polish, which is per root
  int real[4];
};

GD int p3p_kneip_prepare(const double *x2d, const double *X, P3pPrep &S) {
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
  #pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    e1[0] = f1[0];
    e1[1] = f1[1];
    e1[2] = f1[2];
    cross3(f1, f2, e3);
    normalize3(e3);
    cross3(e3, e1, e2);
    #pragma unroll
    for (int k = 0; k < 3; ++k) {
      T[k] = e1[k];
      T[3 + k] = e2[k];
      T[6 + k] = e3[k];
    }
    matvec3(T, f3, f3t);
    if (pass == 0 && f3t[2] > 0.0) {
      #pragma unroll
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
  #pragma unroll
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

  solve_quartic_raw(fac, S.roots, S.real);
  #pragma unroll
  for (int k = 0; k < 5; ++k) S.fac[k] = fac[k];
  S.f_1 = f_1, S.f_2 = f_2, S.p_1 = p_1, S.p_2 = p_2, S.d_12 = d_12, S.b = b;
  #pragma unroll
  for (int k = 0; k < 9; ++k) {
    S.N[k] = N[k];
    S.T[k] = T[k];
  }
  S.P1[0] = P1[0], S.P1[1] = P1[1], S.P1[2] = P1[2];
  return 4;
}

// the model of root i: M = [R | -R C], 3 x 4 row major
GD void p3p_kneip_model(const P3pPrep &S, int i, double *M) {
  const double f_1 = S.f_1, f_2 = S.f_2, p_1 = S.p_1, p_2 = S.p_2, d_12 = S.d_12, b = S.b;
  double N[9], T[9], NT[9];
  #pragma unroll
  for (int k = 0; k < 9; ++k) {
    N[k] = S.N[k];
    T[k] = S.T[k];
  }
  const double P1[3] = {S.P1[0], S.P1[1], S.P1[2]};
  transpose3(N, NT);
  {
    double cos_theta = S.roots[i];
    if (S.real[i]) {
      double fac[5];
      #pragma unroll
      for (int k = 0; k < 5; ++k) fac[k] = S.fac[k];
      cos_theta = quartic_polish(fac, cos_theta);
    }
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
    double QN[9], TT[9], R[9], t[3];
    matmul3(Q, N, QN);
    transpose3(T, TT);
    matmul3(TT, QN, R);
    matvec3(R, C, t);
    #pragma unroll
    for (int r = 0; r < 3; ++r) {
      M[4 * r + 0] = R[3 * r + 0];
      M[4 * r + 1] = R[3 * r + 1];
      M[4 * r + 2] = R[3 * r + 2];
      M[4 * r + 3] = -t[r];
    }
  }
}

GD int p3p_kneip(const double *x2d, const double *X, double *models) {
  P3pPrep S;
  if (!p3p_kneip_prepare(x2d, X, S)) return 0;
  #pragma unroll
  for (int i = 0; i < 4; ++i) p3p_kneip_model(S, i, models + 12 * i);
  return 4;
}

// Pinhole_Intrinsic_Radial_K3::get_ud_pixel (OpenMVG 1.1, restated): cam2ima(remove_disto(ima2cam(p))) where
// remove_disto scales p by sqrt(bisection_Radius_Solve(r2) / r2), the bisection inverting
// distoFunctor(r2) = r2 * (1 + r2 (k1 + r2 (k2 + r2 k3)))^2 to 1e-8.  The bound-search loops are capped so that
// meaningless coefficients cannot hang a kernel (the oracle carries the same cap).
GD double disto_functor_k3(double k1, double k2, double k3, double r2) {
  const double t = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
  return r2 * (t * t);
}
GD void ud_pixel_k3(double f, double ppx, double ppy, double k1, double k2, double k3, double x, double y, double *ox,
                    double *oy) {
  const double px = (x - ppx) / f, py = (y - ppy) / f;
  const double r2 = px * px + py * py;
  double radius = 1.0;
  if (r2 != 0.0) {
    double lo = r2, up = r2;
    for (int it = 0; it < 4096 && disto_functor_k3(k1, k2, k3, lo) > r2; ++it) lo = lo / 1.05;
    for (int it = 0; it < 4096 && disto_functor_k3(k1, k2, k3, up) < r2; ++it) up = up * 1.05;
    for (int it = 0; it < 4096 && 1e-8 < up - lo; ++it) {
      const double mid = 0.5 * (lo + up);
      if (disto_functor_k3(k1, k2, k3, mid) > r2)
        up = mid;
      else
        lo = mid;
    }
    radius = sqrt((0.5 * (lo + up)) / r2);
  }
  *ox = f * (radius * px) + ppx;
  *oy = f * (radius * py) + ppy;
}

// EpipolarDistanceError: squared distance of x2 = (u,v) to the line F (x,y,1)
GD double err_fmatrix(const double *M, double x, double y, double u, double v) {
  const double l0 = (M[0] * x + M[1] * y) + M[2];
  const double l1 = (M[3] * x + M[4] * y) + M[5];
  const double l2 = (M[6] * x + M[7] * y) + M[8];
  const double num = (l0 * u + l1 * v) + l2;
  double e = (num * num) / (l0 * l0 + l1 * l1);
  if (is_nan(e)) e = pos_inf();
  return e;
}

// ResectionSquaredResidualError: ||Project(P, X) - x||^2
GD double err_resection(const double *M, double X, double Y, double Z, double x, double y) {
  const double p0 = ((M[0] * X + M[1] * Y) + M[2] * Z) + M[3];
  const double p1 = ((M[4] * X + M[5] * Y) + M[6] * Z) + M[7];
  const double p2 = ((M[8] * X + M[9] * Y) + M[10] * Z) + M[11];
  const double dx = p0 / p2 - x;
  const double dy = p1 / p2 - y;
  double e = dx * dx + dy * dy;
  if (is_nan(e)) e = pos_inf();
  return e;
}

// OpenMVG KRt_From_P (RQ by Givens), all row-major
GD void krt_from_p(const double *P, double *Kout, double *Rout, double *tout) {
  double K[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
  double Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double T1[9], T2[9], G[9], GT[9];
  if (K[7] != 0.0) {
    double c = -K[8], s = K[7];
    const double l = sqrt(c * c + s * s);
    c = c / l;
    s = s / l;
    G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 0; G[4] = c; G[5] = -s; G[6] = 0; G[7] = s; G[8] = c;
    matmul3(K, G, T1);
    #pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = T1[i];
    transpose3(G, GT);
    matmul3(GT, Q, T2);
    #pragma unroll
    for (int i = 0; i < 9; ++i) Q[i] = T2[i];
  }
  if (K[6] != 0.0) {
    double c = K[8], s = K[6];
    const double l = sqrt(c * c + s * s);
    c = c / l;
    s = s / l;
    G[0] = c; G[1] = 0; G[2] = s; G[3] = 0; G[4] = 1; G[5] = 0; G[6] = -s; G[7] = 0; G[8] = c;
    matmul3(K, G, T1);
    #pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = T1[i];
    transpose3(G, GT);
    matmul3(GT, Q, T2);
    #pragma unroll
    for (int i = 0; i < 9; ++i) Q[i] = T2[i];
  }
  if (K[3] != 0.0) {
    double c = -K[4], s = K[3];
    const double l = sqrt(c * c + s * s);
    c = c / l;
    s = s / l;
    G[0] = c; G[1] = -s; G[2] = 0; G[3] = s; G[4] = c; G[5] = 0; G[6] = 0; G[7] = 0; G[8] = 1;
    matmul3(K, G, T1);
    #pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = T1[i];
    transpose3(G, GT);
    matmul3(GT, Q, T2);
    #pragma unroll
    for (int i = 0; i < 9; ++i) Q[i] = T2[i];
  }
  double R[9];
  #pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = Q[i];
  if (K[8] < 0.0)
    #pragma unroll
    for (int i = 0; i < 9; ++i) {
      K[i] = -K[i];
      R[i] = -R[i];
    }
  if (K[4] < 0.0) {
    #pragma unroll
    for (int i = 0; i < 3; ++i) K[3 * i + 1] = -K[3 * i + 1];
    #pragma unroll
    for (int j = 0; j < 3; ++j) R[3 + j] = -R[3 + j];
  }
  if (K[0] < 0.0) {
    #pragma unroll
    for (int i = 0; i < 3; ++i) K[3 * i] = -K[3 * i];
    #pragma unroll
    for (int j = 0; j < 3; ++j) R[j] = -R[j];
  }
  double t[3];
  t[2] = P[11] / K[8];
  t[1] = (P[7] - K[5] * t[2]) / K[4];
  t[0] = ((P[3] - K[1] * t[1]) - K[2] * t[2]) / K[0];
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                     R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0.0) {
    #pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = -R[i];
    #pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = -t[i];
  }
  const double k22 = K[8];
  #pragma unroll
  for (int i = 0; i < 9; ++i) Kout[i] = K[i] / k22;
  #pragma unroll
  for (int i = 0; i < 9; ++i) Rout[i] = R[i];
  #pragma unroll
  for (int i = 0; i < 3; ++i) tout[i] = t[i];
}

GD void center_from_rt(const double *R, const double *t, double *c) {
  #pragma unroll
  for (int i = 0; i < 3; ++i) c[i] = -((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]);
}

#undef GD
#undef GDN
}  // namespace geom
}  // namespace sfmloc
