// K9: AKAZE detection + full 486-bit M-LDB description for gfx950.
//   replaces cv::AKAZE::create(DESCRIPTOR_MLDB, 0, 3, thres, nOct, nOctLay)->detectAndCompute(gray)
//   (VisionLocalizeCommon/src/AKAZEOpenCV.cpp:44-46,67) and cv::AKAZE::create()->compute(gray, keypoints)
//   (BoWCommon/src/DenseLocalFeatureWrapper.cpp:42,146).
// Algorithm: Alcantarilla, Nuevo, Bartoli, BMVC 2013 -- nonlinear (Perona-Malik g2) scale space by Fast Explicit
// Diffusion, determinant-of-Hessian extrema with Scharr derivatives, sub-pixel refinement, dominant orientation,
// rotated 2x2 / 3x3 / 4x4 grid comparisons of intensity and gradients (M-LDB).  OpenCV 3.0 is not in the image:
// parity is against the build's own CPU restatement, operation order for operation order (float32, no FMA fusion).
//
// Every pixel-wise stage is a plain memory-bound kernel over one pyramid level (<= 1.2 MB at VGA, L2-resident);
// the only sequential step -- OpenCV's order-dependent duplicate suppression over a few thousand candidates -- runs on
// the host between two kernels (candidates carry their 3x3 Hessian-response patch, so sub-pixel refinement needs no
// second trip).  Orientation and description use one 64-lane wave per keypoint with the per-cell / per-window sums
// kept sequential inside a lane so that they round exactly like the restatement.
#include <math.h>
#include <time.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <thread>
#include <vector>

#include "sfmloc_internal.h"

namespace sfmloc {
namespace {

constexpr int kMaxLevels = 32;
constexpr float kPiF = 3.14159265358979323846f;

struct AkLevel {
  int w, h, octave, sublevel, sigma_size, nsteps;
  float esigma, etime;
  float tsteps[64];
  size_t off;  // offset (floats) of this level inside each per-level image stack
};

struct AkPlan {
  int nlev = 0;
  AkLevel lev[kMaxLevels];
  float g16[9], g10[5];
  float gauss25[49];
  float win_ang1[64];
  int n_win = 0;
  uint16_t pair_tab[486 * 2];
  size_t total = 0;
};

int fround_h(float f) { return (int)(f + 0.5f); }

bool is_prime(int n) {
  if (n <= 3) return n > 1;
  if (n % 2 == 0 || n % 3 == 0) return false;
  for (int i = 5; i * i <= n; i += 6)
    if (n % i == 0 || n % (i + 2) == 0) return false;
  return true;
}

// fed_tau_by_process_time(T, 1, 0.25, reordering = true) of OpenCV's fed.cpp
int fed_tau(float T, float tau_max, float *tau) {
  const int n = (int)(ceilf(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);
  if (n <= 0) return 0;
  const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
  float tauh[64];
  const float c = 1.0f / (4.0f * (float)n + 2.0f);
  const float d = scale * tau_max / 2.0f;
  for (int k = 0; k < n; ++k) {
    const float hh = cosf(kPiF * (2.0f * (float)k + 1.0f) * c);
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

void gaussian_kernel(int ksize, float sigma, float *cf) {  // cv::getGaussianKernel(ksize, sigma, CV_32F)
  const double scale2x = -0.5 / ((double)sigma * sigma);
  double sum = 0;
  for (int i = 0; i < ksize; ++i) {
    const double x = i - (ksize - 1) * 0.5;
    cf[i] = (float)exp(scale2x * x * x);
    sum += cf[i];
  }
  sum = 1.0 / sum;
  for (int i = 0; i < ksize; ++i) cf[i] = (float)(cf[i] * sum);
}

// Allocate_Memory_Evolution + FED schedule + constant tables
void make_plan(int w, int h, int omax, int nsub, AkPlan &P) {
  const float soffset = 1.6f, derivative_factor = 1.5f;
  P.nlev = 0;
  P.total = 0;
  for (int i = 0; i < omax; ++i) {
    const float rfactor = 1.0f / powf(2.0f, (float)i);
    const int lh = (int)(h * rfactor), lw = (int)(w * rfactor);
    if ((lw < 80 || lh < 40) && i != 0) break;
    for (int j = 0; j < nsub && P.nlev < kMaxLevels; ++j) {
      AkLevel &L = P.lev[P.nlev++];
      memset(&L, 0, sizeof(L));
      L.w = lw;
      L.h = lh;
      L.esigma = soffset * powf(2.0f, (float)j / (float)nsub + (float)i);
      L.sigma_size = fround_h(L.esigma * derivative_factor / powf(2.0f, (float)i));
      L.etime = 0.5f * (L.esigma * L.esigma);
      L.octave = i;
      L.sublevel = j;
      L.off = P.total;
      P.total += (size_t)lw * lh;
    }
  }
  for (int i = 1; i < P.nlev; ++i)
    P.lev[i].nsteps = fed_tau(P.lev[i].etime - P.lev[i - 1].etime, 0.25f, P.lev[i].tsteps);
  gaussian_kernel(9, 1.6f, P.g16);
  gaussian_kernel(5, 1.0f, P.g10);
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j)
      P.gauss25[7 * i + j] = (float)(exp(-(double)(i * i + j * j) / 12.5) / (2.0 * 3.14159265358979323846 * 6.25));
  P.n_win = 0;
  for (float a = 0.0f; a < 2.0f * kPiF; a += 0.15f) P.win_ang1[P.n_win++] = a;
  // M-LDB comparison pairs: bit dpos compares values[a] > values[b]; values are laid out [cell][channel] with the
  // cells of the three grids back to back (4 + 9 + 16)
  int dpos = 0, cell0 = 0;
  const int counts[3] = {4, 9, 16};
  for (int lvl = 0; lvl < 3; ++lvl) {
    for (int pos = 0; pos < 3; ++pos)
      for (int a = 0; a < counts[lvl]; ++a)
        for (int b = a + 1; b < counts[lvl]; ++b) {
          P.pair_tab[2 * dpos] = (uint16_t)((cell0 + a) * 3 + pos);
          P.pair_tab[2 * dpos + 1] = (uint16_t)((cell0 + b) * 3 + pos);
          ++dpos;
        }
    cell0 += counts[lvl];
  }
}

// ---- fixed-order float math shared with the restatement -------------------------------------------------------
__device__ __forceinline__ int fround_d(float f) { return (int)(f + 0.5f); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int reflect101(int v, int n) {
  if (n == 1) return 0;
  while (v < 0 || v >= n) v = v < 0 ? -v : 2 * (n - 1) - v;
  return v;
}

__device__ float det_atanf(float x) {
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

__device__ float get_angle(float x, float y) {
  if (x == 0.0f && y == 0.0f) return 0.0f;
  if (x >= 0 && y >= 0) return (x == 0.0f) ? 1.5707963267948966f : det_atanf(y / x);
  if (x < 0 && y >= 0) return kPiF - det_atanf(-y / x);
  if (x < 0 && y < 0) return kPiF + det_atanf(y / x);
  return (x == 0.0f) ? (2.0f * kPiF - 1.5707963267948966f) : 2.0f * kPiF - det_atanf(-y / x);
}

__device__ void det_sincosf(float a, float *s, float *c) {
  const int k = (int)(a * 0.6366197723675814f + 0.5f);
  const float kf = (float)k;
  float r = a - kf * 1.5707963705062866f;
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

// ---- pixel kernels -----------------------------------------------------------------------------------------------
constexpr int kGradRows = 16;  // image rows one block of the contrast-factor kernels walks

struct Taps {
  float k[9];
};

__global__ void k_u8_to_f32(const uint8_t *__restrict__ src, float *__restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i] / 255.0f;
}

// one pass of the separable Gaussian, BORDER_REPLICATE; horizontal = 1: along x
__global__ void k_gauss_pass(const float *__restrict__ src, float *__restrict__ dst, int w, int h, Taps t, int ksize,
                             int horizontal) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int r = ksize / 2;
  float acc = 0.0f;
  for (int i = 0; i < ksize; ++i) {
    const float v = horizontal ? src[(size_t)y * w + clampi(x + i - r, 0, w - 1)]
                               : src[(size_t)clampi(y + i - r, 0, h - 1) * w + x];
    acc = acc + t.k[i] * v;
  }
  dst[(size_t)y * w + x] = acc;
}

__global__ void k_scharr(const float *__restrict__ src, float *__restrict__ dst, int w, int h, int xorder, int scale,
                         float ws, float wm) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int xm = reflect101(x - scale, w), xp = reflect101(x + scale, w);
  const int ym = reflect101(y - scale, h), yp = reflect101(y + scale, h);
  float d;
  if (xorder) {
    const float r0 = src[(size_t)y * w + xp] - src[(size_t)y * w + xm];
    const float rm = src[(size_t)ym * w + xp] - src[(size_t)ym * w + xm];
    const float rp = src[(size_t)yp * w + xp] - src[(size_t)yp * w + xm];
    d = wm * r0 + ws * (rm + rp);
  } else {
    const float r0 = src[(size_t)yp * w + x] - src[(size_t)ym * w + x];
    const float rm = src[(size_t)yp * w + xm] - src[(size_t)ym * w + xm];
    const float rp = src[(size_t)yp * w + xp] - src[(size_t)ym * w + xp];
    d = wm * r0 + ws * (rm + rp);
  }
  dst[(size_t)y * w + x] = d;
}

struct HalfsampleBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ src, int sw, int sh, float *__restrict__ dst, int dw, int dh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const double sx = (double)sw / dw, sy = (double)sh / dh;
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
};
__global__ void k_halfsample(const float *__restrict__ src, int sw, int sh, float *__restrict__ dst, int dw, int dh) {
  HalfsampleBody::run(src, sw, sh, dst, dw, dh);
}

// compute_k_percentile: maximum gradient magnitude over the interior, then its 300-bin histogram.
// One atomic per wave (max) / per block and bin (histogram): a single global word hit by every pixel serialises.
__global__ __launch_bounds__(128) void k_grad_max(const float *__restrict__ lx, const float *__restrict__ ly, int w,
                                                  int h, unsigned int *hmax_bits) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.0f;
  for (int y = blockIdx.y * kGradRows; y < min(h, (int)(blockIdx.y + 1) * kGradRows); ++y)
    if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
      const float a = lx[(size_t)y * w + x], b = ly[(size_t)y * w + x];
      m = fmaxf(m, sqrtf(a * a + b * b));
    }
  unsigned int bits = __float_as_uint(m);  // non-negative floats order like their bit patterns
  for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned int)__shfl_xor((int)bits, off, 64));
  if ((threadIdx.x & 63) == 0 && bits != 0) atomicMax(hmax_bits, bits);
}

__global__ __launch_bounds__(128) void k_grad_hist(const float *__restrict__ lx, const float *__restrict__ ly, int w,
                                                   int h, const unsigned int *hmax_bits,
                                                   unsigned int *hist /*[301]: 300 bins + npoints*/) {
  __shared__ unsigned int lh[301];
  for (int i = threadIdx.x; i < 301; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const float hmax = __uint_as_float(*hmax_bits);
  unsigned int mine = 0;
  for (int y = blockIdx.y * kGradRows; y < min(h, (int)(blockIdx.y + 1) * kGradRows); ++y)
    if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
      const float a = lx[(size_t)y * w + x], b = ly[(size_t)y * w + x];
      const float m = sqrtf(a * a + b * b);
      if (m != 0.0f) {
        int nbin = (int)floorf(300.0f * (m / hmax));
        if (nbin == 300) nbin--;
        atomicAdd(&lh[nbin], 1u);
        ++mine;
      }
    }
  if (mine) atomicAdd(&lh[300], mine);
  __syncthreads();
  for (int i = threadIdx.x; i < 301; i += blockDim.x)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// The start of Create_Nonlinear_Scale_Space in four launches instead of thirteen (u8 -> f32, 2 x 2 Gaussian passes, a
// copy, two Scharr passes, two fills, maximum, histogram, percentile).  Per pixel the arithmetic is that of the separate
// kernels above, in the same order, so every image and the contrast factor are the same bit for bit.
//  k_pre_rows   image (u8) -> row pass of the 9-tap Gaussian (level 0) and of the 5-tap one (contrast factor); clears
//               the histogram words
//  k_pre_cols   column passes -> Lt and Lsmooth of level 0, and the smoothed image of the contrast factor
//  k_pre_grad   Scharr x / y of that image -> gradient magnitude (interior pixels; 0 elsewhere) and its maximum
//  k_pre_hist   300-bin histogram of the magnitudes; the last workgroup to arrive turns it into kcontrast
struct Taps2 {
  float k9[9], k5[5];
};
__device__ __forceinline__ float scharr_at(const float *__restrict__ src, int w, int h, int x, int y, int xorder,
                                           int scale, float ws, float wm);
struct PreRowsBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const uint8_t *__restrict__ src, float *__restrict__ rows9, float *__restrict__ rows5, int w,
                   int h, const Taps2 &t, unsigned int *hist /*[304]*/) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0)
      for (int i = threadIdx.x; i < 304; i += blockDim.x) hist[i] = 0u;
    if (x >= w) return;
    float v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = (float)src[(size_t)y * w + clampi(x + i - 4, 0, w - 1)] / 255.0f;
    float a9 = 0.0f, a5 = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) a9 = a9 + t.k9[i] * v[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) a5 = a5 + t.k5[i] * v[i + 2];
    rows9[(size_t)y * w + x] = a9;
    rows5[(size_t)y * w + x] = a5;
  }
};
__global__ void k_pre_rows(const uint8_t *__restrict__ src, float *__restrict__ rows9, float *__restrict__ rows5, int w,
                   int h, Taps2 t, unsigned int *hist /*[304]*/) {
  PreRowsBody::run(src, rows9, rows5, w, h, t, hist);
}
struct PreColsBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ rows9, const float *__restrict__ rows5, float *__restrict__ lt0,
                   float *__restrict__ lsmooth0, float *__restrict__ sm5, int w, int h, const Taps2 &t) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    float a9 = 0.0f, a5 = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) a9 = a9 + t.k9[i] * rows9[(size_t)clampi(y + i - 4, 0, h - 1) * w + x];
#pragma unroll
    for (int i = 0; i < 5; ++i) a5 = a5 + t.k5[i] * rows5[(size_t)clampi(y + i - 2, 0, h - 1) * w + x];
    lt0[(size_t)y * w + x] = a9;
    lsmooth0[(size_t)y * w + x] = a9;
    sm5[(size_t)y * w + x] = a5;
  }
};
__global__ void k_pre_cols(const float *__restrict__ rows9, const float *__restrict__ rows5, float *__restrict__ lt0,
                   float *__restrict__ lsmooth0, float *__restrict__ sm5, int w, int h, Taps2 t) {
  PreColsBody::run(rows9, rows5, lt0, lsmooth0, sm5, w, h, t);
}
struct PreGradBody {
  static constexpr int kGangThreads = 128;
  static __device__ __forceinline__ void run(const float *__restrict__ sm5, float *__restrict__ mag, int w, int h,
                                          unsigned int *hmax_bits) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    for (int y = blockIdx.y * kGradRows; y < min(h, (int)(blockIdx.y + 1) * kGradRows); ++y) {
      if (x >= w) continue;
      float g = 0.0f;
      if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
        const float a = scharr_at(sm5, w, h, x, y, 1, 1, 3.0f, 10.0f);
        const float b = scharr_at(sm5, w, h, x, y, 0, 1, 3.0f, 10.0f);
        g = sqrtf(a * a + b * b);
        m = fmaxf(m, g);
      }
      mag[(size_t)y * w + x] = g;
    }
    unsigned int bits = __float_as_uint(m);  // non-negative floats order like their bit patterns
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, (unsigned int)__shfl_xor((int)bits, off, 64));
    if ((threadIdx.x & 63) == 0 && bits != 0) atomicMax(hmax_bits, bits);
  }
};
__global__ __launch_bounds__(128) void k_pre_grad(const float *__restrict__ sm5, float *__restrict__ mag, int w, int h,
                                          unsigned int *hmax_bits) {
  PreGradBody::run(sm5, mag, w, h, hmax_bits);
}
struct PreHistBody {
  static constexpr int kGangThreads = 128;
  static __device__ __forceinline__ void run(const float *__restrict__ mag, int w, int h,
                                          unsigned int *hist_all /*[0] hmax bits, [1..301] bins + npoints, [302] arrivals*/,
                                          float *kcontrast) {
    __shared__ unsigned int lh[301];
    __shared__ unsigned int ticket;
    for (int i = threadIdx.x; i < 301; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    unsigned int *hist = hist_all + 1;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const float hmax = __uint_as_float(hist_all[0]);
    unsigned int mine = 0;
    for (int y = blockIdx.y * kGradRows; y < min(h, (int)(blockIdx.y + 1) * kGradRows); ++y)
      if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
        const float m = mag[(size_t)y * w + x];
        if (m != 0.0f) {
          int nbin = (int)floorf(300.0f * (m / hmax));
          if (nbin == 300) nbin--;
          atomicAdd(&lh[nbin], 1u);
          ++mine;
        }
      }
    if (mine) atomicAdd(&lh[300], mine);
    __syncthreads();
    for (int i = threadIdx.x; i < 301; i += blockDim.x)
      if (lh[i]) atomicAdd(&hist[i], lh[i]);
    // every wave waits for its OWN histogram adds to have been performed (__syncthreads() does not wait on vmcnt on
    // gfx950); behind the barrier one lane counts the workgroup in (MI355X_MICROARCH.md, hand-over table, producer rule)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(&hist_all[302], 1u);
    __syncthreads();
    if (ticket != gridDim.x * gridDim.y - 1 || threadIdx.x != 0) return;
    __threadfence();
    // compute_k_percentile's tail (k_kcontrast), by the last workgroup; the histogram is read through atomics so that no
    // stale cached word is used
    const int npoints = (int)atomicAdd(&hist[300], 0u);
    const int nthreshold = (int)((float)npoints * 0.7f);
    int nelements = 0, k = 0;
    for (k = 0; nelements < nthreshold && k < 300; k++) nelements += (int)atomicAdd(&hist[k], 0u);
    *kcontrast = (nelements < nthreshold) ? 0.03f : hmax * ((float)k / 300.0f);
  }
};
__global__ __launch_bounds__(128) void k_pre_hist(const float *__restrict__ mag, int w, int h,
                                          unsigned int *hist_all /*[0] hmax bits, [1..301] bins + npoints, [302] arrivals*/,
                                          float *kcontrast) {
  PreHistBody::run(mag, w, h, hist_all, kcontrast);
}

__global__ void k_kcontrast(const unsigned int *hmax_bits, const unsigned int *hist, float *kcontrast) {
  const float hmax = __uint_as_float(*hmax_bits);
  const int npoints = (int)hist[300];
  const int nthreshold = (int)((float)npoints * 0.7f);
  int nelements = 0, k = 0;
  for (k = 0; nelements < nthreshold && k < 300; k++) nelements += (int)hist[k];
  *kcontrast = (nelements < nthreshold) ? 0.03f : hmax * ((float)k / 300.0f);
}

// nld_step_scalar: Ld_out = Ld + half_step * flux, zero flux across the image border -- up to K steps in one launch
// (temporal blocking): a 32 x 8 tile is loaded with a halo of K pixels, step s is computed in LDS on the tile grown by
// K - 1 - s pixels, the last step on the tile itself.  Per pixel the arithmetic is that of a step-per-launch kernel, so
// the images are the same bit for bit; the scale space of a VGA image is 166 steps, each shorter than a launch
// (measured: 1.03 -> 0.96 ms per VGA image at 4 steps per launch; 6 and 8 bring nothing more).
constexpr int kNldFuseMax = 16;
struct NldSteps {
  float half_step[kNldFuseMax];
};
constexpr int kNldTileX = 32, kNldTileY = 8;  // one output pixel per thread of a 256-thread workgroup

template <int K>
struct NldStepsBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ Ld, const float *__restrict__ c,
                                           float *__restrict__ Ld_out, int w, int h, const NldSteps &hs, int nsteps) {
    constexpr int TX = kNldTileX, TY = kNldTileY, RX = TX + 2 * K, RY = TY + 2 * K;
    __shared__ float sL[2][RY][RX + 1];
    __shared__ float sC[RY][RX + 1];
    const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;  // 32 x 8 threads
    const int x0 = blockIdx.x * TX - K, y0 = blockIdx.y * TY - K;
    for (int ly = ty; ly < RY; ly += TY)
      for (int lx = tx; lx < RX; lx += TX) {
        const int gx = x0 + lx, gy = y0 + ly;
        const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
        sL[0][ly][lx] = in ? Ld[(size_t)gy * w + gx] : 0.0f;
        sC[ly][lx] = in ? c[(size_t)gy * w + gx] : 0.0f;
      }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
      const int m = K - (nsteps - 1 - s);  // margin of the region this step still needs: the last step has m = K
      const float half_step = hs.half_step[s];
      for (int ly = m + ty; ly < RY - m; ly += TY)
        for (int lx = m + tx; lx < RX - m; lx += TX) {
          const int gx = x0 + lx, gy = y0 + ly;
          if (gx < 0 || gx >= w || gy < 0 || gy >= h) continue;
          const float cc = sC[ly][lx], v = sL[cur][ly][lx];
          float xpos = 0.0f, xneg = 0.0f, ypos = 0.0f, yneg = 0.0f;
          if (gx + 1 < w) xpos = (cc + sC[ly][lx + 1]) * (sL[cur][ly][lx + 1] - v);
          if (gx > 0) xneg = (sC[ly][lx - 1] + cc) * (v - sL[cur][ly][lx - 1]);
          if (gy + 1 < h) ypos = (cc + sC[ly + 1][lx]) * (sL[cur][ly + 1][lx] - v);
          if (gy > 0) yneg = (sC[ly - 1][lx] + cc) * (v - sL[cur][ly - 1][lx]);
          const float stp = half_step * (((xpos - xneg) + ypos) - yneg);
          sL[cur ^ 1][ly][lx] = v + stp;
        }
      __syncthreads();
      cur ^= 1;
    }
    const int gx = x0 + K + tx, gy = y0 + K + ty;
    if (gx < w && gy < h) Ld_out[(size_t)gy * w + gx] = sL[cur][K + ty][K + tx];
  }
};
template <int K>
__global__ __launch_bounds__(256) void k_nld_steps(const float *__restrict__ Ld, const float *__restrict__ c,
                                           float *__restrict__ Ld_out, int w, int h, NldSteps hs, int nsteps) {
  NldStepsBody<K>::run(Ld, c, Ld_out, w, h, hs, nsteps);
}

// Compute_Multiscale_Derivatives + Compute_Determinant_Hessian_Response, fused.  The reference runs five Scharr
// filters per level (Lx, Ly from Lsmooth; Lxx, Lyy, Lxy from those), scales them by sigma / sigma^2 and forms the
// determinant; here one kernel produces Lx and Ly, a second the determinant straight from them -- the second
// derivatives never touch memory, and Lx / Ly stay unscaled in memory (their only later reader, the descriptor
// kernel, applies the level's factor with the same multiplication the reference's in-place scaling does).
__device__ __forceinline__ float scharr_at(const float *__restrict__ src, int w, int h, int x, int y, int xorder,
                                           int scale, float ws, float wm) {
  const int xm = reflect101(x - scale, w), xp = reflect101(x + scale, w);
  const int ym = reflect101(y - scale, h), yp = reflect101(y + scale, h);
  if (xorder) {
    const float r0 = src[(size_t)y * w + xp] - src[(size_t)y * w + xm];
    const float rm = src[(size_t)ym * w + xp] - src[(size_t)ym * w + xm];
    const float rp = src[(size_t)yp * w + xp] - src[(size_t)yp * w + xm];
    return wm * r0 + ws * (rm + rp);
  }
  const float r0 = src[(size_t)yp * w + x] - src[(size_t)ym * w + x];
  const float rm = src[(size_t)yp * w + xm] - src[(size_t)ym * w + xm];
  const float rp = src[(size_t)yp * w + xp] - src[(size_t)ym * w + xp];
  return wm * r0 + ws * (rm + rp);
}

// gaussian_2D_convolution(5 x 5, separable, BORDER_REPLICATE) of the level's start image -> Lsmooth, and the
// Perona-Malik g2 conductivity 1 / (1 + |grad Lsmooth|^2 / k^2) from it (Scharr, scale 1), in one launch: a 32 x 8
// tile with the source halo in LDS (3 px: 2 for the taps of the column pass's rows / the row pass's columns, 1 for the
// Scharr stencil on Lsmooth).  The row pass, the column pass and the Scharr sums run in the order a kernel per pass
// would use, so the images are the same bit for bit.
struct SmoothFlowBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ src, float *__restrict__ lsmooth,
                                             float *__restrict__ g2, int w, int h, const Taps &t,
                                             const float *kcontrast, int octave) {
    constexpr int TX = 32, TY = 8;
    constexpr int SX = TX + 6, SY = TY + 6;  // source region
    constexpr int MX = TX + 2, MY = TY + 6;  // row-pass region: Lsmooth's columns, the column pass's rows
    constexpr int LX = TX + 2, LY = TY + 2;  // Lsmooth region
    __shared__ float sS[SY][SX + 1];
    __shared__ float sM[MY][MX + 1];
    __shared__ float sL[LY][LX + 1];
    const int X0 = blockIdx.x * TX, Y0 = blockIdx.y * TY;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int ly = ty; ly < SY; ly += TY)
      for (int lx = tx; lx < SX; lx += TX) {
        const int gx = X0 - 3 + lx, gy = Y0 - 3 + ly;
        sS[ly][lx] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? src[(size_t)gy * w + gx] : 0.0f;
      }
    __syncthreads();
    for (int ly = ty; ly < MY; ly += TY)  // row pass at (X0 - 1 + lx, Y0 - 3 + ly)
      for (int lx = tx; lx < MX; lx += TX) {
        const int gx = X0 - 1 + lx, gy = Y0 - 3 + ly;
        if (gx < 0 || gx >= w || gy < 0 || gy >= h) continue;
        float acc = 0.0f;
        for (int i = 0; i < 5; ++i) acc = acc + t.k[i] * sS[ly][clampi(gx + i - 2, 0, w - 1) - (X0 - 3)];
        sM[ly][lx] = acc;
      }
    __syncthreads();
    for (int ly = ty; ly < LY; ly += TY)  // column pass at (X0 - 1 + lx, Y0 - 1 + ly)
      for (int lx = tx; lx < LX; lx += TX) {
        const int gx = X0 - 1 + lx, gy = Y0 - 1 + ly;
        if (gx < 0 || gx >= w || gy < 0 || gy >= h) continue;
        float acc = 0.0f;
        for (int i = 0; i < 5; ++i) acc = acc + t.k[i] * sM[clampi(gy + i - 2, 0, h - 1) - (Y0 - 3)][lx];
        sL[ly][lx] = acc;
        if (lx >= 1 && lx <= TX && ly >= 1 && ly <= TY) lsmooth[(size_t)gy * w + gx] = acc;
      }
    __syncthreads();
    const int x = X0 + tx, y = Y0 + ty;
    if (x >= w || y >= h) return;
    const int xm = reflect101(x - 1, w) - (X0 - 1), xp = reflect101(x + 1, w) - (X0 - 1), xc = tx + 1;
    const int ym = reflect101(y - 1, h) - (Y0 - 1), yp = reflect101(y + 1, h) - (Y0 - 1), yc = ty + 1;
    const float ws = 3.0f, wm = 10.0f;
    float lx_, ly_;
    {
      const float r0 = sL[yc][xp] - sL[yc][xm], rm = sL[ym][xp] - sL[ym][xm], rp = sL[yp][xp] - sL[yp][xm];
      lx_ = wm * r0 + ws * (rm + rp);
    }
    {
      const float r0 = sL[yp][xc] - sL[ym][xc], rm = sL[yp][xm] - sL[ym][xm], rp = sL[yp][xp] - sL[ym][xp];
      ly_ = wm * r0 + ws * (rm + rp);
    }
    float kc = *kcontrast;
    for (int o = 0; o < octave; ++o) kc = kc * 0.75f;
    const float inv_k = 1.0f / (kc * kc);
    g2[(size_t)y * w + x] = 1.0f / (1.0f + inv_k * (lx_ * lx_ + ly_ * ly_));
  }
};
__global__ __launch_bounds__(256) void k_smooth_flow(const float *__restrict__ src, float *__restrict__ lsmooth,
                                             float *__restrict__ g2, int w, int h, Taps t,
                                             const float *kcontrast, int octave) {
  SmoothFlowBody::run(src, lsmooth, g2, w, h, t, kcontrast, octave);
}

// k_smooth_flow and the level's first k_nld_steps in ONE launch: a 32 x 8 tile computes Lsmooth and the conductivity on
// itself grown by the K pixels of halo the K diffusion steps need (so the smoothing of the halo is done again by the
// neighbouring tiles: ~3x the arithmetic of a level's cheapest pass, against a dependent launch per level -- the
// extraction is a chain of launches, and with images in the whole path is bound by launches per second).  Lsmooth and
// the conductivity of the tile proper go to global memory as before (derivatives later; further step launches of the
// level).  Per pixel the arithmetic is that of the two kernels, in the same order: the images are the same bit for bit.
template <int K>
struct SmoothNldBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ src, float *__restrict__ lsmooth,
                                            float *__restrict__ g2, float *__restrict__ Ld_out, int w, int h,
                                            const Taps &t, const float *kcontrast, int octave, const NldSteps &hs, int nsteps) {
    constexpr int TX = kNldTileX, TY = kNldTileY;
    constexpr int CX = TX + 2 * K, CY = TY + 2 * K;              // conductivity / evolving image: tile + K
    constexpr int LX = CX + 2, LY = CY + 2;                      // Lsmooth: + 1 (Scharr)
    constexpr int MX = LX, MY = LY + 4;                          // row pass: Lsmooth's columns, + 2 rows (column taps)
    constexpr int SX = MX + 4, SY = MY;                          // source: + 2 columns (row taps)
    __shared__ float sS[SY][SX + 1];
    __shared__ float sM[MY][MX + 1];
    __shared__ float sLs[LY][LX + 1];
    __shared__ float sC[CY][CX + 1];
    __shared__ float sL[2][CY][CX + 1];
    const int X0 = blockIdx.x * TX, Y0 = blockIdx.y * TY;        // the tile
    const int sx0 = X0 - K - 3, sy0 = Y0 - K - 3;                // origins of the regions in the image
    const int mx0 = X0 - K - 1, my0 = sy0;
    const int lx0 = mx0, ly0 = Y0 - K - 1;
    const int cx0 = X0 - K, cy0 = Y0 - K;
    const int tid = threadIdx.x;
    for (int i = tid; i < SX * SY; i += 256) {
      const int ly = i / SX, lx = i - ly * SX;
      const int gx = sx0 + lx, gy = sy0 + ly;
      sS[ly][lx] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? src[(size_t)gy * w + gx] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < MX * MY; i += 256) {  // row pass
      const int ly = i / MX, lx = i - ly * MX;
      const int gx = mx0 + lx, gy = my0 + ly;
      if (gx < 0 || gx >= w || gy < 0 || gy >= h) continue;
      float acc = 0.0f;
      for (int q = 0; q < 5; ++q) acc = acc + t.k[q] * sS[ly][clampi(gx + q - 2, 0, w - 1) - sx0];
      sM[ly][lx] = acc;
    }
    __syncthreads();
    for (int i = tid; i < LX * LY; i += 256) {  // column pass
      const int ly = i / LX, lx = i - ly * LX;
      const int gx = lx0 + lx, gy = ly0 + ly;
      if (gx < 0 || gx >= w || gy < 0 || gy >= h) continue;
      float acc = 0.0f;
      for (int q = 0; q < 5; ++q) acc = acc + t.k[q] * sM[clampi(gy + q - 2, 0, h - 1) - my0][lx];
      sLs[ly][lx] = acc;
      if (gx >= X0 && gx < X0 + TX && gy >= Y0 && gy < Y0 + TY) lsmooth[(size_t)gy * w + gx] = acc;
    }
    __syncthreads();
    float kc = *kcontrast;
    for (int o = 0; o < octave; ++o) kc = kc * 0.75f;
    const float inv_k = 1.0f / (kc * kc);
    for (int i = tid; i < CX * CY; i += 256) {  // conductivity, and the evolving image's start
      const int ly = i / CX, lx = i - ly * CX;
      const int x = cx0 + lx, y = cy0 + ly;
      const bool in = x >= 0 && x < w && y >= 0 && y < h;
      sL[0][ly][lx] = in ? sS[y - sy0][x - sx0] : 0.0f;
      float c = 0.0f;
      if (in) {
        const int xm = reflect101(x - 1, w) - lx0, xp = reflect101(x + 1, w) - lx0, xc = x - lx0;
        const int ym = reflect101(y - 1, h) - ly0, yp = reflect101(y + 1, h) - ly0, yc = y - ly0;
        const float ws = 3.0f, wm = 10.0f;
        float lx_, ly_;
        {
          const float r0 = sLs[yc][xp] - sLs[yc][xm], rm = sLs[ym][xp] - sLs[ym][xm], rp = sLs[yp][xp] - sLs[yp][xm];
          lx_ = wm * r0 + ws * (rm + rp);
        }
        {
          const float r0 = sLs[yp][xc] - sLs[ym][xc], rm = sLs[yp][xm] - sLs[ym][xm], rp = sLs[yp][xp] - sLs[ym][xp];
          ly_ = wm * r0 + ws * (rm + rp);
        }
        c = 1.0f / (1.0f + inv_k * (lx_ * lx_ + ly_ * ly_));
        if (x >= X0 && x < X0 + TX && y >= Y0 && y < Y0 + TY) g2[(size_t)y * w + x] = c;
      }
      sC[ly][lx] = c;
    }
    __syncthreads();
    // the steps, as in k_nld_steps (region of step s: the tile grown by K - 1 - s ... of the steps still to come)
    const int tx = tid & (TX - 1), ty = tid / TX;
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
      const int m = K - (nsteps - 1 - s);
      const float half_step = hs.half_step[s];
      for (int ly = m + ty; ly < CY - m; ly += TY)
        for (int lx = m + tx; lx < CX - m; lx += TX) {
          const int gx = cx0 + lx, gy = cy0 + ly;
          if (gx < 0 || gx >= w || gy < 0 || gy >= h) continue;
          const float cc = sC[ly][lx], v = sL[cur][ly][lx];
          float xpos = 0.0f, xneg = 0.0f, ypos = 0.0f, yneg = 0.0f;
          if (gx + 1 < w) xpos = (cc + sC[ly][lx + 1]) * (sL[cur][ly][lx + 1] - v);
          if (gx > 0) xneg = (sC[ly][lx - 1] + cc) * (v - sL[cur][ly][lx - 1]);
          if (gy + 1 < h) ypos = (cc + sC[ly + 1][lx]) * (sL[cur][ly + 1][lx] - v);
          if (gy > 0) yneg = (sC[ly - 1][lx] + cc) * (v - sL[cur][ly - 1][lx]);
          const float stp = half_step * (((xpos - xneg) + ypos) - yneg);
          sL[cur ^ 1][ly][lx] = v + stp;
        }
      __syncthreads();
      cur ^= 1;
    }
    const int gx = X0 + tx, gy = Y0 + ty;
    if (gx < w && gy < h) Ld_out[(size_t)gy * w + gx] = sL[cur][K + ty][K + tx];
  }
};
template <int K>
__global__ __launch_bounds__(256) void k_smooth_nld(const float *__restrict__ src, float *__restrict__ lsmooth,
                                            float *__restrict__ g2, float *__restrict__ Ld_out, int w, int h,
                                            Taps t, const float *kcontrast, int octave, NldSteps hs, int nsteps) {
  SmoothNldBody<K>::run(src, lsmooth, g2, Ld_out, w, h, t, kcontrast, octave, hs, nsteps);
}

// A whole octave in ONE launch when its image fits in LDS three times (VGA: the 80 x 60 octave -- 4 800 pixels, but 99 of
// the schedule's 165 diffusion steps, each ~2 us as a launch or a fused part of one: 213 us of a 760 us extraction).  One
// 1024-thread workgroup keeps the evolving image, a second copy for the ping-pong and the conductivity resident and
// walks the octave's levels: 5 x 5 Gaussian (row pass, column pass) -> Lsmooth (stored), Scharr -> conductivity, the
// level's FED steps, Lt stored; the next level starts from what is already in LDS.  Per pixel the arithmetic is that of
// k_smooth_flow and k_nld_steps, in the same order, so the images are the same bit for bit.
constexpr int kOctaveRunMax = 8;  // levels of one octave
constexpr int kResidentPix = 13;  // pixels per thread: 150 KB / 12 B / 1024 threads
struct OctaveRun {
  int n_levels, w, h, octave;
  unsigned int off[kOctaveRunMax];  // offset of the level in the per-level image stacks
  int nsteps[kOctaveRunMax];
  int step0[kOctaveRunMax];         // index of the level's first half step in the table
};

struct OctaveResidentBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(const float *__restrict__ start, float *__restrict__ Lt_all,
                                                  float *__restrict__ Lsmooth_all, const Taps &t,
                                                  const float *__restrict__ kcontrast,
                                                  const float *__restrict__ half_steps, const OctaveRun &R) {
    extern __shared__ float lds_f[];
    const int w = R.w, h = R.h, n = w * h;
    float *A = lds_f, *B = lds_f + n, *C = lds_f + 2 * n;
    const int tid = threadIdx.x;
    for (int p = tid; p < n; p += 1024) A[p] = start[p];
    float kc = *kcontrast;
    for (int o = 0; o < R.octave; ++o) kc = kc * 0.75f;
    const float inv_k = 1.0f / (kc * kc);
    __syncthreads();
    for (int lv = 0; lv < R.n_levels; ++lv) {
      // row pass A -> B, column pass B -> C (= Lsmooth)
      for (int p = tid; p < n; p += 1024) {
        const int y = p / w, x = p - y * w;
        float acc = 0.0f;
        for (int i = 0; i < 5; ++i) acc = acc + t.k[i] * A[y * w + clampi(x + i - 2, 0, w - 1)];
        B[p] = acc;
      }
      __syncthreads();
      float *const lsm = Lsmooth_all + R.off[lv];
      for (int p = tid; p < n; p += 1024) {
        const int y = p / w, x = p - y * w;
        float acc = 0.0f;
        for (int i = 0; i < 5; ++i) acc = acc + t.k[i] * B[clampi(y + i - 2, 0, h - 1) * w + x];
        C[p] = acc;
        lsm[p] = acc;
      }
      __syncthreads();
      // conductivity from Lsmooth (C) -> B
      for (int p = tid; p < n; p += 1024) {
        const int y = p / w, x = p - y * w;
        const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
        const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        const float ws = 3.0f, wm = 10.0f;
        float lx_, ly_;
        {
          const float r0 = C[y * w + xp] - C[y * w + xm], rm = C[ym * w + xp] - C[ym * w + xm],
                      rp = C[yp * w + xp] - C[yp * w + xm];
          lx_ = wm * r0 + ws * (rm + rp);
        }
        {
          const float r0 = C[yp * w + x] - C[ym * w + x], rm = C[yp * w + xm] - C[ym * w + xm],
                      rp = C[yp * w + xp] - C[ym * w + xp];
          ly_ = wm * r0 + ws * (rm + rp);
        }
        B[p] = 1.0f / (1.0f + inv_k * (lx_ * lx_ + ly_ * ly_));
      }
      __syncthreads();
      // FED steps: A -> C -> A ...  The conductivity does not change within a level, so a thread keeps the four sums
      // (c + c_neighbour) of each of its pixels in registers; a neighbour outside the image is replaced by the pixel
      // itself, whose difference is +0 and gives the +0 flux the per-launch kernel writes there.
      float cxp[kResidentPix], cxn[kResidentPix], cyp[kResidentPix], cyn[kResidentPix];
      int nb[kResidentPix];  // bit 0..3: neighbour x+1 / x-1 / y+1 / y-1 exists
#pragma unroll
      for (int k = 0; k < kResidentPix; ++k) {
        const int p = tid + k * 1024;
        cxp[k] = cxn[k] = cyp[k] = cyn[k] = 0.0f;
        nb[k] = 0;
        if (p < n) {
          const int y = p / w, x = p - y * w;
          const float cc = B[p];
          int f = 0;
          if (x + 1 < w) { cxp[k] = cc + B[p + 1]; f |= 1; }
          if (x > 0) { cxn[k] = B[p - 1] + cc; f |= 2; }
          if (y + 1 < h) { cyp[k] = cc + B[p + w]; f |= 4; }
          if (y > 0) { cyn[k] = B[p - w] + cc; f |= 8; }
          nb[k] = f;
        }
      }
      float *cur = A, *nxt = C;
      for (int s = 0; s < R.nsteps[lv]; ++s) {
        const float half_step = half_steps[R.step0[lv] + s];
#pragma unroll
        for (int k = 0; k < kResidentPix; ++k) {
          const int p = tid + k * 1024;
          if (p < n) {
            const int f = nb[k];
            const float v = cur[p];
            const float xpos = cxp[k] * (cur[(f & 1) ? p + 1 : p] - v);
            const float xneg = cxn[k] * (v - cur[(f & 2) ? p - 1 : p]);
            const float ypos = cyp[k] * (cur[(f & 4) ? p + w : p] - v);
            const float yneg = cyn[k] * (v - cur[(f & 8) ? p - w : p]);
            const float stp = half_step * (((xpos - xneg) + ypos) - yneg);
            nxt[p] = v + stp;
          }
        }
        __syncthreads();
        float *const tswap = cur;
        cur = nxt;
        nxt = tswap;
      }
      float *const lt = Lt_all + R.off[lv];
      for (int p = tid; p < n; p += 1024) lt[p] = cur[p];
      // the next level starts from `cur`; its scratch is the other image
      A = cur;
      C = nxt;
      __syncthreads();
    }
  }
};
__global__ __launch_bounds__(1024) void k_octave_resident(const float *__restrict__ start, float *__restrict__ Lt_all,
                                                  float *__restrict__ Lsmooth_all, Taps t,
                                                  const float *__restrict__ kcontrast,
                                                  const float *__restrict__ half_steps, OctaveRun R) {
  OctaveResidentBody::run(start, Lt_all, Lsmooth_all, t, kcontrast, half_steps, R);
}

// Compute_Multiscale_Derivatives, Compute_Determinant_Hessian_Response and the candidate pass of
// Find_Scale_Space_Extrema run on the finished scale space and are independent between evolution levels: one launch over
// ALL levels each (blockIdx.y = a row of the stacked images) instead of one per level -- 48 launches become 3.
struct LevelTab {
  int n;
  int row0[kMaxLevels + 1];  // first stacked row of each level
  int w[kMaxLevels], h[kMaxLevels], sc[kMaxLevels], sigma_size[kMaxLevels];
  unsigned int off[kMaxLevels];
  float ws[kMaxLevels], wm[kMaxLevels];
};

__device__ __forceinline__ int level_of_row(const LevelTab &T, int row) {
  int i = 0;
  while (i + 1 < T.n && row >= T.row0[i + 1]) ++i;
  return i;
}

struct ScharrXyAllBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ ls, float *__restrict__ lx, float *__restrict__ ly, const LevelTab *__restrict__ Tp) {
    const LevelTab &T = *Tp;  // (in device memory, one per extractor: a gang launch carries a pointer per frame, not a table)
    const int i = level_of_row(T, blockIdx.y);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y - T.row0[i], w = T.w[i], h = T.h[i];
    if (x >= w) return;
    const float *src = ls + T.off[i];
    lx[T.off[i] + (size_t)y * w + x] = scharr_at(src, w, h, x, y, 1, T.sc[i], T.ws[i], T.wm[i]);
    ly[T.off[i] + (size_t)y * w + x] = scharr_at(src, w, h, x, y, 0, T.sc[i], T.ws[i], T.wm[i]);
  }
};
__global__ void k_scharr_xy_all(const float *__restrict__ ls, float *__restrict__ lx, float *__restrict__ ly,
                                const LevelTab *__restrict__ Tp) {
  ScharrXyAllBody::run(ls, lx, ly, Tp);
}

struct HessianDetAllBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ lx_all, const float *__restrict__ ly_all,
                          float *__restrict__ ldet, const LevelTab *__restrict__ Tp) {
    const LevelTab &T = *Tp;
    const int i = level_of_row(T, blockIdx.y);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y - T.row0[i], w = T.w[i], h = T.h[i];
    if (x >= w) return;
    const float *lx = lx_all + T.off[i], *ly = ly_all + T.off[i];
    const int scale = T.sc[i];
    const float ws = T.ws[i], wm = T.wm[i], sf2 = (float)(scale * scale);
    const float lxx = scharr_at(lx, w, h, x, y, 1, scale, ws, wm);
    const float lyy = scharr_at(ly, w, h, x, y, 0, scale, ws, wm);
    const float lxy = scharr_at(lx, w, h, x, y, 0, scale, ws, wm);
    const float a = lxx * sf2, b = lxy * sf2, c = lyy * sf2;
    ldet[T.off[i] + (size_t)y * w + x] = a * c - b * b;
  }
};
__global__ void k_hessian_det_all(const float *__restrict__ lx_all, const float *__restrict__ ly_all,
                          float *__restrict__ ldet, const LevelTab *__restrict__ Tp) {
  HessianDetAllBody::run(lx_all, ly_all, ldet, Tp);
}

struct Candidate9 {
  int level, x, y, pad;
  float patch[9];  // Ldet(y-1..y+1, x-1..x+1), row major
  float pad2[3];
};

struct ExtremaAllBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ ldet_all, const LevelTab *__restrict__ Tp, float dthreshold, Candidate9 *out,
                      unsigned int cap, unsigned int *n_out) {
    const LevelTab &T = *Tp;
    const int i = level_of_row(T, blockIdx.y);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y - T.row0[i], w = T.w[i], h = T.h[i];
    if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1) return;
    const float *D = ldet_all + T.off[i];
    const float v = D[(size_t)y * w + x];
    if (!(v > dthreshold && v >= 0.00001f)) return;
    float p[9];
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) p[(dy + 1) * 3 + dx + 1] = D[(size_t)(y + dy) * w + x + dx];
    if (!(v > p[3] && v > p[5] && v > p[0] && v > p[1] && v > p[2] && v > p[6] && v > p[7] && v > p[8])) return;
    const float smax = 10.0f * sqrtf(2.0f);
    const int sigma_size_ = T.sigma_size[i];
    const int left_x = fround_d((float)x - smax * sigma_size_) - 1, right_x = fround_d((float)x + smax * sigma_size_) + 1;
    const int up_y = fround_d((float)y - smax * sigma_size_) - 1, down_y = fround_d((float)y + smax * sigma_size_) + 1;
    if (left_x < 0 || right_x >= w || up_y < 0 || down_y >= h) return;
    const unsigned int slot = atomicAdd(n_out, 1u);
    if (slot >= cap) return;
    Candidate9 c;
    c.level = i;
    c.x = x;
    c.y = y;
    c.pad = 0;
    for (int k = 0; k < 9; ++k) c.patch[k] = p[k];
    c.pad2[0] = c.pad2[1] = c.pad2[2] = 0.0f;
    out[slot] = c;
  }
};
__global__ void k_extrema_all(const float *__restrict__ ldet_all, const LevelTab *__restrict__ Tp, float dthreshold, Candidate9 *out,
                      unsigned int cap, unsigned int *n_out) {
  ExtremaAllBody::run(ldet_all, Tp, dthreshold, out, cap, n_out);
}

struct DevLevel {
  const float *Lt, *Lx, *Ly;  // Lx, Ly unscaled: multiply by sf (= sigma_size) on use
  int w, h, octave;
  float sf;
};
struct DevLevels {
  DevLevel l[kMaxLevels];
};

__device__ __forceinline__ float at_clamped(const float *img, int w, int h, int y, int x) {
  return img[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}

// Compute_Main_Orientation + Get_MLDB_Full_Descriptor: one wave per keypoint
// kp [n x 4] = x, y, size (diameter), class_id ; angle_out [n] ; desc [n x 64] (61 bytes + 3 zero bytes = .desc row)
struct OrientDescribeBody {
  static constexpr int kGangThreads = 64;
  static __device__ __forceinline__ void run(const DevLevels *__restrict__ LVp, const float *__restrict__ kp, int n,
                                        const float *__restrict__ gauss25,
                                        const float *__restrict__ win_ang1, int n_win,
                                        const uint16_t *__restrict__ pair_tab,
                                        float *__restrict__ angle_out, uint8_t *__restrict__ desc) {
    __shared__ float resX[109], resY[109], Ang[109];
    __shared__ float vals[29 * 3];
    const int kidx = blockIdx.x;
    if (kidx >= n) return;
    const int lane = threadIdx.x;
    const float kx = kp[4 * kidx], ky = kp[4 * kidx + 1], ksize = kp[4 * kidx + 2];
    const int level = (int)kp[4 * kidx + 3];
    const DevLevel L = LVp->l[level];
    const float ratio = (float)(1 << L.octave);
    const int s = fround_d(0.5f * ksize / ratio);
    const float xf = kx / ratio, yf = ky / ratio;
    // --- orientation: 109 samples of the disc of radius 6 s ---
    for (int q = lane; q < 109; q += 64) {
      // q-th (i, j) of the double loop i = -6..6, j = -6..6 with i*i + j*j < 36
      int cnt = 0, ii = 0, jj = 0;
      for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
          if (i * i + j * j < 36) {
            if (cnt == q) {
              ii = i;
              jj = j;
            }
            ++cnt;
          }
      const int iy = fround_d(yf + (float)(jj * s)), ix = fround_d(xf + (float)(ii * s));
      const int a = ii < 0 ? -ii : ii, b = jj < 0 ? -jj : jj;  // id[] = |.| mirrored table index
      const float g = gauss25[7 * a + b];
      const float rx = g * (at_clamped(L.Lx, L.w, L.h, iy, ix) * L.sf);
      const float ry = g * (at_clamped(L.Ly, L.w, L.h, iy, ix) * L.sf);
      resX[q] = rx;
      resY[q] = ry;
      Ang[q] = get_angle(rx, ry);
    }
    __syncthreads();
    const float two_pi = 2.0f * kPiF;
    float mag = -1.0f, sumX = 0.0f, sumY = 0.0f;
    if (lane < n_win) {
      const float ang1 = win_ang1[lane];
      const float ang2 = (ang1 + kPiF / 3.0f > two_pi) ? ang1 - 5.0f * kPiF / 3.0f : ang1 + kPiF / 3.0f;
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
      mag = sumX * sumX + sumY * sumY;
    }
    // first window (in sweep order) with the strictly largest magnitude; "max" starts at 0, so an all-zero sweep keeps angle 0
    float best = mag;
    int best_lane = lane;
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int ol = __shfl_xor(best_lane, off, 64);
      if (ob > best || (ob == best && ol < best_lane)) {
        best = ob;
        best_lane = ol;
      }
    }
    const float wx = __shfl(sumX, best_lane, 64), wy = __shfl(sumY, best_lane, 64);
    const float angle = (best > 0.0f) ? get_angle(wx, wy) : 0.0f;
    if (lane == 0) angle_out[kidx] = angle;
    // --- M-LDB: 4 + 9 + 16 grid cells, one lane per cell, samples summed in (k, l) order ---
    float si, co;
    det_sincosf(angle, &si, &co);
    const int scale = s;
    if (lane < 29) {
      int lvl, cell;
      if (lane < 4) {
        lvl = 0;
        cell = lane;
      } else if (lane < 13) {
        lvl = 1;
        cell = lane - 4;
      } else {
        lvl = 2;
        cell = lane - 13;
      }
      const int pattern = 10;
      const int step = lvl == 0 ? 10 : (lvl == 1 ? 7 : 5);
      const int per = lvl + 2;               // cells per side
      const int i0 = -pattern + (cell / per) * step, j0 = -pattern + (cell % per) * step;
      float di = 0.0f, dx = 0.0f, dy = 0.0f;
      int ns = 0;
      for (int kk = i0; kk < i0 + step; ++kk)
        for (int l = j0; l < j0 + step; ++l) {
          const float sample_y = yf + ((float)l * co * (float)scale + (float)kk * si * (float)scale);
          const float sample_x = xf + (-(float)l * si * (float)scale + (float)kk * co * (float)scale);
          const int y1 = fround_d(sample_y), x1 = fround_d(sample_x);
          const float ri = at_clamped(L.Lt, L.w, L.h, y1, x1);
          const float rx = at_clamped(L.Lx, L.w, L.h, y1, x1) * L.sf;
          const float ry = at_clamped(L.Ly, L.w, L.h, y1, x1) * L.sf;
          di += ri;
          const float rry = rx * co + ry * si;
          const float rrx = -rx * si + ry * co;
          dx += rrx;
          dy += rry;
          ns++;
        }
      vals[lane * 3 + 0] = di / (float)ns;
      vals[lane * 3 + 1] = dx / (float)ns;
      vals[lane * 3 + 2] = dy / (float)ns;
    }
    __syncthreads();
    {
      uint8_t byte = 0;
      if (lane < 61) {
        for (int b = 0; b < 8; ++b) {
          const int dpos = 8 * lane + b;
          if (dpos < 486 && vals[pair_tab[2 * dpos]] > vals[pair_tab[2 * dpos + 1]]) byte |= (uint8_t)(1 << b);
        }
      }
      desc[(size_t)kidx * 64 + lane] = byte;  // lanes 61..63 write the zero padding of the .desc row
    }
  }
};
__global__ __launch_bounds__(64) void k_orient_describe(const DevLevels *__restrict__ LVp, const float *__restrict__ kp, int n,
                                        const float *__restrict__ gauss25,
                                        const float *__restrict__ win_ang1, int n_win,
                                        const uint16_t *__restrict__ pair_tab,
                                        float *__restrict__ angle_out, uint8_t *__restrict__ desc) {
  OrientDescribeBody::run(LVp, kp, n, gauss25, win_ang1, n_win, pair_tab, angle_out, desc);
}

}  // namespace

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct Akaze : GangMember {  // (gang.h: `stream` reads as the stream to queue on now; an extractor can record for a gang)
  int device = 0;
  int w = 0, h = 0, omax = 4, nsub = 4;
  float thres = 0.001f;
  AkPlan plan;
  hipStream_t own_stream = nullptr;  // the one created with the extractor (stream may be a context's, see _share_stream)
  uint8_t *d_gray = nullptr;
  float *d_img = nullptr, *d_t0 = nullptr, *d_t1 = nullptr, *d_t2 = nullptr, *d_t3 = nullptr;
  float *d_Lt = nullptr, *d_Lsmooth = nullptr, *d_Lx = nullptr, *d_Ly = nullptr, *d_Ldet = nullptr;  // per-level stacks
  unsigned int *d_hist = nullptr;             // [0] hmax bits, [1..301] histogram + npoints
  float *d_kcontrast = nullptr;
  float *d_half_steps = nullptr;              // [level][64]: 0.5 * tsteps, for k_octave_resident
  void *d_level_tab = nullptr;                // the LevelTab of this image size (the all-level kernels read it by pointer)
  void *d_dev_levels = nullptr;               // the DevLevels of this extractor (k_orient_describe reads it by pointer)
  Candidate9 *d_cand = nullptr;
  unsigned int *d_ncand = nullptr;
  unsigned int cand_cap = 1u << 16;
  float *d_gauss25 = nullptr, *d_win = nullptr;
  uint16_t *d_pair = nullptr;
  float *d_kp = nullptr, *d_angle = nullptr;
  uint8_t *d_desc = nullptr;
  unsigned int kp_cap = 0;
};

namespace {

#define AK_HIP(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return e_ == hipErrorOutOfMemory ? SFMLOC_ENOMEM : SFMLOC_EHIP;          \
    }                                                                          \
  } while (0)

dim3 grid2(int w, int h) { return dim3((w + 127) / 128, h); }

int gauss(Akaze *a, const float *src, float *dst, float *tmp, int w, int h, const float *k, int ksize) {
  Taps t;
  for (int i = 0; i < 9; ++i) t.k[i] = i < ksize ? k[i] : 0.0f;
  hipLaunchKernelGGL(k_gauss_pass, grid2(w, h), dim3(128), 0, a->stream, src, tmp, w, h, t, ksize, 1);
  hipLaunchKernelGGL(k_gauss_pass, grid2(w, h), dim3(128), 0, a->stream, tmp, dst, w, h, t, ksize, 0);
  AK_HIP(hipGetLastError());
  return SFMLOC_OK;
}

int scharr(Akaze *a, const float *src, float *dst, int w, int h, int xorder, int scale, float ws, float wm) {
  hipLaunchKernelGGL(k_scharr, grid2(w, h), dim3(128), 0, a->stream, src, dst, w, h, xorder, scale, ws, wm);
  AK_HIP(hipGetLastError());
  return SFMLOC_OK;
}

LevelTab level_tab(const Akaze *a) {
  const AkPlan &P = a->plan;
  LevelTab T;
  memset(&T, 0, sizeof(T));
  T.n = P.nlev;
  int row = 0;
  for (int i = 0; i < P.nlev; ++i) {
    const AkLevel &L = P.lev[i];
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (float)L.sigma_size * (wgt + 2.0f));
    T.row0[i] = row;
    T.w[i] = L.w;
    T.h[i] = L.h;
    T.sc[i] = L.sigma_size;
    T.off[i] = (unsigned int)L.off;
    T.ws[i] = norm;
    T.wm[i] = wgt * norm;
    T.sigma_size[i] = fround_h(L.esigma * 1.5f / (float)(1 << L.octave));  // the extrema pass's border test
    row += L.h;
  }
  T.row0[P.nlev] = row;
  return T;
}

// Create_Nonlinear_Scale_Space + Compute_Multiscale_Derivatives + Compute_Determinant_Hessian_Response
int build_scale_space(Akaze *a, const uint8_t *gray /*host; null: the image is already in d_gray, written on a->stream*/) {
  const AkPlan &P = a->plan;
  const int w = a->w, h = a->h;
  const size_t n0 = (size_t)w * h;
  // (nothing recorded for this image touches d_gray before the upload: it need not wait for a gang session's launches)
  if (gray) AK_HIP(hipMemcpyAsync(a->d_gray, gray, n0, hipMemcpyHostToDevice, a->stream.unordered()));
#define s ((hipStream_t)a->stream) /* the few launches below that have no gang form (comparison paths) */
  int rc = SFMLOC_OK;
  static const bool kFusedPre = [] {  // SFMLOC_AKAZE_FUSED_PRE=0: the thirteen separate launches (comparison runs)
    const char *e = getenv("SFMLOC_AKAZE_FUSED_PRE");
    return !(e && atoi(e) == 0);
  }();
  const dim3 ggrid((w + 127) / 128, (h + kGradRows - 1) / kGradRows);
  if (kFusedPre) {
    Taps2 t2;
    for (int k = 0; k < 9; ++k) t2.k9[k] = P.g16[k];
    for (int k = 0; k < 5; ++k) t2.k5[k] = P.g10[k];
    sfm_launch<PreRowsBody>(a, k_pre_rows, grid2(w, h), dim3(128), (uint32_t)0, a->d_gray, a->d_t3, a->d_t1, w, h, t2, a->d_hist);
    sfm_launch<PreColsBody>(a, k_pre_cols, grid2(w, h), dim3(128), (uint32_t)0, a->d_t3, a->d_t1, a->d_Lt, a->d_Lsmooth, a->d_t0, w, h, t2);
    sfm_launch<PreGradBody>(a, k_pre_grad, ggrid, dim3(128), (uint32_t)0, a->d_t0, a->d_t2, w, h, a->d_hist);
    sfm_launch<PreHistBody>(a, k_pre_hist, ggrid, dim3(128), (uint32_t)0, a->d_t2, w, h, a->d_hist, a->d_kcontrast);
    AK_HIP(hipGetLastError());
  } else {
  hipLaunchKernelGGL(k_u8_to_f32, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, s, a->d_gray, a->d_img, n0);
  rc = gauss(a, a->d_img, a->d_Lt, a->d_t3, w, h, P.g16, 9);
  if (rc) return rc;
  AK_HIP(hipMemcpyAsync(a->d_Lsmooth, a->d_Lt, n0 * sizeof(float), hipMemcpyDeviceToDevice, s));
  // contrast factor
  rc = gauss(a, a->d_img, a->d_t0, a->d_t3, w, h, P.g10, 5);
  if (!rc) rc = scharr(a, a->d_t0, a->d_t1, w, h, 1, 1, 3.0f, 10.0f);
  if (!rc) rc = scharr(a, a->d_t0, a->d_t2, w, h, 0, 1, 3.0f, 10.0f);
  if (rc) return rc;
  AK_HIP(hipMemsetAsync(a->d_hist, 0, 302 * sizeof(unsigned int), s));
  hipLaunchKernelGGL(k_grad_max, ggrid, dim3(128), 0, s, a->d_t1, a->d_t2, w, h, a->d_hist);
  hipLaunchKernelGGL(k_grad_hist, ggrid, dim3(128), 0, s, a->d_t1, a->d_t2, w, h, a->d_hist, a->d_hist + 1);
  hipLaunchKernelGGL(k_kcontrast, dim3(1), dim3(1), 0, s, a->d_hist, a->d_hist + 1, a->d_kcontrast);
  AK_HIP(hipGetLastError());
  }
  static const bool kResident = [] {  // SFMLOC_AKAZE_RESIDENT=0: the per-level kernels everywhere (comparison runs)
    const char *e = getenv("SFMLOC_AKAZE_RESIDENT");
    return !(e && atoi(e) == 0);
  }();
  for (int i = 1; i < P.nlev; ++i) {
    const AkLevel &L = P.lev[i], &Lp = P.lev[i - 1];
    float *Lt = a->d_Lt + L.off;
    // an octave whose image fits in LDS three times runs in one launch (k_octave_resident)
    {
      int j = i;
      while (j + 1 < P.nlev && P.lev[j + 1].octave == L.octave) ++j;
      const size_t lds = (size_t)3 * L.w * L.h * sizeof(float);
      const bool whole = i == 1 || Lp.octave != L.octave;  // the octave's first level of the loop
      if (kResident && whole && lds <= 150u * 1024u && j - i + 1 <= kOctaveRunMax && a->d_half_steps) {
        const float *start = a->d_Lt + Lp.off;
        if (L.octave > Lp.octave) {
          sfm_launch<HalfsampleBody>(a, k_halfsample, grid2(L.w, L.h), dim3(128), (uint32_t)0, a->d_Lt + Lp.off, Lp.w, Lp.h, a->d_t3, L.w, L.h);
          start = a->d_t3;
        }
        OctaveRun R{};
        R.n_levels = j - i + 1;
        R.w = L.w;
        R.h = L.h;
        R.octave = L.octave;
        for (int q = i; q <= j; ++q) {
          R.off[q - i] = (unsigned int)P.lev[q].off;
          R.nsteps[q - i] = P.lev[q].nsteps;
          R.step0[q - i] = q * 64;
        }
        Taps t5;
        for (int k = 0; k < 9; ++k) t5.k[k] = k < 5 ? P.g10[k] : 0.0f;
        // (a function-local static with an initialiser: set once, safely, whichever host thread gets here first)
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(k_octave_resident),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        AK_HIP(attr);
        sfm_launch<OctaveResidentBody>(a, k_octave_resident, dim3(1), dim3(1024), (uint32_t)lds, start, a->d_Lt, a->d_Lsmooth, t5,
                           a->d_kcontrast, a->d_half_steps, R);
        AK_HIP(hipGetLastError());
        i = j;
        continue;
      }
    }
    // the level starts from the previous level's Lt (half-sampled at an octave change); the FED steps ping-pong
    // between this level's Lt and a scratch image, arranged so that the last step lands in Lt without a copy
    const float *start = a->d_Lt + Lp.off;
    // diffusion steps per launch (k_nld_steps): 4 on the large levels, where the tile halo's redundant work costs more
    // than a launch; more on the upper octaves (<= 160 x 120), whose steps are each far shorter than a launch boundary and
    // which hold most of the schedule's steps (133 of 165 at VGA).  SFMLOC_AKAZE_FUSE / _FUSE_SMALL = 1..16 for tuning.
    static const int kFuseBig = [] {
      const char *e = getenv("SFMLOC_AKAZE_FUSE");
      const int v = e ? atoi(e) : 4;
      return (v >= 1 && v <= kNldFuseMax) ? v : 4;
    }();
    static const int kFuseSmall = [] {
      const char *e = getenv("SFMLOC_AKAZE_FUSE_SMALL");
      const int v = e ? atoi(e) : 8;
      return (v >= 1 && v <= kNldFuseMax) ? v : 8;
    }();
    const int kFuse = (L.w <= 160) ? kFuseSmall : kFuseBig;
    const int n_launch = (L.nsteps + kFuse - 1) / kFuse;
    if (L.octave > Lp.octave) {
      float *half = (n_launch % 2 == 0) ? Lt : a->d_t3;  // even number of launches: start (and end) in Lt
      sfm_launch<HalfsampleBody>(a, k_halfsample, grid2(L.w, L.h), dim3(128), (uint32_t)0, a->d_Lt + Lp.off, Lp.w, Lp.h, half, L.w, L.h);
      start = half;
    }
    Taps t5;
    for (int k = 0; k < 9; ++k) t5.k[k] = k < 5 ? P.g10[k] : 0.0f;
    static const bool kFuseSmooth = [] {  // SFMLOC_AKAZE_FUSE_SMOOTH=0: k_smooth_flow as a launch of its own (comparison runs)
      const char *e = getenv("SFMLOC_AKAZE_FUSE_SMOOTH");
      return !(e && atoi(e) == 0);
    }();
    const bool fuse_smooth = kFuseSmooth && L.nsteps > 0;
    if (!fuse_smooth)
      sfm_launch<SmoothFlowBody>(a, k_smooth_flow, dim3((L.w + 31) / 32, (L.h + 7) / 8), dim3(256), (uint32_t)0, start,
                         a->d_Lsmooth + L.off, a->d_t2, L.w, L.h, t5, a->d_kcontrast, L.octave);
    const float *cur = start;
    // destination of step st: alternate so that step nsteps-1 writes Lt; `start` is never written
    const dim3 tgrid((L.w + kNldTileX - 1) / kNldTileX, (L.h + kNldTileY - 1) / kNldTileY);
    for (int j = 0, st = 0; j < n_launch; ++j) {
      float *dst = ((n_launch - 1 - j) % 2 == 0) ? Lt : a->d_t3;
      if (dst == cur) dst = (dst == Lt) ? a->d_t3 : Lt;  // cannot happen by construction; guards an in-place step
      const int n = std::min(kFuse, L.nsteps - st);
      NldSteps hs;
      for (int k = 0; k < kNldFuseMax; ++k) hs.half_step[k] = k < n ? 0.5f * L.tsteps[st + k] : 0.0f;
      if (j == 0 && fuse_smooth) {  // the level's smoothing + conductivity + first steps in one launch
#define SNLD_CASE(KK)                                                                                                  \
  case KK:                                                                                                             \
    sfm_launch<SmoothNldBody<KK>>(a, k_smooth_nld<KK>, tgrid, dim3(256), 0, cur, a->d_Lsmooth + L.off, a->d_t2, dst, L.w, L.h, t5, \
                       a->d_kcontrast, L.octave, hs, n);                                                               \
    break;
        switch (n) {
          SNLD_CASE(1) SNLD_CASE(2) SNLD_CASE(3) SNLD_CASE(4) SNLD_CASE(5) SNLD_CASE(6) SNLD_CASE(7) SNLD_CASE(8)
          SNLD_CASE(9) SNLD_CASE(10) SNLD_CASE(11) SNLD_CASE(12) SNLD_CASE(13) SNLD_CASE(14) SNLD_CASE(15) SNLD_CASE(16)
        }
#undef SNLD_CASE
        st += n;
        cur = dst;
        continue;
      }
#define NLD_CASE(KK) \
  case KK: sfm_launch<NldStepsBody<KK>>(a, k_nld_steps<KK>, tgrid, dim3(256), 0, cur, a->d_t2, dst, L.w, L.h, hs, n); break;
      switch (n) {
        NLD_CASE(1) NLD_CASE(2) NLD_CASE(3) NLD_CASE(4) NLD_CASE(5) NLD_CASE(6) NLD_CASE(7) NLD_CASE(8)
        NLD_CASE(9) NLD_CASE(10) NLD_CASE(11) NLD_CASE(12) NLD_CASE(13) NLD_CASE(14) NLD_CASE(15) NLD_CASE(16)
      }
#undef NLD_CASE
      st += n;
      cur = dst;
    }
    AK_HIP(hipGetLastError());
    if (L.nsteps == 0 && start != Lt)
      AK_HIP(hipMemcpyAsync(Lt, start, (size_t)L.w * L.h * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  const LevelTab T = level_tab(a);
  const dim3 agrid((a->w + 127) / 128, T.row0[T.n]);
  sfm_launch<ScharrXyAllBody>(a, k_scharr_xy_all, agrid, dim3(128), (uint32_t)0, a->d_Lsmooth, a->d_Lx, a->d_Ly, reinterpret_cast<const LevelTab *>(a->d_level_tab));
  sfm_launch<HessianDetAllBody>(a, k_hessian_det_all, agrid, dim3(128), (uint32_t)0, a->d_Lx, a->d_Ly, a->d_Ldet, reinterpret_cast<const LevelTab *>(a->d_level_tab));
  AK_HIP(hipGetLastError());
  return SFMLOC_OK;
#undef s
}

struct HostKpt {
  float x, y, size, angle, response;
  int octave, class_id;
  float patch[9];
};

// the per-level image pointers and sizes k_orient_describe works on (constant for an extractor: uploaded once)
DevLevels dev_levels(const Akaze *a) {
  DevLevels LV;
  memset(&LV, 0, sizeof(LV));
  for (int i = 0; i < a->plan.nlev; ++i) {
    const AkLevel &L = a->plan.lev[i];
    LV.l[i].Lt = a->d_Lt + L.off;
    LV.l[i].Lx = a->d_Lx + L.off;
    LV.l[i].Ly = a->d_Ly + L.off;
    LV.l[i].w = L.w;
    LV.l[i].h = L.h;
    LV.l[i].octave = L.octave;
    LV.l[i].sf = (float)L.sigma_size;
  }
  return LV;
}

int ensure_kp_cap(Akaze *a, unsigned int n) {
  if (n <= a->kp_cap) return SFMLOC_OK;
  if (a->d_kp) hipFree(a->d_kp);
  if (a->d_angle) hipFree(a->d_angle);
  if (a->d_desc) hipFree(a->d_desc);
  a->d_kp = a->d_angle = nullptr;
  a->d_desc = nullptr;
  const unsigned int cap = std::max(n, 4096u);
  AK_HIP(hipMalloc((void **)&a->d_kp, (size_t)cap * 4 * sizeof(float)));
  AK_HIP(hipMalloc((void **)&a->d_angle, (size_t)cap * sizeof(float)));
  AK_HIP(hipMalloc((void **)&a->d_desc, (size_t)cap * 64));
  a->kp_cap = cap;
  return SFMLOC_OK;
}

// phase 0: upload + kernel + downloads; 1: upload + kernel only; 2: the downloads only (a batch queues every image's kernel
// before the first download, which blocks the host when the destination is pageable memory)
int orient_describe_enqueue(Akaze *a, const std::vector<float> &kin, unsigned int n, float *angle_out, uint8_t *desc64,
                            int phase, unsigned int grid_n = 0 /*workgroups to launch (>= n; a batch launches its largest)*/) {
  if (n == 0 && grid_n == 0) return SFMLOC_OK;
  if (grid_n < n) grid_n = n;
  int rc = ensure_kp_cap(a, n);
  if (rc) return rc;
  if (phase != 2) {
    // (the keypoints go up ahead of anything recorded: nothing recorded for this extractor in the session reads d_kp)
    if (n) AK_HIP(hipMemcpyAsync(a->d_kp, kin.data(), (size_t)n * 4 * sizeof(float), hipMemcpyHostToDevice, a->stream.unordered()));
    sfm_launch<OrientDescribeBody>(a, k_orient_describe, dim3(grid_n), dim3(64), 0,
                                   reinterpret_cast<const DevLevels *>(a->d_dev_levels), a->d_kp, (int)n, a->d_gauss25, a->d_win,
                                   a->plan.n_win, a->d_pair, a->d_angle, a->d_desc);
    AK_HIP(hipGetLastError());
  }
  if (phase != 1 && n) {
    if (angle_out) AK_HIP(hipMemcpyAsync(angle_out, a->d_angle, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, a->stream));
    if (desc64) AK_HIP(hipMemcpyAsync(desc64, a->d_desc, (size_t)n * 64, hipMemcpyDeviceToHost, a->stream));
  }
  return SFMLOC_OK;
}

int orient_describe(Akaze *a, const std::vector<float> &kin, unsigned int n, float *angle_out, uint8_t *desc64) {
  int rc = orient_describe_enqueue(a, kin, n, angle_out, desc64, 0);
  if (rc) return rc;
  AK_HIP(hipStreamSynchronize(a->stream));
  return SFMLOC_OK;
}

}  // namespace

// for callers inside the library (imgbow.hip): see sfmloc_internal.h
uint8_t *akaze_gray_dev(Akaze *a) { return a->d_gray; }
uint8_t *akaze_desc_dev(Akaze *a) { return a->d_desc; }
hipStream_t akaze_stream_now(Akaze *a) { return a->stream; }
int akaze_compute_resident(Akaze *a, const float *d_kin, unsigned int n) {
  int rc = ensure_kp_cap(a, n);  // (before anything is queued: it may free and allocate)
  if (rc) return rc;
  rc = build_scale_space(a, nullptr);
  if (rc || n == 0) return rc;
  sfm_launch<OrientDescribeBody>(a, k_orient_describe, dim3(n), dim3(64), 0,
                                 reinterpret_cast<const DevLevels *>(a->d_dev_levels), d_kin, (int)n, a->d_gauss25, a->d_win,
                                 a->plan.n_win, a->d_pair, a->d_angle, a->d_desc);
  AK_HIP(hipGetLastError());
  return SFMLOC_OK;
}

}  // namespace sfmloc

using namespace sfmloc;

extern "C" {

void sfmloc_akaze_destroy(sfmloc_akaze *ak) {
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  if (!a) return;
  hipSetDevice(a->device);
  if (a->stream.own) hipStreamSynchronize(a->stream.own);
  gang_member_free(a);
  void *ptrs[] = {a->d_gray, a->d_img, a->d_t0, a->d_t1, a->d_t2, a->d_t3, a->d_Lt, a->d_Lsmooth, a->d_Lx, a->d_Ly,
                  a->d_Ldet, a->d_hist, a->d_kcontrast, a->d_half_steps, a->d_level_tab, a->d_dev_levels, a->d_cand, a->d_ncand,
                  a->d_gauss25, a->d_win, a->d_pair, a->d_kp, a->d_angle, a->d_desc};
  for (void *p : ptrs)
    if (p) hipFree(p);
  if (a->own_stream) hipStreamDestroy(a->own_stream);
  delete a;
}

int sfmloc_akaze_create(int device, int width, int height, int n_octaves, int n_sublevels, float threshold,
                        sfmloc_akaze **out) {
  SFM_CHECK(out, SFMLOC_EINVAL, "sfmloc_akaze_create: null argument");
  *out = nullptr;
  SFM_CHECK(width >= 16 && height >= 16 && width <= 16384 && height <= 16384, SFMLOC_EINVAL,
            "sfmloc_akaze_create: image size %dx%d", width, height);
  SFM_CHECK(n_octaves >= 1 && n_octaves <= 8 && n_sublevels >= 1 && n_octaves * n_sublevels <= kMaxLevels,
            SFMLOC_EINVAL, "sfmloc_akaze_create: nOct %d nOctLay %d", n_octaves, n_sublevels);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  SFM_CHECK(e == hipSuccess && ndev > 0, SFMLOC_ENODEV, "no HIP device visible; this library has no CPU fallback");
  SFM_HIP(hipSetDevice(device));
  Akaze *a = new (std::nothrow) Akaze();
  SFM_CHECK(a, SFMLOC_ENOMEM, "out of host memory");
  a->device = device;
  a->w = width;
  a->h = height;
  a->omax = n_octaves;
  a->nsub = n_sublevels;
  a->thres = threshold;
  make_plan(width, height, n_octaves, n_sublevels, a->plan);
  const size_t n0 = (size_t)width * height, tot = a->plan.total;
  hipError_t he = hipStreamCreateWithFlags(&a->own_stream, hipStreamNonBlocking);
  a->stream.own = a->own_stream;
  auto A = [&](void **p, size_t bytes) {
    if (he == hipSuccess) he = hipMalloc(p, bytes);
  };
  A((void **)&a->d_gray, n0);
  A((void **)&a->d_img, n0 * 4);
  A((void **)&a->d_t0, n0 * 4);
  A((void **)&a->d_t1, n0 * 4);
  A((void **)&a->d_t2, n0 * 4);
  A((void **)&a->d_t3, n0 * 4);
  float **stacks[] = {&a->d_Lt, &a->d_Lsmooth, &a->d_Lx, &a->d_Ly, &a->d_Ldet};
  for (float **sp : stacks) A((void **)sp, tot * 4);
  A((void **)&a->d_hist, 304 * 4);
  A((void **)&a->d_kcontrast, 4);
  A((void **)&a->d_half_steps, (size_t)kMaxLevels * 64 * 4);
  A((void **)&a->d_level_tab, sizeof(LevelTab));
  A((void **)&a->d_dev_levels, sizeof(DevLevels));
  A((void **)&a->d_cand, (size_t)a->cand_cap * sizeof(Candidate9));
  A((void **)&a->d_ncand, 4);
  A((void **)&a->d_gauss25, 49 * 4);
  A((void **)&a->d_win, 64 * 4);
  A((void **)&a->d_pair, 486 * 2 * 2);
  if (he == hipSuccess) {
    std::vector<float> hs((size_t)kMaxLevels * 64, 0.0f);
    for (int i = 0; i < a->plan.nlev; ++i)
      for (int k = 0; k < a->plan.lev[i].nsteps && k < 64; ++k) hs[(size_t)i * 64 + k] = 0.5f * a->plan.lev[i].tsteps[k];
    he = hipMemcpy(a->d_half_steps, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  }
  if (he == hipSuccess) {
    const LevelTab T = level_tab(a);
    he = hipMemcpy(a->d_level_tab, &T, sizeof(T), hipMemcpyHostToDevice);
  }
  if (he == hipSuccess) {
    const DevLevels LV = dev_levels(a);
    he = hipMemcpy(a->d_dev_levels, &LV, sizeof(LV), hipMemcpyHostToDevice);
  }
  if (he == hipSuccess) he = hipMemcpy(a->d_gauss25, a->plan.gauss25, 49 * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(a->d_win, a->plan.win_ang1, 64 * 4, hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipMemcpy(a->d_pair, a->plan.pair_tab, 486 * 2 * 2, hipMemcpyHostToDevice);
  if (he != hipSuccess) {
    set_error("sfmloc_akaze_create: %s", hipGetErrorString(he));
    sfmloc_akaze_destroy(reinterpret_cast<sfmloc_akaze *>(a));
    return he == hipErrorOutOfMemory ? SFMLOC_ENOMEM : SFMLOC_EHIP;
  }
  *out = reinterpret_cast<sfmloc_akaze *>(a);
  return SFMLOC_OK;
}

int sfmloc_akaze_share_stream(sfmloc_akaze *ak, sfmloc_context *ctx) {
  SFM_CHECK(ak, SFMLOC_EINVAL, "sfmloc_akaze_share_stream: null extractor");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_CHECK(!c || c->map->device == a->device, SFMLOC_EINVAL, "sfmloc_akaze_share_stream: extractor and context on different devices");
  hipSetDevice(a->device);
  if (a->stream) hipStreamSynchronize(a->stream);
  a->stream.own = c ? c->stream.own : a->own_stream;
  return SFMLOC_OK;
}

int sfmloc_akaze_levels(const sfmloc_akaze *ak, int *n_levels, int *wh /*[32*2]*/) {
  SFM_CHECK(ak && n_levels, SFMLOC_EINVAL, "sfmloc_akaze_levels: null argument");
  const Akaze *a = reinterpret_cast<const Akaze *>(ak);
  *n_levels = a->plan.nlev;
  if (wh)
    for (int i = 0; i < a->plan.nlev; ++i) {
      wh[2 * i] = a->plan.lev[i].w;
      wh[2 * i + 1] = a->plan.lev[i].h;
    }
  return SFMLOC_OK;
}

int sfmloc_akaze_read_levels(sfmloc_akaze *ak, float *ldet, float *lt) {
  SFM_CHECK(ak, SFMLOC_EINVAL, "sfmloc_akaze_read_levels: null argument");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  SFM_HIP(hipSetDevice(a->device));
  SFM_HIP(hipStreamSynchronize(a->stream));
  if (ldet) SFM_HIP(hipMemcpy(ldet, a->d_Ldet, a->plan.total * sizeof(float), hipMemcpyDeviceToHost));
  if (lt) SFM_HIP(hipMemcpy(lt, a->d_Lt, a->plan.total * sizeof(float), hipMemcpyDeviceToHost));
  return SFMLOC_OK;
}

// The device side of detection up to the extrema candidates of every level: launches only (an extractor that records for
// a gang session takes part in ONE launch per kernel with the other images of the batch), no host synchronisation.
static int akaze_detect_enqueue(Akaze *a, const uint8_t *gray) {
  // (the counter is not touched by anything the scale space launches: it need not wait for recorded launches)
  SFM_HIP(hipMemsetAsync(a->d_ncand, 0, sizeof(unsigned int), a->stream.unordered()));
  int rc = build_scale_space(a, gray);
  if (rc) return rc;
  const LevelTab T = level_tab(a);
  sfm_launch<ExtremaAllBody>(a, k_extrema_all, dim3((a->w + 127) / 128, T.row0[T.n]), dim3(128), 0, a->d_Ldet, reinterpret_cast<const LevelTab *>(a->d_level_tab), a->thres,
                             a->d_cand, a->cand_cap, a->d_ncand);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

static int akaze_detect_finish(Akaze *a, float *kpts, uint8_t *desc64, uint32_t cap, uint32_t *n_out, double t_0);

int sfmloc_akaze_detect_and_compute(sfmloc_akaze *ak, const uint8_t *gray, float *kpts /*[cap*6]*/,
                                    uint8_t *desc64 /*[cap*64]*/, uint32_t cap, uint32_t *n_out) {
  SFM_CHECK(ak && gray && n_out, SFMLOC_EINVAL, "sfmloc_akaze_detect_and_compute: null argument");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  SFM_HIP(hipSetDevice(a->device));
  static const bool timing = getenv("SFMLOC_AKAZE_TIMING") != nullptr;
  const double t_0 = timing ? now_s() : 0.0;
  int rc = akaze_detect_enqueue(a, gray);
  if (rc) return rc;
  return akaze_detect_finish(a, kpts, desc64, cap, n_out, t_0);
}

// One image's detection after the device has produced the extrema candidates, in stages so that a batch of images can
// take each stage together (one host synchronisation per stage instead of three per image, the host's refinement of the
// images in parallel): count -> candidates -> host refinement -> orientation + descriptors -> outputs.
struct DetectState {
  unsigned int nc = 0;
  std::vector<Candidate9> cand;
  std::vector<HostKpt> fin;
  std::vector<float> kin, ang;
  uint32_t n = 0;
};

// (a) the candidate count, asynchronously on stream s; the caller synchronises
static int detect_count_enqueue(Akaze *a, DetectState &st, hipStream_t s) {
  SFM_HIP(hipMemcpyAsync(&st.nc, a->d_ncand, sizeof(st.nc), hipMemcpyDeviceToHost, s));
  return SFMLOC_OK;
}
// (b) the candidates themselves
static int detect_candidates_read(Akaze *a, DetectState &st) {
  SFM_CHECK(st.nc <= a->cand_cap, SFMLOC_ECAP, "AKAZE: %u extrema candidates exceed the workspace (%u)", st.nc, a->cand_cap);
  st.cand.resize(st.nc);
  if (st.nc) SFM_HIP(hipMemcpy(st.cand.data(), a->d_cand, (size_t)st.nc * sizeof(Candidate9), hipMemcpyDeviceToHost));
  return SFMLOC_OK;
}
// (c) host only (no HIP call, no shared state: images can take it in parallel threads)
static void detect_refine_host(const Akaze *a, DetectState &st) {
  const AkPlan &P = a->plan;
  std::vector<Candidate9> &cand = st.cand;
  const unsigned int nc = st.nc;
  std::vector<HostKpt> &fin = st.fin;
  std::sort(cand.begin(), cand.end(), [](const Candidate9 &p, const Candidate9 &q) {
    if (p.level != q.level) return p.level < q.level;
    if (p.y != q.y) return p.y < q.y;
    return p.x < q.x;
  });
  // OpenCV's sequential duplicate suppression (same / previous level while scanning, then against the upper level).
  // The reference scans every accepted point for every candidate (quadratic: 2.1 ms for a 1080p frame); a uniform grid
  // over the accepted points finds the same one -- "the first accepted point, in acceptance order, of this or the
  // previous level within the candidate's radius" is the minimum index among the grid hits -- in 0.1 ms.
  std::vector<HostKpt> aux;
  aux.reserve(nc);
  constexpr float kCell = 32.0f;
  const int gw = (int)(a->w / kCell) + 2, gh = (int)(a->h / kCell) + 2;
  std::vector<std::vector<uint32_t>> grid((size_t)gw * gh);
  auto cell_of = [&](float v, int n) {
    int c = (int)floorf(v / kCell);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
  };
  for (const Candidate9 &c : cand) {
    const AkLevel &L = P.lev[c.level];
    HostKpt pt;
    pt.response = fabsf(c.patch[4]);
    pt.size = L.esigma * 1.5f;
    pt.octave = L.octave;
    pt.class_id = c.level;
    pt.angle = 0.0f;
    memcpy(pt.patch, c.patch, sizeof(pt.patch));
    const float ratio = (float)(1 << L.octave);
    pt.x = (float)c.x;
    pt.y = (float)c.y;
    bool is_extremum = true, is_repeated = false;
    size_t id_repeated = 0;
    {
      const float px = pt.x * ratio, py = pt.y * ratio;
      const int cx0 = cell_of(px - pt.size, gw), cx1 = cell_of(px + pt.size, gw);
      const int cy0 = cell_of(py - pt.size, gh), cy1 = cell_of(py + pt.size, gh);
      size_t first = (size_t)-1;
      for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx)
          for (uint32_t ik : grid[(size_t)cy * gw + cx]) {
            if (ik >= first) continue;
            if (pt.class_id - 1 == aux[ik].class_id || pt.class_id == aux[ik].class_id) {
              const float dx = pt.x * ratio - aux[ik].x, dy = pt.y * ratio - aux[ik].y;
              const float dist = dx * dx + dy * dy;
              if (dist <= pt.size * pt.size) first = ik;
            }
          }
      if (first != (size_t)-1) {
        if (pt.response > aux[first].response) {
          id_repeated = first;
          is_repeated = true;
        } else {
          is_extremum = false;
        }
      }
    }
    if (!is_extremum) continue;
    pt.x = pt.x * ratio;
    pt.y = pt.y * ratio;
    const size_t dst_cell = (size_t)cell_of(pt.y, gh) * gw + cell_of(pt.x, gw);
    if (!is_repeated) {
      grid[dst_cell].push_back((uint32_t)aux.size());
      aux.push_back(pt);
    } else {
      std::vector<uint32_t> &old = grid[(size_t)cell_of(aux[id_repeated].y, gh) * gw + cell_of(aux[id_repeated].x, gw)];
      old.erase(std::find(old.begin(), old.end(), (uint32_t)id_repeated));
      grid[dst_cell].push_back((uint32_t)id_repeated);
      aux[id_repeated] = pt;
    }
  }
  std::vector<HostKpt> kept;
  kept.reserve(aux.size());
  for (size_t i = 0; i < aux.size(); ++i) {
    bool rep = false;
    const int cx0 = cell_of(aux[i].x - aux[i].size, gw), cx1 = cell_of(aux[i].x + aux[i].size, gw);
    const int cy0 = cell_of(aux[i].y - aux[i].size, gh), cy1 = cell_of(aux[i].y + aux[i].size, gh);
    for (int cy = cy0; cy <= cy1 && !rep; ++cy)
      for (int cx = cx0; cx <= cx1 && !rep; ++cx)
        for (uint32_t j : grid[(size_t)cy * gw + cx])
          if (j > i && aux[i].class_id + 1 == aux[j].class_id) {
            const float dx = aux[i].x - aux[j].x, dy = aux[i].y - aux[j].y;
            if (dx * dx + dy * dy <= aux[i].size * aux[i].size && aux[i].response < aux[j].response) {
              rep = true;
              break;
            }
          }
    if (!rep) kept.push_back(aux[i]);
  }
  // Do_Subpixel_Refinement on the carried 3x3 patch
  fin.clear();
  fin.reserve(kept.size());
  for (HostKpt k : kept) {
    const float ratio = (float)(1 << k.octave);
    const int x = fround_h(k.x / ratio), y = fround_h(k.y / ratio);
    const float *p = k.patch;  // p[(dy+1)*3 + dx+1]
    const float Dx = 0.5f * (p[5] - p[3]);
    const float Dy = 0.5f * (p[7] - p[1]);
    const float Dxx = (p[5] + p[3]) - 2.0f * p[4];
    const float Dyy = (p[7] + p[1]) - 2.0f * p[4];
    const float Dxy = 0.25f * (p[8] + p[0]) - 0.25f * (p[2] + p[6]);
    const float det = Dxx * Dyy - Dxy * Dxy;
    if (det == 0.0f) continue;
    const float d0 = (-Dx * Dyy + Dy * Dxy) / det;
    const float d1 = (-Dy * Dxx + Dx * Dxy) / det;
    if (fabsf(d0) <= 1.0f && fabsf(d1) <= 1.0f) {
      k.x = ((float)x + d0) * ratio;
      k.y = ((float)y + d1) * ratio;
      k.size = k.size * 2.0f;
      fin.push_back(k);
    }
  }
  const uint32_t n = (uint32_t)fin.size();
  st.n = n;
  st.kin.resize((size_t)n * 4);
  st.ang.assign(n, 0.0f);
  for (uint32_t i = 0; i < n; ++i) {
    st.kin[4 * i] = fin[i].x;
    st.kin[4 * i + 1] = fin[i].y;
    st.kin[4 * i + 2] = fin[i].size;
    st.kin[4 * i + 3] = (float)fin[i].class_id;
  }
}
// (e) the keypoint records
static void detect_outputs(const DetectState &st, float *kpts) {
  if (!kpts) return;
  for (uint32_t i = 0; i < st.n; ++i) {
    kpts[6 * i] = st.fin[i].x;
    kpts[6 * i + 1] = st.fin[i].y;
    kpts[6 * i + 2] = st.fin[i].size;
    kpts[6 * i + 3] = st.ang[i];
    kpts[6 * i + 4] = st.fin[i].response;
    kpts[6 * i + 5] = (float)st.fin[i].class_id;
  }
}

static int akaze_detect_finish(Akaze *a, float *kpts, uint8_t *desc64, uint32_t cap, uint32_t *n_out, double t_0) {
  static const bool timing = getenv("SFMLOC_AKAZE_TIMING") != nullptr;
  DetectState st;
  int rc = detect_count_enqueue(a, st, a->stream);
  if (rc) return rc;
  SFM_HIP(hipStreamSynchronize(a->stream));
  rc = detect_candidates_read(a, st);
  if (rc) return rc;
  const double t_1 = timing ? now_s() : 0.0;
  detect_refine_host(a, st);
  *n_out = st.n;
  SFM_CHECK(st.n <= cap, SFMLOC_ECAP, "AKAZE: %u keypoints, caller buffers hold %u", st.n, cap);
  const double t_2 = timing ? now_s() : 0.0;
  rc = orient_describe_enqueue(a, st.kin, st.n, st.ang.data(), desc64, 0);
  if (rc) return rc;
  SFM_HIP(hipStreamSynchronize(a->stream));
  if (timing)
    fprintf(stderr, "akaze %dx%d: scale space + extrema %.3f ms, host suppression + subpixel %.3f ms (%u candidates -> %u), "
                    "orientation + M-LDB %.3f ms\n", a->w, a->h, (t_1 - t_0) * 1e3, (t_2 - t_1) * 1e3, st.nc, st.n,
            (now_s() - t_2) * 1e3);
  detect_outputs(st, kpts);
  return SFMLOC_OK;
}

// Several images of one size at once: extractor i takes image i; the scale spaces, determinants and extrema of all of them
// go out as ONE launch per kernel (a gang session on the first extractor's stream); the candidates' refinement on the
// host and the orientation + M-LDB launch follow per image.  Results are those of n separate calls.
int sfmloc_akaze_detect_and_compute_batch(sfmloc_akaze *const *aks, const uint8_t *const *grays, uint32_t n,
                                          float *const *kpts, uint8_t *const *descs, uint32_t cap, uint32_t *n_out) {
  SFM_CHECK(aks && grays && kpts && n_out && n >= 1 && n <= (uint32_t)kGangMembers, SFMLOC_EINVAL,
            "sfmloc_akaze_detect_and_compute_batch: 1..%d images", kGangMembers);
  GangMember *ms[kGangMembers];
  Akaze *first = reinterpret_cast<Akaze *>(aks[0]);
  for (uint32_t i = 0; i < n; ++i) {
    Akaze *a = reinterpret_cast<Akaze *>(aks[i]);
    SFM_CHECK(a && grays[i] && kpts[i], SFMLOC_EINVAL, "sfmloc_akaze_detect_and_compute_batch: null argument (image %u)", i);
    SFM_CHECK(a->device == first->device && a->w == first->w && a->h == first->h, SFMLOC_EINVAL,
              "sfmloc_akaze_detect_and_compute_batch: the extractors differ in device or image size");
    SFM_CHECK(a->stream.gang == nullptr, SFMLOC_EINVAL, "sfmloc_akaze_detect_and_compute_batch: extractor %u is in a session", i);
    for (uint32_t j = 0; j < i; ++j) SFM_CHECK(aks[j] != aks[i], SFMLOC_EINVAL, "extractor listed twice");
    ms[i] = a;
  }
  SFM_HIP(hipSetDevice(first->device));
  static const bool timing = getenv("SFMLOC_AKAZE_TIMING") != nullptr;
  const double t_0 = timing ? now_s() : 0.0;
  int rc = n > 1 ? gang_open(ms, (int)n) : SFMLOC_OK;
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) rc = akaze_detect_enqueue(reinterpret_cast<Akaze *>(aks[i]), grays[i]);
  const int rc_close = gang_close(first);
  if (rc == SFMLOC_OK) rc = rc_close;
  if (rc) return rc;
  double t_s[6] = {t_0, 0, 0, 0, 0, 0};
  if (timing) t_s[1] = now_s();
  // every stage for all the images, one host synchronisation per stage
  std::vector<DetectState> st(n);
  hipStream_t s0 = first->stream;  // (the gang's stream: the candidates of every image were produced on it)
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) rc = detect_count_enqueue(reinterpret_cast<Akaze *>(aks[i]), st[i], s0);
  if (rc) return rc;
  SFM_HIP(hipStreamSynchronize(s0));
  if (timing) t_s[2] = now_s();
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) rc = detect_candidates_read(reinterpret_cast<Akaze *>(aks[i]), st[i]);
  if (rc) return rc;
  if (timing) t_s[3] = now_s();
  {  // the host's refinement: images are independent
    const unsigned nthr = n < 8 ? n : 8;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthr; ++t)
      th.emplace_back([&, t] {
        for (uint32_t i = t; i < n; i += nthr) detect_refine_host(reinterpret_cast<const Akaze *>(aks[i]), st[i]);
      });
    for (uint32_t i = 0; i < n; i += nthr) detect_refine_host(reinterpret_cast<const Akaze *>(aks[i]), st[i]);
    for (auto &t : th) t.join();
  }
  if (timing) t_s[4] = now_s();
  for (uint32_t i = 0; i < n; ++i) {
    n_out[i] = st[i].n;
    SFM_CHECK(st[i].n <= cap, SFMLOC_ECAP, "AKAZE: %u keypoints, caller buffers hold %u", st[i].n, cap);
  }
  // orientation + M-LDB of every frame in ONE launch (a second session; the launch is as large as the frame with most
  // keypoints needs, a workgroup beyond a frame's own count leaves at once), then the downloads
  unsigned int most = 0;
  for (uint32_t i = 0; i < n; ++i) most = st[i].n > most ? st[i].n : most;
  if (most) {
    rc = n > 1 ? gang_open(ms, (int)n) : SFMLOC_OK;
    for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i)
      rc = orient_describe_enqueue(reinterpret_cast<Akaze *>(aks[i]), st[i].kin, st[i].n, nullptr, nullptr, 1, most);
    const int rc2 = gang_close(first);
    if (rc == SFMLOC_OK) rc = rc2;
    for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i)
      rc = orient_describe_enqueue(reinterpret_cast<Akaze *>(aks[i]), st[i].kin, st[i].n, st[i].ang.data(),
                                   descs ? descs[i] : nullptr, 2);
  }
  if (rc) return rc;
  for (uint32_t i = 0; i < n; ++i) SFM_HIP(hipStreamSynchronize(reinterpret_cast<Akaze *>(aks[i])->stream));
  for (uint32_t i = 0; i < n; ++i) detect_outputs(st[i], kpts[i]);
  if (timing)
    fprintf(stderr, "akaze batch of %u (%dx%d): launches queued %.3f ms, scale spaces + extrema done %.3f, candidates read "
                    "%.3f, host refinement %.3f, orientation + M-LDB + outputs %.3f\n", n, first->w, first->h,
            (t_s[1] - t_s[0]) * 1e3, (t_s[2] - t_s[1]) * 1e3, (t_s[3] - t_s[2]) * 1e3, (t_s[4] - t_s[3]) * 1e3,
            (now_s() - t_s[4]) * 1e3);
  return SFMLOC_OK;
}


int sfmloc_akaze_compute(sfmloc_akaze *ak, const uint8_t *gray, const float *kin, uint32_t n, uint8_t *desc64,
                         float *angle_out) {
  SFM_CHECK(ak && gray && (n == 0 || (kin && desc64)), SFMLOC_EINVAL, "sfmloc_akaze_compute: null argument");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  SFM_HIP(hipSetDevice(a->device));
  int rc = build_scale_space(a, gray);
  if (rc) return rc;
  std::vector<float> k(kin, kin + (size_t)n * 4);
  for (uint32_t i = 0; i < n; ++i) {
    int lvl = (int)k[4 * i + 3];
    lvl = lvl < 0 ? 0 : (lvl >= a->plan.nlev ? a->plan.nlev - 1 : lvl);
    k[4 * i + 3] = (float)lvl;
  }
  return orient_describe(a, k, n, angle_out, desc64);
}

}  // extern "C"
