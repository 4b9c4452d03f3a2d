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
#include <vector>

#include "sfmloc_internal.h"
#include "geom_device.h"

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
    // (unrolled: a thread's rows are independent, and walked one after the other each row is a round trip to memory)
#pragma unroll 4
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
    // (only where it can still raise the maximum: a value read a moment ago is a lower bound of the current one)
    if ((threadIdx.x & 63) == 0 && bits > __hip_atomic_load(hmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(hmax_bits, bits);
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
    // (the rows' magnitudes first, side by side; then the bins)
    // (the launcher gives a workgroup as many chunks of kGradRows rows as keeps the grid near one workgroup per compute
    // unit: every workgroup ends in ~100 atomics on the same 301 words, and 1 020 workgroups' worth of them were most of
    // this kernel's 34 us at 1080p)
    const int chunks = ((h + kGradRows - 1) / kGradRows + (int)gridDim.y - 1) / (int)gridDim.y;
    for (int ch = 0; ch < chunks; ++ch) {
      const int y0 = ((int)blockIdx.y * chunks + ch) * kGradRows;
      if (y0 >= h) break;
      float mv[kGradRows];
#pragma unroll
      for (int r = 0; r < kGradRows; ++r) {
        const int y = y0 + r;
        mv[r] = (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) ? mag[(size_t)y * w + x] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < kGradRows; ++r) {
        const float m = mv[r];
        if (m != 0.0f) {
          int nbin = (int)floorf(300.0f * (m / hmax));
          if (nbin == 300) nbin--;
          atomicAdd(&lh[nbin], 1u);
          ++mine;
        }
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
    if (ticket != gridDim.x * gridDim.y - 1) return;
    __threadfence();
    // compute_k_percentile's tail (k_kcontrast), by the last workgroup; the histogram is read through atomics so that no
    // stale cached word is used.  (Round 3: all of the workgroup's threads fetch the bins, side by side, and one lane walks
    // them in LDS -- the walk used to be one lane's chain of up to 300 global atomic round trips, most of this kernel's
    // 24 us.  Same integers, same k.)
    for (int i = threadIdx.x; i < 301; i += blockDim.x) lh[i] = atomicAdd(&hist[i], 0u);
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int npoints = (int)lh[300];
    const int nthreshold = (int)((float)npoints * 0.7f);
    int nelements = 0, k = 0;
    for (k = 0; nelements < nthreshold && k < 300; k++) nelements += (int)lh[k];
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
      float *cur = A, *nxt = C;
      // Round 3: when the image's rows cut into runs of kRun pixels give at most 1 024 runs (the 80 x 60 octave of a VGA
      // image: 960), a thread owns ONE RUN: its pixels' values stay in registers from step to step, the neighbours
      // inside the run are registers too, and a step reads 2 + 2 kRun values from LDS instead of 5 per pixel, at addresses
      // fixed for the whole level (a missing neighbour's address is the pixel's own: the +0 difference again).  The
      // workgroup is one compute unit's issue slots -- 99 of the schedule's steps run here -- and this form has about
      // half the instructions per step; the arithmetic per pixel is the same expression in the same order.
      constexpr int kRun = 5;
      const int runs = (w + kRun - 1) / kRun;
      if (h * runs <= 1024) {
        const bool act = tid < h * runs;
        const int y = act ? tid / runs : 0, x0 = act ? (tid - y * runs) * kRun : 0;
        const int nx = act ? min(kRun, w - x0) : 0;
        const int p0 = y * w + x0;
        float cxp[kRun], cxn[kRun], cyp[kRun], cyn[kRun], v[kRun];
        int iU[kRun], iD[kRun];
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
          cxp[j] = cxn[j] = cyp[j] = cyn[j] = v[j] = 0.0f;
          iU[j] = iD[j] = p0;
          if (j < nx) {
            const int p = p0 + j, x = x0 + j;
            const float cc = B[p];
            if (x + 1 < w) cxp[j] = cc + B[p + 1];
            if (x > 0) cxn[j] = B[p - 1] + cc;
            if (y + 1 < h) cyp[j] = cc + B[p + w];
            if (y > 0) cyn[j] = B[p - w] + cc;
            iU[j] = y > 0 ? p - w : p;
            iD[j] = y + 1 < h ? p + w : p;
            v[j] = cur[p];
          }
        }
        const int iL = x0 > 0 ? p0 - 1 : p0;
        const int iR = x0 + nx < w ? p0 + nx : p0 + (nx > 0 ? nx - 1 : 0);
        for (int s = 0; s < R.nsteps[lv]; ++s) {
          const float half_step = half_steps[R.step0[lv] + s];
          if (act) {
            const float left = cur[iL], right = cur[iR];
            float up[kRun], dn[kRun];
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
              up[j] = cur[iU[j]];
              dn[j] = cur[iD[j]];
            }
            float nv[kRun];
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
              const float vj = v[j];
              const float vr = (j + 1 < kRun && j + 1 < nx) ? v[j + 1 < kRun ? j + 1 : j] : right;
              const float vl = j > 0 ? v[j > 0 ? j - 1 : 0] : left;
              const float xpos = cxp[j] * (vr - vj);
              const float xneg = cxn[j] * (vj - vl);
              const float ypos = cyp[j] * (dn[j] - vj);
              const float yneg = cyn[j] * (vj - up[j]);
              const float stp = half_step * (((xpos - xneg) + ypos) - yneg);
              nv[j] = vj + stp;
            }
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
              if (j < nx) nxt[p0 + j] = nv[j];
              v[j] = nv[j];
            }
          }
          __syncthreads();
          float *const tswap = cur;
          cur = nxt;
          nxt = tswap;
        }
      } else {
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
  float ksize[kMaxLevels];  // esigma * 1.5: the radius of the duplicate tests (the keypoint's size is twice that)
  int octave[kMaxLevels];
};

__device__ __forceinline__ int level_of_row(const LevelTab &T, int row) {
  int i = 0;
  while (i + 1 < T.n && row >= T.row0[i + 1]) ++i;
  return i;
}

constexpr int kAllPix = 4;  // pixels of a row per thread in the all-level derivative kernels
struct ScharrXyAllBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const float *__restrict__ ls, float *__restrict__ lx, float *__restrict__ ly, const LevelTab *__restrict__ Tp) {
    const LevelTab &T = *Tp;  // (in device memory, one per extractor: a gang launch carries a pointer per frame, not a table)
    const int i = level_of_row(T, blockIdx.y);
    const int y = blockIdx.y - T.row0[i], w = T.w[i], h = T.h[i];
    const float *src = ls + T.off[i];
    // (kAllPix pixels of the row per thread, blockDim apart: one pixel per thread made a 1080p frame's launch 121 000
    // workgroups of two waves, most of them of the small levels and empty)
#pragma unroll
    for (int k = 0; k < kAllPix; ++k) {
      const int x = (blockIdx.x * kAllPix + k) * blockDim.x + threadIdx.x;
      if (x < w) {
        lx[T.off[i] + (size_t)y * w + x] = scharr_at(src, w, h, x, y, 1, T.sc[i], T.ws[i], T.wm[i]);
        ly[T.off[i] + (size_t)y * w + x] = scharr_at(src, w, h, x, y, 0, T.sc[i], T.ws[i], T.wm[i]);
      }
    }
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
    const int y = blockIdx.y - T.row0[i], w = T.w[i], h = T.h[i];
    const float *lx = lx_all + T.off[i], *ly = ly_all + T.off[i];
    const int scale = T.sc[i];
    const float ws = T.ws[i], wm = T.wm[i], sf2 = (float)(scale * scale);
#pragma unroll
    for (int k = 0; k < kAllPix; ++k) {
      const int x = (blockIdx.x * kAllPix + k) * blockDim.x + threadIdx.x;
      if (x < w) {
        const float lxx = scharr_at(lx, w, h, x, y, 1, scale, ws, wm);
        const float lyy = scharr_at(ly, w, h, x, y, 0, scale, ws, wm);
        const float lxy = scharr_at(lx, w, h, x, y, 0, scale, ws, wm);
        const float a = lxx * sf2, b = lxy * sf2, c = lyy * sf2;
        ldet[T.off[i] + (size_t)y * w + x] = a * c - b * b;
      }
    }
  }
};
__global__ void k_hessian_det_all(const float *__restrict__ lx_all, const float *__restrict__ ly_all,
                          float *__restrict__ ldet, const LevelTab *__restrict__ Tp) {
  HessianDetAllBody::run(lx_all, ly_all, ldet, Tp);
}

// ---------------------------------------------------------------------------------------------------------------------
// The rest of Find_Scale_Space_Extrema + Do_Subpixel_Refinement on the device (round 3; it ran on the host before, behind
// a D2H of the candidates and two host synchronisations per image).
//
// OpenCV walks the candidates in raster order, level by level, and keeps a list of accepted points: a candidate looks
// for the FIRST accepted point (in acceptance order) of its own or the previous level within its radius; if there is one
// the stronger of the two keeps that point's place in the list, otherwise the candidate is appended; a second pass drops
// every point that has a stronger point of the next level within its radius further down the list; the survivors are
// refined to sub-pixel positions in list order.  The first pass is order dependent, but only LOCALLY: a candidate's
// outcome depends on the accepted points within its radius r, and an earlier candidate can change those only if it lies
// within 2 r (it sits there itself, or it took over a point that did).  So the candidates of a level are decided in
// ROUNDS by one workgroup: a candidate is ready when every earlier candidate of its level within 2 r (a box, slightly
// larger: conservative) has been decided, all ready candidates of a round decide at once -- two of them are never within
// 2 r of each other, so they touch disjoint points -- and a level takes as many rounds as its longest chain of such
// dependencies (a handful).  A point's place in the list is carried as a key: the raster index of the candidate that
// was appended there (appends happen in raster order, so ascending key IS acceptance order; the candidate that takes a
// point over inherits its key).  The survivors leave in ascending key order, which is the reference's output order.
//   k_extrema_seg    per 64-pixel segment of every level row: the extrema's x offsets, in order (ballot) + their count
//   k_seg_scan       exclusive sum over the segments -> where each segment's candidates start in the raster-ordered list
//   k_extrema_place  the candidates (level, x, y, response, 3x3 patch) into that list
//   k_suppress       the two passes + sub-pixel refinement + ordered compaction, one workgroup per image
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSegPx = 64;          // a segment = what one wave of the extrema kernels covers
constexpr int kSegMax = kSegPx / 2; // strict 3x3 maxima are never horizontal neighbours
constexpr int kSegPerWave = 4;      // segments of a row one wave of the extrema kernels takes

__device__ __forceinline__ bool extremum_at(const float *__restrict__ D, int w, int h, int x, int y, float dthreshold,
                                            int sigma_size_, float (&p)[9]) {
  if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1) return false;
  const float v = D[(size_t)y * w + x];
  if (!(v > dthreshold && v >= 0.00001f)) return false;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) p[(dy + 1) * 3 + dx + 1] = D[(size_t)(y + dy) * w + x + dx];
  if (!(v > p[3] && v > p[5] && v > p[0] && v > p[1] && v > p[2] && v > p[6] && v > p[7] && v > p[8])) return false;
  const float smax = 10.0f * sqrtf(2.0f);
  const int left_x = fround_d((float)x - smax * sigma_size_) - 1, right_x = fround_d((float)x + smax * sigma_size_) + 1;
  const int up_y = fround_d((float)y - smax * sigma_size_) - 1, down_y = fround_d((float)y + smax * sigma_size_) + 1;
  return !(left_x < 0 || right_x >= w || up_y < 0 || down_y >= h);
}

struct ExtremaSegBody {
  static constexpr int kGangThreads = 128;
  // (kSegPerWave segments of the row per wave: one each made a 1080p frame's launch 121 000 workgroups, most of them of the
  // small levels and empty)
  static __device__ __forceinline__ void run(const float *__restrict__ ldet_all, const LevelTab *__restrict__ Tp,
                                             float dthreshold, uint8_t *__restrict__ seg_x, unsigned int *__restrict__ seg_cnt,
                                             unsigned int segs_per_row) {
    const LevelTab &T = *Tp;
    const int i = level_of_row(T, blockIdx.y);
    const int y = blockIdx.y - T.row0[i];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < kSegPerWave; ++k) {
      const unsigned int sidx = (blockIdx.x * kSegPerWave + k) * 2u + (threadIdx.x >> 6);  // the segment of the row
      if (sidx >= segs_per_row) break;
      const int x = (int)sidx * kSegPx + lane;
      float p[9];
      const bool is = extremum_at(ldet_all + T.off[i], T.w[i], T.h[i], x, y, dthreshold, T.sigma_size[i], p);
      const unsigned long long m = __ballot(is);
      const unsigned int seg = blockIdx.y * segs_per_row + sidx;
      if (is) seg_x[(size_t)seg * kSegMax + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
      if (lane == 0) seg_cnt[seg] = (unsigned int)__popcll(m);
    }
  }
};
__global__ __launch_bounds__(128) void k_extrema_seg(const float *__restrict__ ldet_all, const LevelTab *__restrict__ Tp,
                                                     float dthreshold, uint8_t *__restrict__ seg_x,
                                                     unsigned int *__restrict__ seg_cnt, unsigned int segs_per_row) {
  ExtremaSegBody::run(ldet_all, Tp, dthreshold, seg_x, seg_cnt, segs_per_row);
}

// exclusive sum of seg_cnt[0 .. n_seg) in place, the total in seg_cnt[n_seg] and in *n_out.  Each of the 16 waves owns a
// contiguous chunk: it sums it with coalesced loads, the 16 totals meet in LDS, then the wave walks its chunk again with a
// shuffle scan and a running carry (no block barrier inside the walks).
struct SegScanBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(unsigned int *__restrict__ seg_cnt, unsigned int n_seg, unsigned int cap,
                                             unsigned int *__restrict__ n_out) {
    __shared__ unsigned int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (four consecutive entries per lane and step -- one 16-byte load, a 1 KB line per wave --: a wave's walk is a chain of
    // load -> shuffle scan -> store steps, and at one entry per lane a VGA image's 36 000 segments were 35 of them)
    const unsigned int chunk = ((n_seg + 15u) / 16u + 255u) & ~255u;
    const unsigned int lo = min(n_seg, (unsigned int)wv * chunk), hi = min(n_seg, lo + chunk);
    auto load4 = [&](unsigned int k, unsigned int (&c)[4]) {  // entries k .. k + 3, zero past hi (k is a multiple of 4)
      if (k + 4 <= hi) {
        const uint4 v = *reinterpret_cast<const uint4 *>(seg_cnt + k);
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = k + u < hi ? seg_cnt[k + u] : 0u;
      }
    };
    unsigned int sum = 0;
    for (unsigned int k = lo + 4 * lane; k < hi; k += 256) {
      unsigned int c[4];
      load4(k, c);
      sum += (c[0] + c[1]) + (c[2] + c[3]);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) wsum[wv] = sum;
    __syncthreads();
    unsigned int carry = 0;
    for (int q = 0; q < wv; ++q) carry += wsum[q];
    for (unsigned int k0 = lo; k0 < hi; k0 += 256) {
      const unsigned int k = k0 + 4 * lane;
      unsigned int c[4] = {0u, 0u, 0u, 0u};
      if (k < hi) load4(k, c);
      const unsigned int mine = (c[0] + c[1]) + (c[2] + c[3]);
      unsigned int inc = mine;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
      }
      unsigned int run = carry + inc - mine;
      if (k + 4 <= hi) {
        uint4 o4;
        o4.x = run; o4.y = run + c[0]; o4.z = run + c[0] + c[1]; o4.w = run + c[0] + c[1] + c[2];
        *reinterpret_cast<uint4 *>(seg_cnt + k) = o4;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (k + u < hi) seg_cnt[k + u] = run;
          run += c[u];
        }
      }
      carry += __shfl(inc, 63, 64);
    }
    if (tid == 0) {
      unsigned int total = 0;
      for (int q = 0; q < 16; ++q) total += wsum[q];
      seg_cnt[n_seg] = total;
      *n_out = total;
      n_out[6] = 0u;  // (k_suppress_nbr counts the spilled records of the frame here)
      n_out[7] = 0u;
    }
    (void)cap;
  }
};
// The same for a large image (a 1080p frame has 243 000 segments: 1 MB that ONE workgroup reads twice and writes once
// -- 50 us, the reach of a single compute unit's memory path): two launches over chunks of kSegChunk entries.
//   k_seg_sum    workgroup g adds up chunk g -> part[g]; the workgroup that arrives last turns part[] into the chunks'
//                starting sums (exclusive scan) and writes the total (nobody waits: the k_pre_hist hand-over)
//   k_seg_write  workgroup g scans chunk g from part[g]
constexpr unsigned int kSegChunk = 4096;  // 4 entries per thread of a 1 024-thread workgroup
__device__ __forceinline__ unsigned int block_scan_1024(unsigned int v, unsigned int *wtot /*[16] LDS*/, unsigned int *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  __syncthreads();  // (wtot of a previous call is no longer read)
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  unsigned int before = 0, all = 0;
  for (int q = 0; q < 16; ++q) {
    if (q < wv) before += wtot[q];
    all += wtot[q];
  }
  if (total) *total = all;
  return before + inc - v;  // exclusive
}
struct SegSumBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(unsigned int *__restrict__ seg_cnt, unsigned int n_seg,
                                             unsigned int *__restrict__ part /*[chunks] sums, [chunks] arrivals*/,
                                             unsigned int *__restrict__ n_out) {
    __shared__ unsigned int wtot[16];
    __shared__ unsigned int s_ticket;
    const unsigned int chunks = gridDim.x, k = blockIdx.x * kSegChunk + 4u * threadIdx.x;
    unsigned int c = 0;
    if (k + 4u <= n_seg) {
      const uint4 v = *reinterpret_cast<const uint4 *>(seg_cnt + k);
      c = (v.x + v.y) + (v.z + v.w);
    } else {
      for (unsigned int u = 0; u < 4u; ++u) c += k + u < n_seg ? seg_cnt[k + u] : 0u;
    }
    unsigned int total = 0;
    (void)block_scan_1024(c, wtot, &total);
    if (threadIdx.x == 0) {
      __hip_atomic_store(&part[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s_ticket = atomicAdd(&part[chunks], 1u);
    }
    __syncthreads();
    if (s_ticket != chunks - 1u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // (chunks <= 1 024: the launcher's bound)
    const unsigned int mine = threadIdx.x < chunks ? __hip_atomic_load(&part[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    unsigned int all = 0;
    const unsigned int before = block_scan_1024(mine, wtot, &all);
    if (threadIdx.x < chunks) part[threadIdx.x] = before;
    if (threadIdx.x == 0) {
      part[chunks] = 0u;  // the next frame counts from zero
      seg_cnt[n_seg] = all;
      *n_out = all;
      n_out[6] = 0u;  // (k_suppress_nbr counts the spilled records of the frame here)
      n_out[7] = 0u;
    }
  }
};
__global__ __launch_bounds__(1024) void k_seg_sum(unsigned int *__restrict__ seg_cnt, unsigned int n_seg,
                                                  unsigned int *__restrict__ part, unsigned int *__restrict__ n_out) {
  SegSumBody::run(seg_cnt, n_seg, part, n_out);
}
struct SegWriteBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(unsigned int *__restrict__ seg_cnt, unsigned int n_seg,
                                             const unsigned int *__restrict__ part) {
    __shared__ unsigned int wtot[16];
    const unsigned int k = blockIdx.x * kSegChunk + 4u * threadIdx.x;
    unsigned int c[4] = {0u, 0u, 0u, 0u};
    if (k + 4u <= n_seg) {
      const uint4 v = *reinterpret_cast<const uint4 *>(seg_cnt + k);
      c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    } else {
      for (unsigned int u = 0; u < 4u; ++u) c[u] = k + u < n_seg ? seg_cnt[k + u] : 0u;
    }
    unsigned int run = part[blockIdx.x] + block_scan_1024((c[0] + c[1]) + (c[2] + c[3]), wtot, nullptr);
    if (k + 4u <= n_seg) {
      uint4 o4;
      o4.x = run; o4.y = run + c[0]; o4.z = run + c[0] + c[1]; o4.w = run + c[0] + c[1] + c[2];
      *reinterpret_cast<uint4 *>(seg_cnt + k) = o4;
    } else {
      for (unsigned int u = 0; u < 4u; ++u) {
        if (k + u < n_seg) seg_cnt[k + u] = run;
        run += c[u];
      }
    }
  }
};
__global__ __launch_bounds__(1024) void k_seg_write(unsigned int *__restrict__ seg_cnt, unsigned int n_seg,
                                                    const unsigned int *__restrict__ part) {
  SegWriteBody::run(seg_cnt, n_seg, part);
}

__global__ __launch_bounds__(1024) void k_seg_scan(unsigned int *__restrict__ seg_cnt, unsigned int n_seg, unsigned int cap,
                                                   unsigned int *__restrict__ n_out) {
  SegScanBody::run(seg_cnt, n_seg, cap, n_out);
}

struct CandArrays {
  uint32_t *xy;           // level coordinates: x | y << 16
  uint8_t *level;         // the level; | 0x80: Do_Subpixel_Refinement rejects the candidate
  float *resp;            // |Ldet| at the candidate
  float2 *sub;            // the sub-pixel offsets from the candidate's 3 x 3 neighbourhood
};

struct ExtremaPlaceBody {
  static constexpr int kGangThreads = 128;
  static __device__ __forceinline__ void run(const float *__restrict__ ldet_all, const LevelTab *__restrict__ Tp,
                                             const uint8_t *__restrict__ seg_x, const unsigned int *__restrict__ seg_base,
                                             unsigned int cap, CandArrays C, unsigned int segs_per_row) {
    const LevelTab &T = *Tp;
    const int i = level_of_row(T, blockIdx.y);
    const int y = blockIdx.y - T.row0[i], w = T.w[i];
    const int lane = threadIdx.x & 63;
    const float *D = ldet_all + T.off[i];
#pragma unroll
    for (int k = 0; k < kSegPerWave; ++k) {
      const unsigned int sidx = (blockIdx.x * kSegPerWave + k) * 2u + (threadIdx.x >> 6);
      if (sidx >= segs_per_row) break;
      const unsigned int seg = blockIdx.y * segs_per_row + sidx;
      const unsigned int base = seg_base[seg], cnt = seg_base[seg + 1] - base;
      if ((unsigned int)lane >= cnt || base + lane >= cap) continue;
      const int x = (int)sidx * kSegPx + seg_x[(size_t)seg * kSegMax + lane];
      const unsigned int g = base + lane;
      float p[9];
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) p[(dy + 1) * 3 + dx + 1] = D[(size_t)(y + dy) * w + x + dx];
      // Do_Subpixel_Refinement's 2 x 2 system here, where the neighbourhood is at hand and every candidate has a thread
      const float Dx = 0.5f * (p[5] - p[3]);
      const float Dy = 0.5f * (p[7] - p[1]);
      const float Dxx = (p[5] + p[3]) - 2.0f * p[4];
      const float Dyy = (p[7] + p[1]) - 2.0f * p[4];
      const float Dxy = 0.25f * (p[8] + p[0]) - 0.25f * (p[2] + p[6]);
      const float det = Dxx * Dyy - Dxy * Dxy;
      float d0 = 0.0f, d1 = 0.0f;
      bool ok = false;
      if (det != 0.0f) {
        d0 = (-Dx * Dyy + Dy * Dxy) / det;
        d1 = (-Dy * Dxx + Dx * Dxy) / det;
        ok = fabsf(d0) <= 1.0f && fabsf(d1) <= 1.0f;
      }
      C.xy[g] = (uint32_t)x | ((uint32_t)y << 16);
      C.level[g] = (uint8_t)(i | (ok ? 0 : 0x80));
      C.sub[g] = make_float2(d0, d1);
      C.resp[g] = fabsf(p[4]);
    }
  }
};
__global__ __launch_bounds__(128) void k_extrema_place(const float *__restrict__ ldet_all, const LevelTab *__restrict__ Tp,
                                                       const uint8_t *__restrict__ seg_x,
                                                       const unsigned int *__restrict__ seg_base, unsigned int cap,
                                                       CandArrays C, unsigned int segs_per_row) {
  ExtremaPlaceBody::run(ldet_all, Tp, seg_x, seg_base, cap, C, segs_per_row);
}

// ---- the suppression's neighbour lists ---------------------------------------------------------------------------------
// Which candidates CAN interact is geometry, known before any of them is decided; only their statuses change during the
// passes.  k_suppress_nbr finds, one wave per candidate and over the whole chip, the few candidates each one will ever look
// at -- A: the earlier candidates of its level within 2 r (they decide when it is ready; those within r are also the only
// points of its own level it can meet: a later candidate within r is blocked by it and still undecided when it decides),
// P: the candidates of the previous level within r, N (second pass): the stronger candidates of the next level within r
// -- with exactly the tests of the scans they replace, so that the one workgroup of k_suppress walks a handful of list
// entries per candidate and round instead of bands of rows (a round of 1 000 candidates: 9 us of band scans).  A list that
// does not fit its record (15 entries for A + P, 3 for N) or its 15 / 16-bit offsets is marked "spilled" and k_suppress
// scans for that candidate as before.
//   record (16 x u16): [0] = nA | nP << 4 | spilled << 15; A entries = (g - o) | within_r << 15; then P entries = o - p0
//   record2 (4 x u16): [0] = nN | spilled << 15; N entries = o - n0          (p0 / n0: where the other level starts)
constexpr int kNbrEntries = 15;
constexpr int kNbr2Entries = 3;
struct NbrBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(const LevelTab *__restrict__ Tp, const unsigned int *__restrict__ seg_base,
                                             unsigned int spr, unsigned int cap, const unsigned int *__restrict__ n_cand,
                                             const uint32_t *__restrict__ cxy, const uint8_t *__restrict__ clevel,
                                             const float *__restrict__ cresp, uint4 *__restrict__ nbr, uint2 *__restrict__ nbr2,
                                             unsigned int *__restrict__ spilled /*[2] diagnostics*/, unsigned int force_spill) {
    const LevelTab &T = *Tp;
    const unsigned int N = min(*n_cand, cap);
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned int n_waves = gridDim.x * (kGangThreads / 64);
    auto row_start = [&](int lv, int yy) -> unsigned int { return min(seg_base[(unsigned int)(T.row0[lv] + yy) * spr], N); };
    for (unsigned int g = blockIdx.x * (kGangThreads / 64) + (threadIdx.x >> 6); g < N; g += n_waves) {
      const int l = clevel[g] & 0x7F;
      const uint32_t me = cxy[g];
      const int x = (int)(me & 0xFFFFu), y = (int)(me >> 16);
      const float size = T.ksize[l];
      const float ratio = (float)(1 << T.octave[l]);
      const float px = (float)x * ratio, py = (float)y * ratio;
      uint16_t *rec = reinterpret_cast<uint16_t *>(nbr + 2 * (size_t)g);
      uint16_t *rec2 = reinterpret_cast<uint16_t *>(nbr2 + g);
      unsigned int nA = 0, nP = 0, nN = 0;
      bool spill = false, spill2 = false;
      {  // A: earlier in raster order -- the rows above within reach, and this row up to the candidate itself
        const float dep2 = 4.0f * size * size * 1.0001f + 0.01f;   // (2 r)^2 with slack: conservative
        const int dep = (int)ceilf(2.0f * size / ratio) + 1;
        const int hit = (int)ceilf(size / ratio) + 1;
        const unsigned int from = row_start(l, max(y - dep, 0));
        for (unsigned int o0 = from; o0 < g; o0 += 64) {
          const unsigned int o = o0 + lane;
          bool in = false, within = false;
          if (o < g) {
            const uint32_t q = cxy[o];
            const int xo = (int)(q & 0xFFFFu), yo = (int)(q >> 16);
            const int ddx = x - xo, ddy = y - yo;
            if (ddx <= dep && ddx >= -dep) {
              const float dx = (float)ddx * ratio, dy = (float)ddy * ratio;
              in = dx * dx + dy * dy <= dep2;
              if (in && ddx <= hit && ddx >= -hit && ddy <= hit) {  // (the band and the box of the decision's own-level scan)
                const float ex = px - (float)xo * ratio, ey = py - (float)yo * ratio;
                within = ex * ex + ey * ey <= size * size;
              }
            }
          }
          const unsigned long long m = __ballot(in);
          const unsigned int d = g - o;
          if (in) {
            const unsigned int pos = nA + (unsigned int)__popcll(m & below);
            if (pos < (unsigned int)kNbrEntries && d < 32768u) rec[1 + pos] = (uint16_t)(d | (within ? 0x8000u : 0u));
          }
          if (__ballot(in && d >= 32768u) != 0ull) spill = true;
          nA += (unsigned int)__popcll(m);
        }
      }
      if (l > 0) {  // P: the level before
        const float r1 = (float)(1 << T.octave[l - 1]);
        const int wp = T.w[l - 1], hp = T.h[l - 1];
        const int x0 = max((int)floorf((px - size) / r1) - 1, 0), x1 = min((int)ceilf((px + size) / r1) + 1, wp - 1);
        const int y0 = max((int)floorf((py - size) / r1) - 1, 0), y1 = min((int)ceilf((py + size) / r1) + 1, hp - 1);
        if (x0 <= x1 && y0 <= y1) {
          const unsigned int p0 = row_start(l - 1, 0), from = row_start(l - 1, y0), to = row_start(l - 1, y1 + 1);
          for (unsigned int o0 = from; o0 < to; o0 += 64) {
            const unsigned int o = o0 + lane;
            bool in = false;
            if (o < to) {
              const uint32_t q = cxy[o];
              const int xo = (int)(q & 0xFFFFu), yo = (int)(q >> 16);
              if (xo >= x0 && xo <= x1) {
                const float dx = px - (float)xo * r1, dy = py - (float)yo * r1;
                in = dx * dx + dy * dy <= size * size;
              }
            }
            const unsigned long long m = __ballot(in);
            const unsigned int d = o - p0;
            if (in) {
              const unsigned int pos = nA + nP + (unsigned int)__popcll(m & below);
              if (pos < (unsigned int)kNbrEntries && d < 65536u) rec[1 + pos] = (uint16_t)d;
            }
            if (__ballot(in && d >= 65536u) != 0ull) spill = true;
            nP += (unsigned int)__popcll(m);
          }
        }
      }
      if (nA + nP > (unsigned int)kNbrEntries || force_spill) spill = true;
      if (lane == 0) {
        rec[0] = spill ? (uint16_t)0x8000u : (uint16_t)(nA | (nP << 4));
        if (spill) atomicAdd(&spilled[0], 1u);
      }
      if (l + 1 < T.n) {  // N: the next level, stronger (the second pass)
        const float rsp = cresp[g];
        const float rn = (float)(1 << T.octave[l + 1]);
        const int wn = T.w[l + 1], hn = T.h[l + 1];
        const int x0 = max((int)floorf((px - size) / rn) - 1, 0), x1 = min((int)ceilf((px + size) / rn) + 1, wn - 1);
        const int y0 = max((int)floorf((py - size) / rn) - 1, 0), y1 = min((int)ceilf((py + size) / rn) + 1, hn - 1);
        if (x0 <= x1 && y0 <= y1) {
          const unsigned int n0 = row_start(l + 1, 0), from = row_start(l + 1, y0), to = row_start(l + 1, y1 + 1);
          for (unsigned int o0 = from; o0 < to; o0 += 64) {
            const unsigned int o = o0 + lane;
            bool in = false;
            if (o < to) {
              const uint32_t q = cxy[o];
              const int xo = (int)(q & 0xFFFFu), yo = (int)(q >> 16);
              if (xo >= x0 && xo <= x1) {
                const float dx = px - (float)xo * rn, dy = py - (float)yo * rn;
                in = dx * dx + dy * dy <= size * size && rsp < cresp[o];
              }
            }
            const unsigned long long m = __ballot(in);
            const unsigned int d = o - n0;
            if (in) {
              const unsigned int pos = nN + (unsigned int)__popcll(m & below);
              if (pos < (unsigned int)kNbr2Entries && d < 65536u) rec2[1 + pos] = (uint16_t)d;
            }
            if (__ballot(in && d >= 65536u) != 0ull) spill2 = true;
            nN += (unsigned int)__popcll(m);
          }
        }
      }
      if (nN > (unsigned int)kNbr2Entries || force_spill) spill2 = true;
      if (lane == 0) {
        rec2[0] = spill2 ? (uint16_t)0x8000u : (uint16_t)nN;
        if (spill2) atomicAdd(&spilled[1], 1u);
      }
    }
  }
};
__global__ __launch_bounds__(256) void k_suppress_nbr(const LevelTab *__restrict__ Tp, const unsigned int *__restrict__ seg_base,
                                                      unsigned int spr, unsigned int cap, const unsigned int *__restrict__ n_cand,
                                                      const uint32_t *__restrict__ cxy, const uint8_t *__restrict__ clevel,
                                                      const float *__restrict__ cresp, uint4 *__restrict__ nbr,
                                                      uint2 *__restrict__ nbr2, unsigned int *__restrict__ spilled,
                                                      unsigned int force_spill) {
  NbrBody::run(Tp, seg_base, spr, cap, n_cand, cxy, clevel, cresp, nbr, nbr2, spilled, force_spill);
}
// a record in registers: its u16 are taken from the bottom, one at a time (the whole record moves down 16 bits: indexing
// registers by a run-time number would put the record into scratch memory, and unrolling the walks by 15 made the
// kernel 17 000 instructions that spilled)
struct NbrRec {
  uint32_t w[8];
  __device__ __forceinline__ void load(const uint4 *p) {
    const uint4 a = p[0], b = p[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  }
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = 0u;
  }
  __device__ __forceinline__ uint32_t pop() {
    const uint32_t e = w[0] & 0xFFFFu;
#pragma unroll
    for (int j = 0; j < 7; ++j) w[j] = (w[j] >> 16) | (w[j + 1] << 16);
    w[7] >>= 16;
    return e;
  }
};

struct SuppressArgs {
  const LevelTab *T;
  const unsigned int *seg_base;  // [n_seg + 1]
  unsigned int segs_per_row;     // 64-pixel segments of an image row (2 x ceil(width / 128))
  unsigned int cap;
  CandArrays C;
  const uint4 *nbr;              // [cap x 2] k_suppress_nbr: the A and P lists
  const uint2 *nbr2;             // [cap] the N lists
  uint8_t *status;               // per candidate: 0 undecided, 1 holds a point of the list, 2 dropped, 3 taken over
  uint8_t *ready;
  unsigned int *key;             // the list position a candidate holds (raster index of the candidate appended there)
  int *slot;                     // [cap] key -> the candidate that holds it at the end (-1: nobody), then output positions
  float *kp;                     // out [n x 4]: x, y, size, class_id
  float *resp_out;               // out [n]
  unsigned int *n_kp;            // out
  unsigned int *n_cand;          // in (k_seg_scan)
  unsigned int *rounds_out;      // diagnostics: rounds the first pass took over all levels
  unsigned int force_global;     // test hook: every level out of the global arrays (the form of a level above kSupLevelCap)
};

enum : uint8_t { kUndecided = 0, kHolds = 1, kDropped = 2, kTakenOver = 3 };

// The first pass works out of LDS: one packed word per candidate -- x | y << 12 | status << 24: a status is its own byte
// (ds_write_b8), and a spilled candidate's scan fetches FOUR neighbours with one ds_read_b128 -- and the keys and
// responses, of the level and of the one before it, plus the row starts (raster index of a row's first candidate; a band
// of rows is ONE contiguous index range) for the spilled candidates' scans.  A level with more candidates than the LDS
// arrays hold, or wider than 12 bits of x, works on the global arrays through the same code.
constexpr int kSupMaxRows = 4096 + 1;  // level rows the row-start tables hold (sfmloc_akaze_create: height <= 4096)
constexpr int kSupLevelCap = 4096;     // candidates of one level kept in LDS
constexpr int kSupOwn = kSupLevelCap / 1024;  // candidates of an LDS level per thread: their records stay in registers
constexpr int kSupMaxW = 4096;         // level width the packed word holds
struct SuppressLds {
  alignas(16) uint32_t pk[2][kSupLevelCap];
  unsigned int key[2][kSupLevelCap];
  float resp[2][kSupLevelCap];
  unsigned int row[2][kSupMaxRows];
  unsigned int lvl[kMaxLevels + 1];   // where each level starts in the raster-ordered list
  unsigned int left[2];
  unsigned int wsum[16];
};
struct SuppressBody {
  static constexpr int kGangThreads = 1024;

  // candidates [from, to) of a level, four per step: f(o, x, y, status) -> true ends the scan.  LDS: the level's packed
  // words (slot pk, index = raster index - base); otherwise the global arrays.
  template <bool LDS, class F>
  static __device__ __forceinline__ void scan(const SuppressArgs &A, const uint32_t *pk, unsigned int base, unsigned int from,
                                              unsigned int to, F &&f) {
    if (LDS) {
      for (unsigned int o4 = base + ((from - base) & ~3u); o4 < to; o4 += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pk + (o4 - base));
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
        bool stop = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned int o = o4 + u;
          if (o < from || o >= to) continue;
          if (f(o, (int)(q[u] & 0xFFFu), (int)((q[u] >> 12) & 0xFFFu), (uint8_t)(q[u] >> 24))) stop = true;
        }
        if (stop) break;
      }
    } else {
      for (unsigned int o4 = from; o4 < to; o4 += 4) {
        uint32_t q[4];
        uint8_t sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // (the loads of a step issued together: the scan is a chain of latencies otherwise)
          const unsigned int o = min(o4 + u, to - 1);
          q[u] = A.C.xy[o];
          sv[u] = A.status[o];
        }
        bool stop = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned int o = o4 + u;
          if (o >= to) continue;
          if (f(o, (int)(q[u] & 0xFFFFu), (int)(q[u] >> 16), sv[u])) stop = true;
        }
        if (stop) break;
      }
    }
  }

  // One level of the first pass.  CL / PL: this level's / the previous level's tables are in LDS (slot b / b ^ 1, indexed
  // by raster index minus c0 / p0) or in the global arrays -- compile-time, so that the LDS form is ds_read / ds_write and
  // not a generic (flat) access, which costs a global-memory round trip even when it lands in LDS.  rc: the records of
  // this thread's candidates (LDS form; the global form reads them as it goes).
  template <bool CL, bool PL>
  static __device__ __forceinline__ void level(const SuppressArgs &A, const LevelTab &T, SuppressLds &L, int l, int b,
                                               unsigned int c0, unsigned int c1, unsigned int p0, unsigned int &rounds,
                                               const NbrRec (&rc)[kSupOwn]) {
    const int tid = threadIdx.x;
    const int h = T.h[l];
    const unsigned int *R = L.row[b], *Rp = L.row[b ^ 1];
    const uint32_t *pk_c = L.pk[b], *pk_p = L.pk[b ^ 1];
    auto st_c = [&](unsigned int g) __attribute__((always_inline)) -> uint8_t & {
      return CL ? reinterpret_cast<uint8_t *>(&L.pk[b][g - c0])[3] : A.status[g];
    };
    auto st_p = [&](unsigned int g) __attribute__((always_inline)) -> uint8_t & {
      return PL ? reinterpret_cast<uint8_t *>(&L.pk[b ^ 1][g - p0])[3] : A.status[g];
    };
    auto key_c = [&](unsigned int g) __attribute__((always_inline)) -> unsigned int & { return CL ? L.key[b][g - c0] : A.key[g]; };
    auto key_p = [&](unsigned int g) __attribute__((always_inline)) -> unsigned int { return PL ? L.key[b ^ 1][g - p0] : A.key[g]; };
    auto rsp_c = [&](unsigned int g) __attribute__((always_inline)) -> float { return CL ? L.resp[b][g - c0] : A.C.resp[g]; };
    auto rsp_p = [&](unsigned int g) __attribute__((always_inline)) -> float { return PL ? L.resp[b ^ 1][g - p0] : A.C.resp[g]; };
    // a thread's own candidate: coordinates and status
    auto own = [&](unsigned int g, int &x, int &y) __attribute__((always_inline)) -> uint8_t {
      if (CL) {
        const uint32_t w = L.pk[b][g - c0];
        x = (int)(w & 0xFFFu);
        y = (int)((w >> 12) & 0xFFFu);
        return (uint8_t)(w >> 24);
      }
      const uint32_t me = A.C.xy[g];
      x = (int)(me & 0xFFFFu);
      y = (int)(me >> 16);
      return A.status[g];
    };
    // this thread's candidates in turn: f(g, record, k); k = which of the thread's (compile-time in the LDS form).  (A macro
    // and not a function that takes f: a closure handed on as an argument kept everything it captures in scratch memory.)
#define SFM_FOR_OWN(f)                                                   \
  if constexpr (CL) {                                                    \
    _Pragma("unroll") for (int k = 0; k < kSupOwn; ++k) {                \
      const unsigned int g = c0 + tid + 1024u * k;                       \
      if (g < c1) f(g, rc[k], k);                                        \
    }                                                                    \
  } else {                                                               \
    for (unsigned int g = c0 + tid; g < c1; g += 1024) {                 \
      NbrRec r;                                                          \
      r.load(A.nbr + 2 * (size_t)g);                                     \
      f(g, r, 0);                                                        \
    }                                                                    \
  }
    const float size = T.ksize[l];
    const float ratio = (float)(1 << T.octave[l]);
    const float dep2 = 4.0f * size * size * 1.0001f + 0.01f;   // (2 r)^2 with slack: conservative
    const int dep = (int)ceilf(2.0f * size / ratio) + 1;
    const int hit = (int)ceilf(size / ratio) + 1;
    const float r1 = l > 0 ? (float)(1 << T.octave[l - 1]) : 1.0f;
    const int wp = l > 0 ? T.w[l - 1] : 0, hp = l > 0 ? T.h[l - 1] : 0;
    const int lane = tid & 63;
    unsigned long long ta = 0, tb = 0;
    for (; c0 != c1;) {
      const unsigned long long t0 = wall_clock64();
      // (a) who is ready: no earlier undecided candidate of this level within 2 r (its A list); is anybody undecided at all
      // (the LDS form keeps a thread's "ready" bits in a register)
      unsigned int undecided = 0, ready_bits = 0;
      auto phase_a = [&](unsigned int g, const NbrRec &r, int k) __attribute__((always_inline)) {
        int x, y;
        if (own(g, x, y) != kUndecided) return;
        ++undecided;
        bool blocked = false;
        NbrRec t = r;
        const uint32_t hd = t.pop();
        if (hd & 0x8000u) {  // spilled: the rows above within reach, and this row up to the candidate itself
          scan<CL>(A, pk_c, c0, R[max(y - dep, 0)], g, [&](unsigned int, int xo, int yo, uint8_t sv) __attribute__((always_inline)) -> bool {
            const int ddx = x - xo, ddy = y - yo;
            if (ddx > dep || ddx < -dep || sv != kUndecided) return false;
            const float dx = (float)ddx * ratio, dy = (float)ddy * ratio;
            if (dx * dx + dy * dy <= dep2) blocked = true;
            return blocked;
          });
        } else {
          const int nA = (int)(hd & 15u);
          for (int i = 0; __ballot(i < nA) != 0ull; ++i) {  // (as many steps as the longest list of the lanes that are here)
            const uint32_t e = t.pop();
            const unsigned int o = g - (i < nA ? (e & 0x7FFFu) : 0u);
            const uint8_t sv = st_c(o);
            if (i < nA && sv == kUndecided) blocked = true;
          }
        }
        if (CL) ready_bits |= (blocked ? 0u : 1u) << k;
        else A.ready[g] = blocked ? 0 : 1;
      };
      SFM_FOR_OWN(phase_a)
      // (one plain store per wave that has an undecided candidate: only "none left" is asked)
      if (__ballot(undecided != 0) != 0ull && lane == 0) L.left[rounds & 1] = 1u;
      if (tid == 0) L.left[(rounds + 1) & 1] = 0u;
      __syncthreads();
      const unsigned long long t1 = wall_clock64();
      ta += t1 - t0;
      if (L.left[rounds & 1] == 0u) {
        ++rounds;
        break;
      }
      // (b) the ready ones decide: the first point of the list (lowest key) among the holders within r -- of this level
      // (A entries marked "within r") and of the level before (P entries)
      // (tried in round 3: counting the decided candidates so that a round which decides everything ends the level --
      // 33 -> 20 rounds on a sparse VGA frame, 34 -> 23 on a rich one -- and measured SLOWER, 77 -> 90 us and 250 -> 260 us:
      // the closing round that finds nothing left costs less than the count's LDS atomics and the second exit cost every
      // other round)
      auto phase_b = [&](unsigned int g, const NbrRec &r, int k) __attribute__((always_inline)) {
        int x, y;
        if (own(g, x, y) != kUndecided) return;
        if (CL ? !((ready_bits >> k) & 1u) : !A.ready[g]) return;
        const float px = (float)x * ratio, py = (float)y * ratio;
        unsigned int first = 0xFFFFFFFFu, first_key = 0xFFFFFFFFu;
        NbrRec t = r;
        const uint32_t hd = t.pop();
        if (hd & 0x8000u) {  // spilled: the bands of rows within r
          scan<CL>(A, pk_c, c0, R[max(y - hit, 0)], R[min(y + hit, h - 1) + 1], [&](unsigned int o, int xo, int yo, uint8_t sv) __attribute__((always_inline)) -> bool {
            const int ddx = x - xo;
            if (ddx > hit || ddx < -hit || o == g || sv != kHolds) return false;
            const float dx = px - (float)xo * ratio, dy = py - (float)yo * ratio;
            if (dx * dx + dy * dy <= size * size) {
              const unsigned int ko = key_c(o);
              if (ko < first_key) {
                first_key = ko;
                first = o;
              }
            }
            return false;
          });
          if (l > 0) {  // the level before
            const int x0 = max((int)floorf((px - size) / r1) - 1, 0), x1 = min((int)ceilf((px + size) / r1) + 1, wp - 1);
            const int y0 = max((int)floorf((py - size) / r1) - 1, 0), y1 = min((int)ceilf((py + size) / r1) + 1, hp - 1);
            if (x0 <= x1 && y0 <= y1) {
              scan<PL>(A, pk_p, p0, Rp[y0], Rp[y1 + 1], [&](unsigned int o, int xo, int yo, uint8_t sv) __attribute__((always_inline)) -> bool {
                if (xo < x0 || xo > x1 || sv != kHolds) return false;
                const float dx = px - (float)xo * r1, dy = py - (float)yo * r1;
                if (dx * dx + dy * dy <= size * size) {
                  const unsigned int ko = key_p(o);
                  if (ko < first_key) {
                    first_key = ko;
                    first = o;
                  }
                }
                return false;
              });
            }
          }
        } else {
          const int nA = (int)(hd & 15u), nAP = nA + (int)((hd >> 4) & 15u);
          for (int i = 0; __ballot(i < nAP) != 0ull; ++i) {
            const uint32_t e = t.pop();
            if (i < nA) {
              if (e & 0x8000u) {
                const unsigned int o = g - (e & 0x7FFFu);
                if (st_c(o) == kHolds) {
                  const unsigned int ko = key_c(o);
                  if (ko < first_key) {
                    first_key = ko;
                    first = o;
                  }
                }
              }
            } else if (i < nAP) {
              const unsigned int o = p0 + e;
              if (st_p(o) == kHolds) {
                const unsigned int ko = key_p(o);
                if (ko < first_key) {
                  first_key = ko;
                  first = o;
                }
              }
            }
          }
        }
        if (first == 0xFFFFFFFFu) {
          st_c(g) = kHolds;  // appended: its own raster index is its place
        } else if (rsp_c(g) > (first >= c0 ? rsp_c(first) : rsp_p(first))) {
          key_c(g) = first_key;  // takes the point over, keeps its place
          st_c(g) = kHolds;
          if (first >= c0) st_c(first) = kTakenOver;
          else st_p(first) = kTakenOver;
        } else {
          st_c(g) = kDropped;
        }
      };
      SFM_FOR_OWN(phase_b)
      ++rounds;
      __syncthreads();
      tb += wall_clock64() - t1;
    }
    if (tid == 0 && A.rounds_out) {  // diagnostics: 10 ns ticks of the level's (a) and (b) phases
      A.rounds_out[80 + 2 * l] = (unsigned int)ta;
      A.rounds_out[81 + 2 * l] = (unsigned int)tb;
    }
#undef SFM_FOR_OWN
  }

  static __device__ __forceinline__ void run(SuppressArgs A) {
    const LevelTab &T = *A.T;
    extern __shared__ __attribute__((aligned(16))) unsigned char sup_smem[];
    SuppressLds &L = *reinterpret_cast<SuppressLds *>(sup_smem);
    const int tid = threadIdx.x;
    const unsigned int N = min(*A.n_cand, A.cap);
    for (unsigned int g = tid; g < N; g += 1024) {
      A.status[g] = kUndecided;
      A.key[g] = g;
      A.slot[g] = -1;
    }
    unsigned int rounds = 0, global_levels = 0;
    const unsigned long long t_start = wall_clock64();
    if (tid < 2) L.left[tid] = 0u;
    if (tid <= T.n) L.lvl[tid] = min(A.seg_base[(unsigned int)T.row0[tid] * A.segs_per_row], N);
    // level l's tables sit in slot l & 1 of the LDS arrays (or, a level with more than kSupLevelCap candidates, in the
    // global arrays); the previous level's stay where they are while level l is decided
    bool prev_lds = false;
    unsigned int p0 = 0, p1 = 0;
    __syncthreads();
    // ---- first pass, level by level ----
    for (int l = 0; l < T.n; ++l) {
      const int b = l & 1;
      const int h = T.h[l];
      // the level's extent straight from the segment table (every thread the same two loads: no barrier in front of the
      // fills), then row starts, coordinates and the candidates' neighbour records under ONE barrier
      const unsigned long long t_l0 = wall_clock64();
      const unsigned int rounds_before = rounds;
      const unsigned int c0 = L.lvl[l], c1 = L.lvl[l + 1];
      for (int y = tid; y <= h; y += 1024) L.row[b][y] = min(A.seg_base[(unsigned int)(T.row0[l] + y) * A.segs_per_row], N);
      const bool in_lds = c1 - c0 <= (unsigned int)kSupLevelCap && T.w[l] <= kSupMaxW && !A.force_global;
      if (!in_lds && c1 != c0) ++global_levels;
      NbrRec rc[kSupOwn];
      if (in_lds) {
#pragma unroll
        for (int k = 0; k < kSupOwn; ++k) {
          const unsigned int g = c0 + tid + 1024u * k;
          if (g < c1) {
            const uint32_t me = A.C.xy[g];
            L.pk[b][g - c0] = (me & 0xFFFu) | ((me >> 16) << 12) | ((uint32_t)kUndecided << 24);
            L.resp[b][g - c0] = A.C.resp[g];
            L.key[b][g - c0] = g;
            rc[k].load(A.nbr + 2 * (size_t)g);
          } else {
            rc[k].clear();
          }
        }
      }
      __syncthreads();
      const unsigned long long t_l1 = wall_clock64();
      if (in_lds) {
        if (prev_lds) level<true, true>(A, T, L, l, b, c0, c1, p0, rounds, rc);
        else level<true, false>(A, T, L, l, b, c0, c1, p0, rounds, rc);
      } else {
        if (prev_lds) level<false, true>(A, T, L, l, b, c0, c1, p0, rounds, rc);
        else level<false, false>(A, T, L, l, b, c0, c1, p0, rounds, rc);
      }
      // the previous level is final now: its statuses go back to the global array (the second pass reads them there)
      if (prev_lds)
        for (unsigned int g = p0 + tid; g < p1; g += 1024) {
          A.status[g] = (uint8_t)(L.pk[b ^ 1][g - p0] >> 24);
          A.key[g] = L.key[b ^ 1][g - p0];
        }
      const unsigned long long t_l2 = wall_clock64();
      prev_lds = in_lds;
      p0 = c0;
      p1 = c1;
      __syncthreads();
      if (tid == 0 && A.rounds_out) {  // diagnostics: per level candidates, rounds, 10 ns ticks of the fill and of the rounds
        unsigned int *dbg = A.rounds_out + 8 + 4 * l;
        dbg[0] = c1 - c0;
        dbg[1] = rounds - rounds_before;
        dbg[2] = (unsigned int)(t_l1 - t_l0);
        dbg[3] = (unsigned int)(t_l2 - t_l1);
      }
    }
    if (prev_lds)  // ... and the last level's
      for (unsigned int g = p0 + tid; g < p1; g += 1024) {
        A.status[g] = (uint8_t)(L.pk[(T.n - 1) & 1][g - p0] >> 24);
        A.key[g] = L.key[(T.n - 1) & 1][g - p0];
      }
    __syncthreads();
    const unsigned long long t_pass1 = wall_clock64();
    // ---- second pass: a point with a stronger point of the NEXT level within its radius, further down the list (its N
    // list), is dropped; the survivors that Do_Subpixel_Refinement accepts take their places: slot[key] = the candidate ----
    for (unsigned int g = tid; g < N; g += 1024) {
      if (A.status[g] != kHolds) continue;
      const uint8_t lv = A.C.level[g];
      if (lv & 0x80) continue;  // (rejected by the sub-pixel step whatever the second pass says)
      const int l = lv & 0x7F;
      bool rep = false;
      if (l + 1 < T.n) {
        const unsigned int kg = A.key[g];
        const uint2 r2 = A.nbr2[g];
        if (r2.x & 0x8000u) {  // spilled: the band of rows
          const uint32_t me = A.C.xy[g];
          const float size = T.ksize[l], ratio = (float)(1 << T.octave[l]);
          const float px = (float)(me & 0xFFFFu) * ratio, py = (float)(me >> 16) * ratio, rsp = A.C.resp[g];
          const float rn = (float)(1 << T.octave[l + 1]);
          const int wn = T.w[l + 1], hn = T.h[l + 1];
          const int x0 = max((int)floorf((px - size) / rn) - 1, 0), x1 = min((int)ceilf((px + size) / rn) + 1, wn - 1);
          const int y0 = max((int)floorf((py - size) / rn) - 1, 0), y1 = min((int)ceilf((py + size) / rn) + 1, hn - 1);
          if (x0 <= x1 && y0 <= y1) {
            const unsigned int from = min(A.seg_base[(unsigned int)(T.row0[l + 1] + y0) * A.segs_per_row], N);
            const unsigned int to = min(A.seg_base[(unsigned int)(T.row0[l + 1] + y1 + 1) * A.segs_per_row], N);
            for (unsigned int o = from; o < to; ++o) {
              const uint32_t q = A.C.xy[o];
              const int xo = (int)(q & 0xFFFFu);
              if (xo < x0 || xo > x1 || A.status[o] != kHolds || A.key[o] <= kg) continue;
              const float dx = px - (float)xo * rn, dy = py - (float)(q >> 16) * rn;
              if (dx * dx + dy * dy <= size * size && rsp < A.C.resp[o]) rep = true;
            }
          }
        } else {
          const unsigned int nN = r2.x & 3u, n0 = L.lvl[l + 1];
          const unsigned int e[kNbr2Entries] = {r2.x >> 16, r2.y & 0xFFFFu, r2.y >> 16};
#pragma unroll
          for (int i = 0; i < kNbr2Entries; ++i) {
            const unsigned int o = (unsigned int)i < nN ? n0 + e[i] : g;  // (own: holds, key not above its own)
            if (A.status[o] == kHolds && A.key[o] > kg) rep = true;
          }
        }
      }
      if (!rep) A.slot[A.key[g]] = (int)g;
    }
    __syncthreads();
    const unsigned long long t_pass2 = wall_clock64();
    // ---- ordered compaction over the keys: a wave per contiguous chunk, as k_seg_scan ----
    const int lane = tid & 63, wv = tid >> 6;
    const unsigned int chunk = ((N + 15u) / 16u + 63u) & ~63u;
    const unsigned int lo = min(N, (unsigned int)wv * chunk), hi = min(N, lo + chunk);
    unsigned int sum = 0;
    for (unsigned int k = lo + lane; k < hi; k += 64) sum += A.slot[k] >= 0 ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) L.wsum[wv] = sum;
    __syncthreads();
    unsigned int carry = 0;
    for (int q = 0; q < wv; ++q) carry += L.wsum[q];
    for (unsigned int k0 = lo; k0 < hi; k0 += 64) {
      const unsigned int k = k0 + lane;
      const int g = k < hi ? A.slot[k] : -1;
      const unsigned long long m = __ballot(g >= 0);
      if (g >= 0) {
        const unsigned int pos = carry + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        const int l = A.C.level[g] & 0x7F;
        const uint32_t me = A.C.xy[g];
        const float ratio = (float)(1 << T.octave[l]);
        const float2 d = A.C.sub[g];
        A.kp[4 * pos] = ((float)(me & 0xFFFFu) + d.x) * ratio;
        A.kp[4 * pos + 1] = ((float)(me >> 16) + d.y) * ratio;
        A.kp[4 * pos + 2] = T.ksize[l] * 2.0f;
        A.kp[4 * pos + 3] = (float)l;
        A.resp_out[pos] = A.C.resp[g];
      }
      carry += (unsigned int)__popcll(m);
    }
    if (tid == 0) {
      unsigned int total = 0;
      for (int q = 0; q < 16; ++q) total += L.wsum[q];
      *A.n_kp = total;
      if (A.rounds_out) A.rounds_out[6] = global_levels;
      if (A.rounds_out) {  // diagnostics (SFMLOC_AKAZE_TIMING): rounds, then 10 ns ticks of the first pass, the second, the rest
        A.rounds_out[0] = rounds;
        A.rounds_out[1] = (unsigned int)(t_pass1 - t_start);
        A.rounds_out[2] = (unsigned int)(t_pass2 - t_pass1);
        A.rounds_out[3] = (unsigned int)(wall_clock64() - t_pass2);
      }
    }
  }
};
__global__ __launch_bounds__(1024) void k_suppress(SuppressArgs A) { SuppressBody::run(A); }

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
                                        float *__restrict__ angle_out, uint8_t *__restrict__ desc,
                                        const unsigned int *__restrict__ n_dev /*or null: the count, on the device*/,
                                        const float *__restrict__ resp /*with kp6: the responses*/,
                                        float *__restrict__ kp6 /*or null: out [n x 6] x, y, size, angle, response, class*/,
                                        float2 *__restrict__ qkpt /*or null: out [n] (x, y): a query's keypoints ...*/,
                                        float2 *__restrict__ qkpt6 /*... and the same after the .feat text round trip*/) {
    __shared__ float resX[109], resY[109], Ang[109];
    __shared__ float vals[29 * 3];
    // the M-LDB sample grid of the keypoint: positions (kk, l) in [-10, 10]^2, shared by the three cell grids
    __shared__ float sDI[441], sDX[441], sDY[441];
    __shared__ int8_t disc_i[109], disc_j[109];  // the q-th (i, j) of the orientation disc
    if (threadIdx.x == 0) {
      int cnt = 0;
      for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
          if (i * i + j * j < 36) {
            disc_i[cnt] = (int8_t)i;
            disc_j[cnt] = (int8_t)j;
            ++cnt;
          }
    }
    __syncthreads();
    if (n_dev) n = (int)*n_dev;
    const int lane = threadIdx.x;
    // rows n .. the next multiple of 64 are zeroed: the descriptor array then IS a query block (sfmloc_query_create_view)
    if (qkpt)
      for (int kidx = n + blockIdx.x; kidx < ((n + 63) & ~63); kidx += gridDim.x) desc[(size_t)kidx * 64 + lane] = 0;
    // (a launch sized before the count is known covers any count: workgroup b takes keypoints b, b + gridDim.x, ...)
    for (int kidx = blockIdx.x; kidx < n; kidx += gridDim.x) {
    const float kx = kp[4 * kidx], ky = kp[4 * kidx + 1], ksize = kp[4 * kidx + 2];
    const int level = (int)kp[4 * kidx + 3];
    const DevLevel L = LVp->l[level];
    const float ratio = (float)(1 << L.octave);
    const int s = fround_d(0.5f * ksize / ratio);
    const float xf = kx / ratio, yf = ky / ratio;
    // --- orientation: 109 samples of the disc of radius 6 s ---
    for (int q = lane; q < 109; q += 64) {
      // q-th (i, j) of the double loop i = -6..6, j = -6..6 with i*i + j*j < 36
      const int ii = disc_i[q], jj = disc_j[q];
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
    if (lane == 0) {
      angle_out[kidx] = angle;
      if (kp6) {
        kp6[6 * kidx] = kx;
        kp6[6 * kidx + 1] = ky;
        kp6[6 * kidx + 2] = ksize;
        kp6[6 * kidx + 3] = angle;
        kp6[6 * kidx + 4] = resp[kidx];
        kp6[6 * kidx + 5] = (float)level;
      }
      if (qkpt) {
        qkpt[kidx] = make_float2(kx, ky);
        qkpt6[kidx] = make_float2(geom::round6_dev(kx), geom::round6_dev(ky));
      }
    }
    // --- M-LDB: 4 + 9 + 16 grid cells, one lane per cell, samples summed in (k, l) order ---
    float si, co;
    det_sincosf(angle, &si, &co);
    const int scale = s;
    // Round 3: the three cell grids (2 x 2 cells of 10 x 10 samples, 3 x 3 of 7 x 7, 4 x 4 of 5 x 5) sample the SAME
    // positions (kk, l) of the rotated pattern; a cell's lane used to fetch its samples itself, one after the other --
    // 100 dependent rounds of three global loads for a lane of the coarse grid, 50 us per keypoint whatever the machine
    // does meanwhile.  Now the wave fetches the 21 x 21 positions once, side by side (7 per lane), and a cell's lane sums
    // its samples from LDS -- the same values in the same (kk, l) order.
    for (int t = lane; t < 441; t += 64) {
      const int kk = t / 21 - 10, l = t - (t / 21) * 21 - 10;
      const float sample_y = yf + ((float)l * co * (float)scale + (float)kk * si * (float)scale);
      const float sample_x = xf + (-(float)l * si * (float)scale + (float)kk * co * (float)scale);
      const int y1 = fround_d(sample_y), x1 = fround_d(sample_x);
      const float ri = at_clamped(L.Lt, L.w, L.h, y1, x1);
      const float rx = at_clamped(L.Lx, L.w, L.h, y1, x1) * L.sf;
      const float ry = at_clamped(L.Ly, L.w, L.h, y1, x1) * L.sf;
      const float rry = rx * co + ry * si;
      const float rrx = -rx * si + ry * co;
      sDI[t] = ri;
      sDX[t] = rrx;
      sDY[t] = rry;
    }
    __syncthreads();
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
          const int t = (kk + 10) * 21 + (l + 10);
          di += sDI[t];
          dx += sDX[t];
          dy += sDY[t];
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
    __syncthreads();  // the LDS arrays serve the next keypoint of this workgroup
    }
  }
};
__global__ __launch_bounds__(64) void k_orient_describe(const DevLevels *__restrict__ LVp, const float *__restrict__ kp, int n,
                                        const float *__restrict__ gauss25,
                                        const float *__restrict__ win_ang1, int n_win,
                                        const uint16_t *__restrict__ pair_tab,
                                        float *__restrict__ angle_out, uint8_t *__restrict__ desc,
                                        const unsigned int *__restrict__ n_dev, const float *__restrict__ resp,
                                        float *__restrict__ kp6, float2 *__restrict__ qkpt, float2 *__restrict__ qkpt6) {
  OrientDescribeBody::run(LVp, kp, n, gauss25, win_ang1, n_win, pair_tab, angle_out, desc, n_dev, resp, kp6, qkpt, qkpt6);
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
  // detection tail (k_extrema_seg .. k_suppress): the raster-ordered candidate list and the suppression's state
  unsigned int *d_ncand = nullptr;            // [0] candidates, [1] keypoints, [2] rounds of the first pass, [3..5] clocks,
                                              // [6] spilled A/P records, [7] spilled N records, [8] levels out of global arrays
  unsigned int cand_cap = 1u << 16;
  unsigned int n_seg = 0, segs_per_row = 0;
  unsigned int *d_seg = nullptr;              // [n_seg + 1] counts, then (k_seg_scan) where each segment starts
  unsigned int *d_seg_part = nullptr;         // [chunks + 1] k_seg_sum / k_seg_write: the chunks' sums, then arrivals
  uint8_t *d_seg_x = nullptr;                 // [n_seg x kSegMax]
  uint32_t *d_cxy = nullptr;
  uint8_t *d_clevel = nullptr, *d_cstatus = nullptr, *d_cready = nullptr;
  float *d_cresp = nullptr;
  float2 *d_csub = nullptr;
  unsigned int sup_force_spill = 0, sup_force_global = 0;  // test hooks (SFMLOC_AKAZE_SUPPRESS, read when the extractor is made)
  uint4 *d_nbr = nullptr;                     // [cand_cap x 2] k_suppress_nbr: the A and P lists
  uint2 *d_nbr2 = nullptr;                    // [cand_cap] the N lists
  unsigned int *d_ckey = nullptr;
  int *d_cslot = nullptr;
  float *d_resp = nullptr, *d_kp6 = nullptr;  // per keypoint: response; the six-float records of the C ABI
  float2 *d_qkpt = nullptr, *d_qkpt6 = nullptr;  // the keypoints as a query holds them: (x, y), and after the .feat round trip
  unsigned int *h_counts = nullptr;           // pinned copy of d_ncand
  // sfmloc_akaze_detect_and_compute reads the counts and -- in the same transfer batch, before it knows the count -- the
  // first spec_n keypoints and descriptors into pinned memory: an image with no more keypoints than that costs ONE
  // synchronisation instead of two (~45 us); spec_n follows the last image's count
  unsigned char *h_spec = nullptr;
  unsigned int spec_n = 512;
  static constexpr unsigned int kSpecMax = 4096;
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
    T.ksize[i] = L.esigma * 1.5f;
    T.octave[i] = L.octave;
    row += L.h;
  }
  T.row0[P.nlev] = row;
  return T;
}

// Create_Nonlinear_Scale_Space + Compute_Multiscale_Derivatives + Compute_Determinant_Hessian_Response
// need: the levels [0, need) are built (0 = all of them) -- a description-only call builds no level above its keypoints'
// highest (the dense BoW grid lives on levels 0 .. 3: three of the four octaves used to be evolved for nothing) and no
// Hessian determinant (with_det = false); every level that IS built has the bits it always had.
int build_scale_space(Akaze *a, const uint8_t *gray /*host; null: the image is already in d_gray, written on a->stream*/,
                      int need = 0, bool with_det = true) {
  const AkPlan &P = a->plan;
  const int nlev = (need > 0 && need < P.nlev) ? need : P.nlev;
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
    {
      unsigned int hy = ggrid.y;
      if (ggrid.x * hy > 256u) hy = std::max(1u, 256u / ggrid.x);
      sfm_launch<PreHistBody>(a, k_pre_hist, dim3(ggrid.x, hy), dim3(128), (uint32_t)0, a->d_t2, w, h, a->d_hist, a->d_kcontrast);
    }
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
  for (int i = 1; i < nlev; ++i) {
    const AkLevel &L = P.lev[i], &Lp = P.lev[i - 1];
    float *Lt = a->d_Lt + L.off;
    // an octave whose image fits in LDS three times runs in one launch (k_octave_resident)
    {
      int j = i;
      while (j + 1 < nlev && P.lev[j + 1].octave == L.octave) ++j;
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
    // (the 240 x 135 octave of a 1080p frame was tried with the small levels' count: no effect, 1.10 / 1.23 ms either way)
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
  // (the rows of the levels that were built: levels come in order; kAllPix pixels of a row per thread)
  const dim3 agrid(((a->w + 127) / 128 + kAllPix - 1) / kAllPix, T.row0[nlev]);
  sfm_launch<ScharrXyAllBody>(a, k_scharr_xy_all, agrid, dim3(128), (uint32_t)0, a->d_Lsmooth, a->d_Lx, a->d_Ly, reinterpret_cast<const LevelTab *>(a->d_level_tab));
  if (with_det)
    sfm_launch<HessianDetAllBody>(a, k_hessian_det_all, agrid, dim3(128), (uint32_t)0, a->d_Lx, a->d_Ly, a->d_Ldet, reinterpret_cast<const LevelTab *>(a->d_level_tab));
  AK_HIP(hipGetLastError());
  return SFMLOC_OK;
#undef s
}

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
  if (a->d_resp) hipFree(a->d_resp);
  if (a->d_kp6) hipFree(a->d_kp6);
  a->d_resp = a->d_kp6 = nullptr;
  if (a->d_qkpt) hipFree(a->d_qkpt);
  if (a->d_qkpt6) hipFree(a->d_qkpt6);
  a->d_qkpt = a->d_qkpt6 = nullptr;
  AK_HIP(hipMalloc((void **)&a->d_qkpt, (size_t)cap * sizeof(float2)));
  AK_HIP(hipMalloc((void **)&a->d_qkpt6, (size_t)cap * sizeof(float2)));
  AK_HIP(hipMalloc((void **)&a->d_kp, (size_t)cap * 4 * sizeof(float)));
  AK_HIP(hipMalloc((void **)&a->d_angle, (size_t)cap * sizeof(float)));
  AK_HIP(hipMalloc((void **)&a->d_desc, ((size_t)cap + 64) * 64));  // (+ the zero rows up to a multiple of 64)
  AK_HIP(hipMalloc((void **)&a->d_resp, (size_t)cap * sizeof(float)));
  AK_HIP(hipMalloc((void **)&a->d_kp6, (size_t)cap * 6 * sizeof(float)));
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
                                   a->plan.n_win, a->d_pair, a->d_angle, a->d_desc,
                                   (const unsigned int *)nullptr, (const float *)nullptr, (float *)nullptr,
                                   (float2 *)nullptr, (float2 *)nullptr);
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
// The extractor's stream for work queued NOW: its own is created at the first call that needs one.  An extractor that
// has worked in gang sessions before (on the leader's stream) must not overtake that work on its new stream.
static int akaze_ensure_stream(Akaze *a) {
  if (a->stream.own || a->stream.gang) return SFMLOC_OK;  // (recording for a session: the leader's stream carries it)
  if (!a->own_stream) {
    SFM_HIP(hipSetDevice(a->device));
    SFM_HIP(hipStreamCreateWithFlags(&a->own_stream, hipStreamNonBlocking));
    if (a->ever_ganged) SFM_HIP(hipDeviceSynchronize());
  }
  a->stream.own = a->own_stream;
  return SFMLOC_OK;
}
GangMember *akaze_member(Akaze *a) { return a; }
hipStream_t akaze_stream_now(Akaze *a) {
  (void)akaze_ensure_stream(a);
  return a->stream;
}
int akaze_compute_resident(Akaze *a, const float *d_kin, unsigned int n, int need_levels) {
  int rc = akaze_ensure_stream(a);
  if (rc) return rc;
  rc = ensure_kp_cap(a, n);  // (before anything is queued: it may free and allocate)
  if (rc) return rc;
  rc = build_scale_space(a, nullptr, need_levels, false);
  if (rc || n == 0) return rc;
  sfm_launch<OrientDescribeBody>(a, k_orient_describe, dim3(n), dim3(64), 0,
                                 reinterpret_cast<const DevLevels *>(a->d_dev_levels), d_kin, (int)n, a->d_gauss25, a->d_win,
                                 a->plan.n_win, a->d_pair, a->d_angle, a->d_desc,
                                   (const unsigned int *)nullptr, (const float *)nullptr, (float *)nullptr,
                                   (float2 *)nullptr, (float2 *)nullptr);
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
                  a->d_Ldet, a->d_hist, a->d_kcontrast, a->d_half_steps, a->d_level_tab, a->d_dev_levels, a->d_ncand,
                  a->d_gauss25, a->d_win, a->d_pair, a->d_kp, a->d_angle, a->d_desc, a->d_seg, a->d_seg_part, a->d_seg_x, a->d_cxy,
                  a->d_clevel, a->d_cstatus, a->d_cready, a->d_cresp, a->d_csub, a->d_nbr, a->d_nbr2, a->d_ckey, a->d_cslot, a->d_resp, a->d_kp6, a->d_qkpt, a->d_qkpt6};
  for (void *p : ptrs)
    if (p) hipFree(p);
  if (a->h_counts) hipHostFree(a->h_counts);
  if (a->h_spec) hipHostFree(a->h_spec);
  if (a->own_stream) hipStreamDestroy(a->own_stream);
  delete a;
}

int sfmloc_akaze_create(int device, int width, int height, int n_octaves, int n_sublevels, float threshold,
                        sfmloc_akaze **out) {
  SFM_CHECK(out, SFMLOC_EINVAL, "sfmloc_akaze_create: null argument");
  *out = nullptr;
  SFM_CHECK(width >= 16 && height >= 16 && width <= 16384 && height <= kSupMaxRows - 1, SFMLOC_EINVAL,
            "sfmloc_akaze_create: image size %dx%d (at most 16384 x %d)", width, height, kSupMaxRows - 1);
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
  // (no stream yet: akaze_ensure_stream creates one at the first call that needs it -- an extractor that is given a
  // context's stream right away, or only ever works in sessions led by another, never needs one, and a stream that merely
  // EXISTS shares a hardware queue with streams that work: DESIGN.md 4 Concurrency)
  hipError_t he = hipSuccess;
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
  {
    const LevelTab T0 = level_tab(a);
    a->segs_per_row = 2u * (unsigned int)((width + 127) / 128);
    a->n_seg = (unsigned int)T0.row0[T0.n] * a->segs_per_row;
  }
  const size_t cc = a->cand_cap;
  A((void **)&a->d_ncand, 160 * sizeof(unsigned int));
  A((void **)&a->d_seg, ((size_t)a->n_seg + 1) * sizeof(unsigned int));
  A((void **)&a->d_seg_part, 1025 * sizeof(unsigned int));
  if (he == hipSuccess) he = hipMemset(a->d_seg_part, 0, 1025 * sizeof(unsigned int));
  if (he == hipSuccess) he = hipStreamSynchronize(nullptr);  // (null stream: the extractor's streams are not ordered with it)
  A((void **)&a->d_seg_x, (size_t)a->n_seg * kSegMax);
  A((void **)&a->d_cxy, cc * sizeof(uint32_t));
  A((void **)&a->d_clevel, cc);
  A((void **)&a->d_cstatus, cc);
  A((void **)&a->d_cready, cc);
  A((void **)&a->d_cresp, cc * sizeof(float));
  A((void **)&a->d_csub, cc * sizeof(float2));
  A((void **)&a->d_nbr, cc * 2 * sizeof(uint4));
  A((void **)&a->d_nbr2, cc * sizeof(uint2));
  if (const char *e = getenv("SFMLOC_AKAZE_SUPPRESS")) {  // test hooks: "spill" = every record spilled (the band scans
    a->sup_force_spill = strstr(e, "spill") != nullptr;    // decide everything), "global" = no level in LDS
    a->sup_force_global = strstr(e, "global") != nullptr;
  }
  A((void **)&a->d_ckey, cc * sizeof(unsigned int));
  A((void **)&a->d_cslot, cc * sizeof(int));
  if (he == hipSuccess) he = hipHostMalloc((void **)&a->h_counts, 160 * sizeof(unsigned int), hipHostMallocDefault);
  if (he == hipSuccess) he = hipHostMalloc((void **)&a->h_spec, (size_t)Akaze::kSpecMax * (6 * sizeof(float) + 64), hipHostMallocDefault);
  if (he == hipSuccess && ensure_kp_cap(a, a->cand_cap) != SFMLOC_OK) he = hipErrorOutOfMemory;  // (never regrown later)
  if (he == hipSuccess) {  // (k_suppress keeps two levels' candidates in LDS: more than the default 64 KB per workgroup)
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(k_suppress),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SuppressLds));
    he = attr;
  }
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
  if (a->stream.own) hipStreamSynchronize(a->stream);
  a->stream.own = c ? c->stream.own : a->own_stream;  // (nullptr: created when first needed, akaze_ensure_stream)
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
  if (a->stream.own) SFM_HIP(hipStreamSynchronize(a->stream));
  if (ldet) SFM_HIP(hipMemcpy(ldet, a->d_Ldet, a->plan.total * sizeof(float), hipMemcpyDeviceToHost));
  if (lt) SFM_HIP(hipMemcpy(lt, a->d_Lt, a->plan.total * sizeof(float), hipMemcpyDeviceToHost));
  return SFMLOC_OK;
}

int sfmloc_akaze_suppress_stats(sfmloc_akaze *ak, uint32_t *out9) {
  SFM_CHECK(ak && out9, SFMLOC_EINVAL, "sfmloc_akaze_suppress_stats: null argument");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  SFM_HIP(hipSetDevice(a->device));
  if (a->stream.own) SFM_HIP(hipStreamSynchronize(a->stream));
  SFM_HIP(hipMemcpy(out9, a->d_ncand, 9 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return SFMLOC_OK;
}

// The whole of detectAndCompute on the device, launches only (an extractor that records for a gang session takes part in
// ONE launch per kernel with the other images of the batch): scale space -> extrema, raster ordered -> duplicate
// suppression + sub-pixel refinement (k_suppress) -> orientation + M-LDB of however many keypoints that left (the launch
// is sized before the count is known: its workgroups stride over the keypoints).  No host synchronisation, nothing
// leaves the device: d_ncand[1] keypoints, d_kp6 their six-float records, d_desc their .desc rows.
static int akaze_detect_enqueue(Akaze *a, const uint8_t *gray) {
  int rc = build_scale_space(a, gray);
  if (rc) return rc;
  const LevelTab T = level_tab(a);
  const dim3 egrid(((a->w + 127) / 128 + kSegPerWave - 1) / kSegPerWave, T.row0[T.n]);  // (kSegPerWave segments per wave)
  const LevelTab *dT = reinterpret_cast<const LevelTab *>(a->d_level_tab);
  CandArrays C{a->d_cxy, a->d_clevel, a->d_cresp, a->d_csub};
  sfm_launch<ExtremaSegBody>(a, k_extrema_seg, egrid, dim3(128), 0, a->d_Ldet, dT, a->thres, a->d_seg_x, a->d_seg, a->segs_per_row);
  {
    const unsigned int chunks = (a->n_seg + kSegChunk - 1u) / kSegChunk;
    if (a->n_seg > 65536u && chunks <= 1024u && a->d_seg_part) {  // a large image: two launches over many compute units
      sfm_launch<SegSumBody>(a, k_seg_sum, dim3(chunks), dim3(1024), 0, a->d_seg, a->n_seg, a->d_seg_part, a->d_ncand);
      sfm_launch<SegWriteBody>(a, k_seg_write, dim3(chunks), dim3(1024), 0, a->d_seg, a->n_seg, (const unsigned int *)a->d_seg_part);
    } else {
      sfm_launch<SegScanBody>(a, k_seg_scan, dim3(1), dim3(1024), 0, a->d_seg, a->n_seg, a->cand_cap, a->d_ncand);
    }
  }
  sfm_launch<ExtremaPlaceBody>(a, k_extrema_place, egrid, dim3(128), 0, a->d_Ldet, dT, a->d_seg_x, a->d_seg, a->cand_cap, C,
                               a->segs_per_row);
  {  // the neighbour lists: one wave per candidate, a fixed grid that strides over them (their number is on the device)
    unsigned int ngrid = (unsigned int)((size_t)a->w * a->h / 384);
    ngrid = ngrid < 256u ? 256u : (ngrid > 4096u ? 4096u : ngrid);
    sfm_launch<NbrBody>(a, k_suppress_nbr, dim3(ngrid), dim3(256), 0, dT, (const unsigned int *)a->d_seg, a->segs_per_row,
                        a->cand_cap, (const unsigned int *)a->d_ncand, (const uint32_t *)a->d_cxy, (const uint8_t *)a->d_clevel,
                        (const float *)a->d_cresp, a->d_nbr, a->d_nbr2, a->d_ncand + 6, a->sup_force_spill);
  }
  SuppressArgs S;
  S.T = dT;
  S.nbr = a->d_nbr;
  S.nbr2 = a->d_nbr2;
  S.force_global = a->sup_force_global;
  S.seg_base = a->d_seg;
  S.segs_per_row = a->segs_per_row;
  S.cap = a->cand_cap;
  S.C = C;
  S.status = a->d_cstatus;
  S.ready = a->d_cready;
  S.key = a->d_ckey;
  S.slot = a->d_cslot;
  S.kp = a->d_kp;
  S.resp_out = a->d_resp;
  S.n_kp = a->d_ncand + 1;
  S.n_cand = a->d_ncand;
  S.rounds_out = a->d_ncand + 2;
  sfm_launch<SuppressBody>(a, k_suppress, dim3(1), dim3(1024), (uint32_t)sizeof(SuppressLds), S);
  // orientation + M-LDB: a fixed grid that strides over the keypoints (their number is on the device)
  unsigned int dgrid = (unsigned int)((size_t)a->w * a->h / 96);
  dgrid = dgrid < 1024u ? 1024u : (dgrid > 16384u ? 16384u : dgrid);
  sfm_launch<OrientDescribeBody>(a, k_orient_describe, dim3(dgrid), dim3(64), 0,
                                 reinterpret_cast<const DevLevels *>(a->d_dev_levels), a->d_kp, 0, a->d_gauss25, a->d_win,
                                 a->plan.n_win, a->d_pair, a->d_angle, a->d_desc, (const unsigned int *)(a->d_ncand + 1),
                                 (const float *)a->d_resp, a->d_kp6, a->d_qkpt, a->d_qkpt6);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// the counts of one or more extractions (asynchronous on stream s; the caller synchronises), then their outputs
static int akaze_counts_enqueue(Akaze *a, hipStream_t s) {
  static const bool timing_counts = getenv("SFMLOC_AKAZE_TIMING") != nullptr;
  SFM_HIP(hipMemcpyAsync(a->h_counts, a->d_ncand, (timing_counts ? 160 : 9) * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  return SFMLOC_OK;
}
static int akaze_outputs_enqueue(Akaze *a, float *kpts, uint8_t *desc64, uint32_t cap, uint32_t *n_out, hipStream_t s) {
  const unsigned int nc = a->h_counts[0], n = a->h_counts[1];
  SFM_CHECK(nc <= a->cand_cap, SFMLOC_ECAP, "AKAZE: %u extrema candidates exceed the workspace (%u)", nc, a->cand_cap);
  *n_out = n;
  SFM_CHECK(n <= cap, SFMLOC_ECAP, "AKAZE: %u keypoints, caller buffers hold %u", n, cap);
  if (n && kpts) SFM_HIP(hipMemcpyAsync(kpts, a->d_kp6, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, s));
  if (n && desc64) SFM_HIP(hipMemcpyAsync(desc64, a->d_desc, (size_t)n * 64, hipMemcpyDeviceToHost, s));
  return SFMLOC_OK;
}

int sfmloc_akaze_detect_and_compute(sfmloc_akaze *ak, const uint8_t *gray, float *kpts /*[cap*6]*/,
                                    uint8_t *desc64 /*[cap*64]*/, uint32_t cap, uint32_t *n_out) {
  SFM_CHECK(ak && gray && n_out, SFMLOC_EINVAL, "sfmloc_akaze_detect_and_compute: null argument");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  SFM_HIP(hipSetDevice(a->device));
  static const bool timing = getenv("SFMLOC_AKAZE_TIMING") != nullptr;
  const double t_0 = timing ? now_s() : 0.0;
  int rc = akaze_ensure_stream(a);
  if (rc) return rc;
  rc = akaze_detect_enqueue(a, gray);
  if (rc) return rc;
  hipStream_t s = a->stream;
  rc = akaze_counts_enqueue(a, s);
  if (rc) return rc;
  // (speculative: the first spec_n records with the counts -- see Akaze::h_spec)
  const unsigned int spec = a->spec_n < cap ? a->spec_n : cap;
  float *const h_kp = reinterpret_cast<float *>(a->h_spec);
  uint8_t *const h_desc = a->h_spec + (size_t)Akaze::kSpecMax * 6 * sizeof(float);
  if (spec && kpts) SFM_HIP(hipMemcpyAsync(h_kp, a->d_kp6, (size_t)spec * 6 * sizeof(float), hipMemcpyDeviceToHost, s));
  if (spec && desc64) SFM_HIP(hipMemcpyAsync(h_desc, a->d_desc, (size_t)spec * 64, hipMemcpyDeviceToHost, s));
  SFM_HIP(hipStreamSynchronize(s));
  const double t_1 = timing ? now_s() : 0.0;
  {
    const unsigned int n = a->h_counts[1];
    unsigned int next = ((n + n / 4 + 255u) / 256u) * 256u;  // a quarter of headroom, in steps of 256
    a->spec_n = next < 256u ? 256u : (next > Akaze::kSpecMax ? Akaze::kSpecMax : next);
    if (a->h_counts[0] <= a->cand_cap && n <= cap && n <= spec) {  // everything is here already
      *n_out = n;
      if (n && kpts) memcpy(kpts, h_kp, (size_t)n * 6 * sizeof(float));
      if (n && desc64) memcpy(desc64, h_desc, (size_t)n * 64);
    } else {
      rc = akaze_outputs_enqueue(a, kpts, desc64, cap, n_out, s);
      if (rc) return rc;
      SFM_HIP(hipStreamSynchronize(s));
    }
  }
  if (timing)
    fprintf(stderr, "akaze %dx%d: device (scale space .. descriptors) %.3f ms, outputs %.3f ms (%u candidates -> %u keypoints, "
                    "%u suppression rounds: first pass %.1f us, second %.1f, sub-pixel + compaction %.1f)\n", a->w, a->h,
            (t_1 - t_0) * 1e3, (now_s() - t_1) * 1e3, a->h_counts[0], a->h_counts[1], a->h_counts[2], a->h_counts[3] * 0.01,
            a->h_counts[4] * 0.01, a->h_counts[5] * 0.01);
  if (timing) {
    for (int l = 0; l < a->plan.nlev; ++l)
      fprintf(stderr, "  level %2d: %5u candidates, %2u rounds, fill %.1f us, rounds %.1f us (a %.1f b %.1f)\n", l, a->h_counts[10 + 4 * l],
              a->h_counts[11 + 4 * l], a->h_counts[12 + 4 * l] * 0.01, a->h_counts[13 + 4 * l] * 0.01, a->h_counts[82 + 2 * l] * 0.01, a->h_counts[83 + 2 * l] * 0.01);
  }
  return SFMLOC_OK;
}

// Several images of one size at once: extractor i takes image i; every kernel of the chain goes out as ONE launch for all
// of them (a gang session on the first extractor's stream), then one synchronisation for the counts and one for the
// outputs.  Results are those of n separate calls.
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
  int rc = akaze_ensure_stream(first);
  if (rc) return rc;
  rc = n > 1 ? gang_open(ms, (int)n) : SFMLOC_OK;
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) rc = akaze_detect_enqueue(reinterpret_cast<Akaze *>(aks[i]), grays[i]);
  const int rc_close = gang_close(first);
  if (rc == SFMLOC_OK) rc = rc_close;
  if (rc) return rc;
  const double t_1 = timing ? now_s() : 0.0;
  hipStream_t s0 = first->stream;  // (the gang's stream: everything of every image was produced on it)
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) rc = akaze_counts_enqueue(reinterpret_cast<Akaze *>(aks[i]), s0);
  if (rc) return rc;
  SFM_HIP(hipStreamSynchronize(s0));
  const double t_2 = timing ? now_s() : 0.0;
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i)
    rc = akaze_outputs_enqueue(reinterpret_cast<Akaze *>(aks[i]), kpts[i], descs ? descs[i] : nullptr, cap, &n_out[i], s0);
  if (rc) return rc;
  SFM_HIP(hipStreamSynchronize(s0));
  // the members' own streams were made to wait for the session (gang_close); nothing of theirs is pending beyond it
  if (timing)
    fprintf(stderr, "akaze batch of %u (%dx%d): launches queued %.3f ms, device %.3f, outputs %.3f\n", n, first->w, first->h,
            (t_1 - t_0) * 1e3, (t_2 - t_1) * 1e3, (now_s() - t_2) * 1e3);
  return SFMLOC_OK;
}


// detectAndCompute whose outputs STAY on the device, laid out as a query block: descriptors [n x 64] followed by zero
// rows up to a multiple of 64, keypoints (x, y) and the keypoints after the .feat text round trip -- what
// sfmloc_query_create_view takes.  Only the count comes back (one 4-byte copy + one synchronisation per call).
int sfmloc_akaze_detect_resident_batch(sfmloc_akaze *const *aks, const uint8_t *const *grays, uint32_t n, uint32_t *n_out) {
  SFM_CHECK(aks && grays && n_out && n >= 1 && n <= (uint32_t)kGangMembers, SFMLOC_EINVAL,
            "sfmloc_akaze_detect_resident_batch: 1..%d images", kGangMembers);
  GangMember *ms[kGangMembers];
  Akaze *first = reinterpret_cast<Akaze *>(aks[0]);
  for (uint32_t i = 0; i < n; ++i) {
    Akaze *a = reinterpret_cast<Akaze *>(aks[i]);
    SFM_CHECK(a && grays[i], SFMLOC_EINVAL, "sfmloc_akaze_detect_resident_batch: null argument (image %u)", i);
    SFM_CHECK(a->device == first->device && a->w == first->w && a->h == first->h, SFMLOC_EINVAL,
              "sfmloc_akaze_detect_resident_batch: the extractors differ in device or image size");
    SFM_CHECK(a->stream.gang == nullptr, SFMLOC_EINVAL, "sfmloc_akaze_detect_resident_batch: extractor %u is in a session", i);
    for (uint32_t j = 0; j < i; ++j) SFM_CHECK(aks[j] != aks[i], SFMLOC_EINVAL, "extractor listed twice");
    ms[i] = a;
  }
  SFM_HIP(hipSetDevice(first->device));
  int rc = akaze_ensure_stream(first);  // (the session runs on the first extractor's stream; the others need none)
  if (rc) return rc;
  rc = n > 1 ? gang_open(ms, (int)n) : SFMLOC_OK;
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) rc = akaze_detect_enqueue(reinterpret_cast<Akaze *>(aks[i]), grays[i]);
  const int rc_close = n > 1 ? gang_close(first) : SFMLOC_OK;
  if (rc == SFMLOC_OK) rc = rc_close;
  if (rc) return rc;
  hipStream_t s0 = first->stream;
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) rc = akaze_counts_enqueue(reinterpret_cast<Akaze *>(aks[i]), s0);
  if (rc) return rc;
  SFM_HIP(hipStreamSynchronize(s0));
  for (uint32_t i = 0; i < n; ++i) {
    Akaze *a = reinterpret_cast<Akaze *>(aks[i]);
    SFM_CHECK(a->h_counts[0] <= a->cand_cap, SFMLOC_ECAP, "AKAZE: %u extrema candidates exceed the workspace (%u)",
              a->h_counts[0], a->cand_cap);
    n_out[i] = a->h_counts[1];
  }
  return SFMLOC_OK;
}

int sfmloc_akaze_detect_resident(sfmloc_akaze *ak, const uint8_t *gray, uint32_t *n_out) {
  return sfmloc_akaze_detect_resident_batch(&ak, &gray, 1, n_out);
}

int sfmloc_akaze_resident_arrays(sfmloc_akaze *ak, const void **desc_dev, const void **kpt_dev, const void **kpt6_dev,
                                 const void **kp6_dev) {
  SFM_CHECK(ak, SFMLOC_EINVAL, "sfmloc_akaze_resident_arrays: null argument");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  if (desc_dev) *desc_dev = a->d_desc;
  if (kpt_dev) *kpt_dev = a->d_qkpt;
  if (kpt6_dev) *kpt6_dev = a->d_qkpt6;
  if (kp6_dev) *kp6_dev = a->d_kp6;
  return SFMLOC_OK;
}

int sfmloc_akaze_compute(sfmloc_akaze *ak, const uint8_t *gray, const float *kin, uint32_t n, uint8_t *desc64,
                         float *angle_out) {
  SFM_CHECK(ak && gray && (n == 0 || (kin && desc64)), SFMLOC_EINVAL, "sfmloc_akaze_compute: null argument");
  Akaze *a = reinterpret_cast<Akaze *>(ak);
  SFM_HIP(hipSetDevice(a->device));
  int rc = akaze_ensure_stream(a);
  if (rc) return rc;
  std::vector<float> k(kin, kin + (size_t)n * 4);
  int top = 0;
  for (uint32_t i = 0; i < n; ++i) {
    int lvl = (int)k[4 * i + 3];
    lvl = lvl < 0 ? 0 : (lvl >= a->plan.nlev ? a->plan.nlev - 1 : lvl);
    k[4 * i + 3] = (float)lvl;
    top = lvl > top ? lvl : top;
  }
  rc = build_scale_space(a, gray, top + 1, false);  // (no level above the keypoints' highest, no determinant)
  if (rc) return rc;
  return orient_describe(a, k, n, angle_out, desc64);
}

}  // extern "C"
