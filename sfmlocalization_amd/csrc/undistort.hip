// The server's per-user image undistortion in front of LocalizeEngine::localize
// (VisionLocalizeServer/src/localizeImage.cc:149-177):
//
//   newCameraMat = getOptimalNewCameraMatrix(K, dist, size, 1.0, size, &validRoi);         once per user camera
//   undistort(image, undistortImage, K, dist, newCameraMat); undistortImage = undistortImage(validRoi).clone();
//
// sfmloc_undistorter_create does the first line and builds cv::undistort's fixed-point maps on the host (doubles, the
// column-by-column accumulation of the source's inner loop); sfmloc_undistorter_apply is the per-image part on the
// GPU: one thread per output pixel of the valid region, 15-bit fixed-point bilinear remap with a zero border.
// Restated from OpenCV 3.0's algorithms (undistort.cpp, imgwarp.cpp, calibration.cpp); OpenCV is not in this image, so
// parity with it is unpinned -- the test oracle (oracle/oracle_undistort.py) is an independent NumPy restatement.
#include <cmath>
#include <cstring>
#include <vector>

#include "sfmloc_internal.h"

namespace sfmloc {
namespace {

constexpr int kInterBits = 5, kInterTab = 1 << kInterBits;

struct Undistorter {
  int device = 0;
  uint32_t w = 0, h = 0;
  double new_camera[9];
  int32_t roi[4];               // x y w h
  std::vector<int16_t> h_xy;    // [h*w*2] integer source coordinates
  std::vector<uint16_t> h_frac; // [h*w] (fy << 5) | fx
  short2 *d_xy = nullptr;
  uint16_t *d_frac = nullptr;
  uint8_t *d_src = nullptr, *d_dst = nullptr;
  size_t src_cap = 0, dst_cap = 0;
  hipStream_t stream = nullptr;
};

void dist8(const double *dist, uint32_t n, double *k) {
  for (int i = 0; i < 8; ++i) k[i] = (uint32_t)i < n ? dist[i] : 0.0;
}

// cvUndistortPoints, R = I: five fixed-point iterations in double, the result stored as float
void undistort_points(const float *pts, int n, const double *K, const double *k, const double *P, float *out) {
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const double ifx = 1.0 / fx, ify = 1.0 / fy;
  for (int i = 0; i < n; ++i) {
    double x = ((double)pts[2 * i] - cx) * ifx, y = ((double)pts[2 * i + 1] - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
      const double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
      x = (x0 - dx) * icdist;
      y = (y0 - dy) * icdist;
    }
    const double xx = P[0] * x + P[1] * y + P[2], yy = P[3] * x + P[4] * y + P[5];
    const double ww = 1.0 / (P[6] * x + P[7] * y + P[8]);
    out[2 * i] = (float)(xx * ww);
    out[2 * i + 1] = (float)(yy * ww);
  }
}

// icvGetRectangles: 9x9 grid -> inner / outer rectangle (x, y, w, h), float arithmetic as in the source
void get_rectangles(const double *K, const double *k, const double *P, int w, int h, float *inner, float *outer) {
  constexpr int N = 9;
  float pts[N * N * 2], u[N * N * 2];
  for (int y = 0; y < N; ++y)
    for (int x = 0; x < N; ++x) {
      pts[2 * (y * N + x)] = (float)x * (float)w / (float)(N - 1);
      pts[2 * (y * N + x) + 1] = (float)y * (float)h / (float)(N - 1);
    }
  undistort_points(pts, N * N, K, k, P, u);
  const float big = 3.402823466e+38f;
  float iX0 = -big, iX1 = big, iY0 = -big, iY1 = big, oX0 = big, oX1 = -big, oY0 = big, oY1 = -big;
  for (int y = 0; y < N; ++y)
    for (int x = 0; x < N; ++x) {
      const float px = u[2 * (y * N + x)], py = u[2 * (y * N + x) + 1];
      oX0 = std::fmin(oX0, px), oX1 = std::fmax(oX1, px), oY0 = std::fmin(oY0, py), oY1 = std::fmax(oY1, py);
      if (x == 0) iX0 = std::fmax(iX0, px);
      if (x == N - 1) iX1 = std::fmin(iX1, px);
      if (y == 0) iY0 = std::fmax(iY0, py);
      if (y == N - 1) iY1 = std::fmin(iY1, py);
    }
  inner[0] = iX0, inner[1] = iY0, inner[2] = iX1 - iX0, inner[3] = iY1 - iY0;
  outer[0] = oX0, outer[1] = oY0, outer[2] = oX1 - oX0, outer[3] = oY1 - oY0;
}

// cvGetOptimalNewCameraMatrix(K, dist, size, alpha, size, &validRoi, centerPrincipalPoint = 0)
void optimal_new_camera(const double *K, const double *k, int w, int h, double alpha, double *M, int32_t *roi) {
  float inner[4], outer[4];
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  get_rectangles(K, k, I3, w, h, inner, outer);  // in normalised coordinates: M maps that rectangle onto the viewport
  const double fx0 = (double)((float)(w - 1) / inner[2]), fy0 = (double)((float)(h - 1) / inner[3]);
  const double cx0 = -fx0 * (double)inner[0], cy0 = -fy0 * (double)inner[1];
  const double fx1 = (double)((float)(w - 1) / outer[2]), fy1 = (double)((float)(h - 1) / outer[3]);
  const double cx1 = -fx1 * (double)outer[0], cy1 = -fy1 * (double)outer[1];
  for (int i = 0; i < 9; ++i) M[i] = 0.0;
  M[0] = fx0 * (1 - alpha) + fx1 * alpha;
  M[4] = fy0 * (1 - alpha) + fy1 * alpha;
  M[2] = cx0 * (1 - alpha) + cx1 * alpha;
  M[5] = cy0 * (1 - alpha) + cy1 * alpha;
  M[8] = 1.0;
  get_rectangles(K, k, M, w, h, inner, outer);
  const long rx = std::lrint((double)inner[0]), ry = std::lrint((double)inner[1]);  // cv::Rect r = inner
  const long rw = std::lrint((double)inner[2]), rh = std::lrint((double)inner[3]);
  const long x0 = std::max(rx, 0L), y0 = std::max(ry, 0L), x1 = std::min(rx + rw, (long)w), y1 = std::min(ry + rh, (long)h);
  if (x1 > x0 && y1 > y0)
    roi[0] = (int32_t)x0, roi[1] = (int32_t)y0, roi[2] = (int32_t)(x1 - x0), roi[3] = (int32_t)(y1 - y0);
  else
    roi[0] = roi[1] = roi[2] = roi[3] = 0;
}

// cv::invert of a 3x3 double matrix (OpenCV's closed form for n <= 3)
void inv3(const double *S, double *t) {
  const double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  const double d = 1.0 / det;
  t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

inline long sat_int(double v) {  // saturate_cast<int>(double) = cvRound with clamping
  if (!(v > -2147483648.0)) return -2147483648L;
  if (!(v < 2147483647.0)) return 2147483647L;
  return std::lrint(v);
}

// the maps of cv::undistort: stripes of max(1, 4096 / width) rows, the new camera's cy shifted by the stripe's first
// row, initUndistortRectifyMap(..., CV_16SC2) inside each
void build_maps(const double *K, const double *k, const double *P, int w, int h, int16_t *xy, uint16_t *frac) {
  const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
  const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
  double Ar[9];
  memcpy(Ar, P, sizeof(Ar));
  const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(w, 1)), h);
  const double vv0 = Ar[5];
  for (int y = 0; y < h; y += stripe0) {
    const int n = std::min(stripe0, h - y);
    Ar[5] = vv0 - y;
    double ir[9];
    inv3(Ar, ir);
    for (int i = 0; i < n; ++i) {
      int16_t *m1 = xy + ((size_t)(y + i) * w) * 2;
      uint16_t *m2 = frac + (size_t)(y + i) * w;
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double wi = 1. / _w, x = _x * wi, yy = _y * wi;
        const double x2 = x * x, y2 = yy * yy;
        const double r2 = x2 + y2, _2xy = 2 * x * yy;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
        const double v = fy * (yy * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
        const long iu = sat_int(u * kInterTab), iv = sat_int(v * kInterTab);
        m1[2 * j] = (int16_t)(iu >> kInterBits);
        m1[2 * j + 1] = (int16_t)(iv >> kInterBits);
        m2[j] = (uint16_t)((iv & (kInterTab - 1)) * kInterTab + (iu & (kInterTab - 1)));
      }
    }
  }
}

// remap INTER_LINEAR, BORDER_CONSTANT(0), 8 bit: weights (32 - fy)(32 - fx) * 32 ... (BilinearTab_i, scale 2^15; the
// identity entry is {32767, 0, 0, 1} after saturate_cast<short> and OpenCV's sum fix-up), FixedPtCast<int, uchar, 15>
template <int C>
__global__ __launch_bounds__(256) void k_undistort_remap(const uint8_t *__restrict__ src, int W, int H,
                                                         const short2 *__restrict__ xy, const uint16_t *__restrict__ frac,
                                                         int rx, int ry, int rw, int rh, uint8_t *__restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= rw || y >= rh) return;
  const size_t mi = (size_t)(y + ry) * W + (x + rx);
  const short2 s = xy[mi];
  const int f = frac[mi], fx = f & 31, fy = f >> 5;
  int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
  if (f == 0) w00 = 32767, w11 = 1;
  const int sx = s.x, sy = s.y;
  const bool in_x0 = sx >= 0 && sx < W, in_x1 = sx + 1 >= 0 && sx + 1 < W;
  const bool in_y0 = sy >= 0 && sy < H, in_y1 = sy + 1 >= 0 && sy + 1 < H;
  const uint8_t *p00 = src + ((size_t)sy * W + sx) * C;
  uint8_t *o = dst + ((size_t)y * rw + x) * C;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int v00 = (in_y0 && in_x0) ? p00[c] : 0, v01 = (in_y0 && in_x1) ? p00[C + c] : 0;
    const int v10 = (in_y1 && in_x0) ? p00[(size_t)W * C + c] : 0, v11 = (in_y1 && in_x1) ? p00[(size_t)W * C + C + c] : 0;
    const int acc = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
    o[c] = (uint8_t)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
  }
}

}  // namespace
}  // namespace sfmloc

using namespace sfmloc;

extern "C" int sfmloc_undistorter_create(int device, const double *K, const double *dist, uint32_t n_dist, uint32_t width,
                                         uint32_t height, sfmloc_undistorter **out) {
  SFM_CHECK(K && out && (dist || n_dist == 0), SFMLOC_EINVAL, "sfmloc_undistorter_create: null argument");
  SFM_CHECK(width >= 2 && height >= 2 && width <= 16384 && height <= 16384, SFMLOC_EINVAL,
            "sfmloc_undistorter_create: image %ux%u", width, height);
  SFM_CHECK(n_dist == 0 || n_dist == 4 || n_dist == 5 || n_dist == 8, SFMLOC_EINVAL,
            "sfmloc_undistorter_create: %u distortion coefficients (4, 5 or 8: k1 k2 p1 p2 [k3 [k4 k5 k6]])", n_dist);
  SFM_CHECK(K[0] != 0.0 && K[4] != 0.0, SFMLOC_EINVAL, "sfmloc_undistorter_create: zero focal length");
  Undistorter *u = new (std::nothrow) Undistorter;
  SFM_CHECK(u, SFMLOC_ENOMEM, "sfmloc_undistorter_create: out of host memory");
  u->device = device;
  u->w = width;
  u->h = height;
  double k[8];
  dist8(dist, n_dist, k);
  optimal_new_camera(K, k, (int)width, (int)height, 1.0, u->new_camera, u->roi);
  const size_t n = (size_t)width * height;
  u->h_xy.resize(2 * n);
  u->h_frac.resize(n);
  build_maps(K, k, u->new_camera, (int)width, (int)height, u->h_xy.data(), u->h_frac.data());
  *out = reinterpret_cast<sfmloc_undistorter *>(u);
  return SFMLOC_OK;
}

extern "C" void sfmloc_undistorter_destroy(sfmloc_undistorter *handle) {
  Undistorter *u = reinterpret_cast<Undistorter *>(handle);
  if (!u) return;
  if (u->d_xy || u->d_src || u->stream) hipSetDevice(u->device);
  if (u->d_xy) hipFree(u->d_xy);
  if (u->d_frac) hipFree(u->d_frac);
  if (u->d_src) hipFree(u->d_src);
  if (u->d_dst) hipFree(u->d_dst);
  if (u->stream) hipStreamDestroy(u->stream);
  delete u;
}

extern "C" int sfmloc_undistorter_info(const sfmloc_undistorter *handle, double *new_camera, int32_t *roi) {
  SFM_CHECK(handle, SFMLOC_EINVAL, "sfmloc_undistorter_info: null handle");
  const Undistorter *u = reinterpret_cast<const Undistorter *>(handle);
  if (new_camera) memcpy(new_camera, u->new_camera, sizeof(u->new_camera));
  if (roi) memcpy(roi, u->roi, sizeof(u->roi));
  return SFMLOC_OK;
}

extern "C" int sfmloc_undistorter_maps(const sfmloc_undistorter *handle, int16_t *xy, uint16_t *frac) {
  SFM_CHECK(handle, SFMLOC_EINVAL, "sfmloc_undistorter_maps: null handle");
  const Undistorter *u = reinterpret_cast<const Undistorter *>(handle);
  if (xy) memcpy(xy, u->h_xy.data(), u->h_xy.size() * sizeof(int16_t));
  if (frac) memcpy(frac, u->h_frac.data(), u->h_frac.size() * sizeof(uint16_t));
  return SFMLOC_OK;
}

extern "C" int sfmloc_undistorter_apply(sfmloc_undistorter *handle, const uint8_t *src, uint32_t channels, uint8_t *dst,
                                        uint64_t cap) {
  SFM_CHECK(handle && src && dst, SFMLOC_EINVAL, "sfmloc_undistorter_apply: null argument");
  SFM_CHECK(channels == 1 || channels == 3, SFMLOC_EINVAL, "sfmloc_undistorter_apply: %u channels (1 or 3)", channels);
  Undistorter *u = reinterpret_cast<Undistorter *>(handle);
  const size_t n_src = (size_t)u->w * u->h * channels, n_dst = (size_t)u->roi[2] * u->roi[3] * channels;
  SFM_CHECK(cap >= n_dst, SFMLOC_ECAP, "sfmloc_undistorter_apply: buffer holds %llu bytes, the valid region needs %llu",
            (unsigned long long)cap, (unsigned long long)n_dst);
  if (n_dst == 0) return SFMLOC_OK;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  SFM_CHECK(e == hipSuccess && ndev > 0, SFMLOC_ENODEV, "no HIP device visible; this library has no CPU fallback");
  SFM_HIP(hipSetDevice(u->device));
  if (!u->d_xy) {  // first image: the maps go to the device (the plan itself is host arithmetic and needs none)
    const size_t n = (size_t)u->w * u->h;
    SFM_HIP(hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking));
    SFM_HIP(hipMalloc((void **)&u->d_xy, n * sizeof(short2)));
    SFM_HIP(hipMalloc((void **)&u->d_frac, n * sizeof(uint16_t)));
    SFM_HIP(hipMemcpyAsync(u->d_xy, u->h_xy.data(), n * sizeof(short2), hipMemcpyHostToDevice, u->stream));
    SFM_HIP(hipMemcpyAsync(u->d_frac, u->h_frac.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice, u->stream));
  }
  if (u->src_cap < n_src) {
    if (u->d_src) SFM_HIP(hipFree(u->d_src));
    u->d_src = nullptr, u->src_cap = 0;
    SFM_HIP(hipMalloc((void **)&u->d_src, n_src));
    u->src_cap = n_src;
  }
  if (u->dst_cap < n_dst) {
    if (u->d_dst) SFM_HIP(hipFree(u->d_dst));
    u->d_dst = nullptr, u->dst_cap = 0;
    SFM_HIP(hipMalloc((void **)&u->d_dst, n_dst));
    u->dst_cap = n_dst;
  }
  SFM_HIP(hipMemcpyAsync(u->d_src, src, n_src, hipMemcpyHostToDevice, u->stream));
  const dim3 grid((u->roi[2] + 63) / 64, (u->roi[3] + 3) / 4), block(256);
  if (channels == 3)
    hipLaunchKernelGGL(k_undistort_remap<3>, grid, block, 0, u->stream, u->d_src, (int)u->w, (int)u->h, u->d_xy, u->d_frac,
                       u->roi[0], u->roi[1], u->roi[2], u->roi[3], u->d_dst);
  else
    hipLaunchKernelGGL(k_undistort_remap<1>, grid, block, 0, u->stream, u->d_src, (int)u->w, (int)u->h, u->d_xy, u->d_frac,
                       u->roi[0], u->roi[1], u->roi[2], u->roi[3], u->d_dst);
  SFM_HIP(hipGetLastError());
  SFM_HIP(hipMemcpyAsync(dst, u->d_dst, n_dst, hipMemcpyDeviceToHost, u->stream));
  SFM_HIP(hipStreamSynchronize(u->stream));
  return SFMLOC_OK;
}
