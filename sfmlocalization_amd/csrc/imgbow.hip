// sfmloc_imgbow: the query-side BoW vector straight from the IMAGE, resident on the device.
//
// The reference computes it per query in front of the shortlist (localization.cpp:346-361, LocalizeEngine.cc:205-232):
//   DenseLocalFeatureWrapper::calcDenseLocalFeature (DenseLocalFeatureWrapper.cpp:83-183): colour image -> 300x300
//     INTER_CUBIC -> gray -> min-max -> 50 x 50 x 4 grid keypoints (DenseFeatureDetector.cpp:44-69) -> cv::AKAZE::compute
//     -> the 61 descriptor bytes as floats
//   PcaWrapper::calcPcaProject (PcaWrapper.cpp:67-89) -> BoFSpatialPyramids::calcBoF (BoFSpatialPyramids.cpp:108-302).
// The stage-level entry points (sfmloc_dense_gray, sfmloc_akaze_compute, sfmloc_bof_compute) do the three steps as
// three synchronous calls with the image, 10 000 x 64 descriptor bytes and 10 000 x 61 floats crossing PCIe in between
// (and, for the first, seven allocations per call): 1.7 ms for a frame alone and 8 ms when four workers share the GPU.
// Here everything an image needs is allocated once (the resize tables, the 300 x 300 extractor, the grid, the BoF
// work space) and a call only queues kernels on ONE stream: image (H2D through a pinned staging buffer) -> resize + gray +
// min-max -> non-linear scale space of the 300 x 300 image -> orientation + M-LDB at the 10 000 grid points -> PCA /
// nearest word / pyramid histogram reading the descriptor bytes where they lie -> the float32 vector, written into the
// query's resident BoW slot (what sfmloc_query_set_bow uploads) and/or returned as the reference's float64 vector.
// Same kernels as the stage-level calls, so the same bits (tests/test_gpu_imgbow.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "sfmloc_internal.h"

namespace sfmloc {

struct ImgBow {
  int device = 0;
  uint32_t w = 0, h = 0, channels = 3;
  sfmloc_bof *bof = nullptr;       // the model (owned)
  sfmloc_akaze *ak = nullptr;      // size x size extractor, cv::AKAZE::create() defaults (owned)
  DenseGrayPlan plan;
  uint8_t *d_src = nullptr;        // [h*w*channels]
  uint8_t *h_src = nullptr;        // pinned staging of the same size
  float *d_grid = nullptr;         // [n_grid x 4] x, y, size, class_id
  float *d_kxy = nullptr;          // [n_grid x 2]
  uint32_t n_grid = 0;
  uint32_t *d_counts = nullptr;
  double *d_out = nullptr;         // [dim] the reference's float64 vector
  float *d_out_f32 = nullptr;      // [dim] as the shortlist reads it (BoFUtils.cpp:51-54 converts to CV_32F)
  double *h_out = nullptr;         // pinned [dim]
  hipEvent_t ordered = nullptr;    // sfmloc_imgbow_order_before
  hipEvent_t staged = nullptr;     // the previous call's H2D out of h_src has completed
  bool staged_pending = false;
  hipStream_t last_stream = nullptr;  // where the last call's work was queued (a batch: the leading extractor's stream)
};

}  // namespace sfmloc

using namespace sfmloc;

extern "C" {

void sfmloc_imgbow_destroy(sfmloc_imgbow *p) {
  ImgBow *ib = reinterpret_cast<ImgBow *>(p);
  if (!ib) return;
  hipSetDevice(ib->device);
  if (ib->ak) {
    hipStream_t s = akaze_stream_now(reinterpret_cast<Akaze *>(ib->ak));
    if (s) (void)hipStreamSynchronize(s);
  }
  dense_gray_plan_destroy(&ib->plan);
  void *ptrs[] = {ib->d_src, ib->d_grid, ib->d_kxy, ib->d_counts, ib->d_out, ib->d_out_f32};
  for (void *q : ptrs)
    if (q) (void)hipFree(q);
  if (ib->h_src) (void)hipHostFree(ib->h_src);
  if (ib->h_out) (void)hipHostFree(ib->h_out);
  if (ib->staged) (void)hipEventDestroy(ib->staged);
  if (ib->ordered) (void)hipEventDestroy(ib->ordered);
  if (ib->ak) sfmloc_akaze_destroy(ib->ak);
  if (ib->bof) sfmloc_bof_destroy(ib->bof);
  delete ib;
}

int sfmloc_imgbow_create(const sfmloc_bof_desc *model, int device, uint32_t width, uint32_t height, uint32_t channels,
                         sfmloc_imgbow **out) {
  SFM_CHECK(model && out, SFMLOC_EINVAL, "sfmloc_imgbow_create: null argument");
  *out = nullptr;
  SFM_CHECK(width >= 1 && height >= 1 && width <= 16384 && height <= 16384 && (channels == 1 || channels == 3),
            SFMLOC_EINVAL, "sfmloc_imgbow_create: image %ux%ux%u", width, height, channels);
  SFM_CHECK(model->in_dim == 61, SFMLOC_EINVAL, "sfmloc_imgbow_create: the dense features are 61 M-LDB bytes, model in_dim %d",
            model->in_dim);
  const int size = model->resized_image_size;
  SFM_CHECK(size >= 16 && size <= 4096, SFMLOC_EINVAL, "sfmloc_imgbow_create: ResizedImageSize %d", size);
  ImgBow *ib = new (std::nothrow) ImgBow();
  SFM_CHECK(ib, SFMLOC_ENOMEM, "out of host memory");
  ib->device = device;
  ib->w = width;
  ib->h = height;
  ib->channels = channels;
  int rc = sfmloc_bof_create(model, device, &ib->bof);
  if (!rc) rc = sfmloc_akaze_create(device, size, size, 4, 4, 0.001f, &ib->ak);  // DenseLocalFeatureWrapper.cpp:42
  if (!rc) rc = dense_gray_plan_create(&ib->plan, (int)width, (int)height, (int)channels, size);
  std::vector<float> grid, kxy;
  if (!rc) {
    try {  // DenseFeatureDetector.cpp:44-69 with the constants of DenseLocalFeatureWrapper.h:32-38
      float fs = 4.0f;
      for (int s = 0; s < 4; ++s) {
        for (int y = 0; y < size; y += 6)
          for (int x = 0; x < size; x += 6) {
            const float g[4] = {(float)x, (float)y, fs, (float)s};
            grid.insert(grid.end(), g, g + 4);
            kxy.push_back((float)x);
            kxy.push_back((float)y);
          }
        fs = fs * 1.5f;
      }
    } catch (const std::bad_alloc &) {
      set_error("sfmloc_imgbow_create: out of host memory");
      rc = SFMLOC_ENOMEM;
    }
  }
  if (!rc) {
    ib->n_grid = (uint32_t)(grid.size() / 4);
    const BofModel *b = reinterpret_cast<const BofModel *>(ib->bof);
    const size_t dim = (size_t)b->K * b->cells, n_src = (size_t)width * height * channels;
    hipError_t he = hipMalloc((void **)&ib->d_src, n_src);
    if (he == hipSuccess) he = hipHostMalloc((void **)&ib->h_src, n_src, hipHostMallocDefault);
    if (he == hipSuccess) he = hipMalloc((void **)&ib->d_grid, grid.size() * sizeof(float));
    if (he == hipSuccess) he = hipMalloc((void **)&ib->d_kxy, kxy.size() * sizeof(float));
    if (he == hipSuccess) he = hipMalloc((void **)&ib->d_counts, dim * sizeof(uint32_t));
    if (he == hipSuccess) he = hipMalloc((void **)&ib->d_out, dim * sizeof(double));
    if (he == hipSuccess) he = hipMalloc((void **)&ib->d_out_f32, dim * sizeof(float));
    if (he == hipSuccess) he = hipHostMalloc((void **)&ib->h_out, dim * sizeof(double), hipHostMallocDefault);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&ib->staged, hipEventDisableTiming);
    if (he == hipSuccess) he = hipMemcpy(ib->d_grid, grid.data(), grid.size() * sizeof(float), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(ib->d_kxy, kxy.data(), kxy.size() * sizeof(float), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
      set_error("sfmloc_imgbow_create: %s", hipGetErrorString(he));
      rc = he == hipErrorOutOfMemory ? SFMLOC_ENOMEM : SFMLOC_EHIP;
    }
  }
  if (rc) {
    sfmloc_imgbow_destroy(reinterpret_cast<sfmloc_imgbow *>(ib));
    return rc;
  }
  *out = reinterpret_cast<sfmloc_imgbow *>(ib);
  return SFMLOC_OK;
}

int sfmloc_imgbow_dim(const sfmloc_imgbow *p) {
  const ImgBow *ib = reinterpret_cast<const ImgBow *>(p);
  return ib ? sfmloc_bof_dim(ib->bof) : 0;
}

const void *sfmloc_imgbow_vector_dev(const sfmloc_imgbow *p) {
  const ImgBow *ib = reinterpret_cast<const ImgBow *>(p);
  return ib ? ib->d_out_f32 : nullptr;
}

// the context's stream waits for what this extractor has queued so far (an extractor on a stream of its own runs beside
// the feature extraction of the same frame; the localisation that reads its vector is ordered behind it by this)
int sfmloc_imgbow_order_before(sfmloc_imgbow *p, sfmloc_context *ctx) {
  SFM_CHECK(p && ctx, SFMLOC_EINVAL, "sfmloc_imgbow_order_before: null argument");
  ImgBow *ib = reinterpret_cast<ImgBow *>(p);
  Ctx *c = reinterpret_cast<Ctx *>(ctx);
  SFM_HIP(hipSetDevice(ib->device));
  // (an extractor that last worked as a member of a batch has its work on the batch's stream, and may have none of its own)
  hipStream_t s = ib->last_stream ? ib->last_stream : akaze_stream_now(reinterpret_cast<Akaze *>(ib->ak));
  hipStream_t cs = c->stream;
  if (s == cs) return SFMLOC_OK;  // (shared: stream order already)
  if (!ib->ordered) SFM_HIP(hipEventCreateWithFlags(&ib->ordered, hipEventDisableTiming));
  SFM_HIP(hipEventRecord(ib->ordered, s));
  SFM_HIP(hipStreamWaitEvent(cs, ib->ordered, 0));
  return SFMLOC_OK;
}

int sfmloc_imgbow_share_stream(sfmloc_imgbow *p, sfmloc_context *ctx) {
  SFM_CHECK(p, SFMLOC_EINVAL, "sfmloc_imgbow_share_stream: null argument");
  ImgBow *ib = reinterpret_cast<ImgBow *>(p);
  return sfmloc_akaze_share_stream(ib->ak, ctx);
}

int sfmloc_imgbow_compute(sfmloc_imgbow *p, const uint8_t *image, sfmloc_query *query, double *out_bow) {
  SFM_CHECK(p && image, SFMLOC_EINVAL, "sfmloc_imgbow_compute: null argument");
  ImgBow *ib = reinterpret_cast<ImgBow *>(p);
  Query *q = reinterpret_cast<Query *>(query);
  const BofModel *b = reinterpret_cast<const BofModel *>(ib->bof);
  const size_t dim = (size_t)b->K * b->cells;
  if (q) {
    SFM_CHECK(q->map && q->map->bow_dim == dim, SFMLOC_EINVAL,
              "sfmloc_imgbow_compute: the model's vector has %zu entries, the map's .bow vectors %u", dim,
              q->map ? q->map->bow_dim : 0u);
    SFM_CHECK(!q->is_view, SFMLOC_EINVAL, "sfmloc_imgbow_compute: a view's BoW vector belongs to the caller");
    SFM_CHECK(q->map->device == ib->device, SFMLOC_EINVAL, "sfmloc_imgbow_compute: query and extractor on different devices");
  }
  SFM_HIP(hipSetDevice(ib->device));
  if (q && !q->d_bow) SFM_HIP(hipMalloc((void **)&q->d_bow, dim * sizeof(float)));  // (once per query object)
  Akaze *a = reinterpret_cast<Akaze *>(ib->ak);
  const size_t n_src = (size_t)ib->w * ib->h * ib->channels;
  // the staging buffer is free once the previous call's upload has left it
  if (ib->staged_pending) SFM_HIP(hipEventSynchronize(ib->staged));
  memcpy(ib->h_src, image, n_src);
  hipStream_t s = akaze_stream_now(a);
  SFM_HIP(hipMemcpyAsync(ib->d_src, ib->h_src, n_src, hipMemcpyHostToDevice, s));
  SFM_HIP(hipEventRecord(ib->staged, s));
  ib->staged_pending = true;
  int rc = dense_gray_enqueue(&ib->plan, s, ib->d_src, akaze_gray_dev(a));
  if (!rc) rc = akaze_compute_resident(a, ib->d_grid, ib->n_grid, 4);  // (the grid's four scales are levels 0 .. 3)
  if (rc) return rc;
  s = akaze_stream_now(a);
  rc = launch_bof(b, s, nullptr, ib->d_kxy, (int)ib->n_grid, ib->d_counts, ib->d_out, q ? q->d_bow : ib->d_out_f32,
                  akaze_desc_dev(a));
  if (rc) return rc;
  ib->last_stream = s;
  if (out_bow) {
    SFM_HIP(hipMemcpyAsync(ib->h_out, ib->d_out, dim * sizeof(double), hipMemcpyDeviceToHost, s));
    SFM_HIP(hipStreamSynchronize(s));
    memcpy(out_bow, ib->h_out, dim * sizeof(double));
  }
  return SFMLOC_OK;
}

// the float64 vector of the extractor's last call (single or batch), after waiting for it
int sfmloc_imgbow_vector_read(sfmloc_imgbow *p, double *out_bow) {
  SFM_CHECK(p && out_bow, SFMLOC_EINVAL, "sfmloc_imgbow_vector_read: null argument");
  ImgBow *ib = reinterpret_cast<ImgBow *>(p);
  SFM_CHECK(ib->last_stream, SFMLOC_EINVAL, "sfmloc_imgbow_vector_read: nothing computed yet");
  const BofModel *b = reinterpret_cast<const BofModel *>(ib->bof);
  const size_t dim = (size_t)b->K * b->cells;
  SFM_HIP(hipSetDevice(ib->device));
  SFM_HIP(hipMemcpyAsync(ib->h_out, ib->d_out, dim * sizeof(double), hipMemcpyDeviceToHost, ib->last_stream));
  SFM_HIP(hipStreamSynchronize(ib->last_stream));
  memcpy(out_bow, ib->h_out, dim * sizeof(double));
  return SFMLOC_OK;
}

// The BoW vectors of n frames, one launch per kernel for all of them: the extractors work as ONE gang session on the
// first one's stream (gang.h) -- resize + gray + min-max, the 300 x 300 scale space, orientation + M-LDB at the grid, PCA /
// words / pyramid histogram -- where n calls of sfmloc_imgbow_compute queue n chains of ~20 small kernels each.  The
// vectors stay on the device (sfmloc_imgbow_vector_dev of each extractor); same kernels on the same inputs, so the same
// bits as the single calls (tests/test_gpu_imgbow.py).  The reference computes the vector per query image
// (localization.cpp:346-361); a batch is what a server with several frames in hand does.
int sfmloc_imgbow_compute_batch(sfmloc_imgbow *const *ibs, const uint8_t *const *images, uint32_t n) {
  SFM_CHECK(ibs && images && n >= 1 && n <= (uint32_t)kGangMembers, SFMLOC_EINVAL, "sfmloc_imgbow_compute_batch: 1..%d images",
            kGangMembers);
  GangMember *ms[kGangMembers];
  ImgBow *first = reinterpret_cast<ImgBow *>(ibs[0]);
  for (uint32_t i = 0; i < n; ++i) {
    ImgBow *ib = reinterpret_cast<ImgBow *>(ibs[i]);
    SFM_CHECK(ib && images[i], SFMLOC_EINVAL, "sfmloc_imgbow_compute_batch: null argument (image %u)", i);
    SFM_CHECK(ib->device == first->device && ib->w == first->w && ib->h == first->h && ib->channels == first->channels,
              SFMLOC_EINVAL, "sfmloc_imgbow_compute_batch: the extractors differ in device or image size");
    for (uint32_t j = 0; j < i; ++j) SFM_CHECK(ibs[j] != ibs[i], SFMLOC_EINVAL, "extractor listed twice");
    ms[i] = akaze_member(reinterpret_cast<Akaze *>(ib->ak));
    SFM_CHECK(ms[i]->stream.gang == nullptr, SFMLOC_EINVAL, "sfmloc_imgbow_compute_batch: extractor %u is in a session", i);
  }
  SFM_HIP(hipSetDevice(first->device));
  const size_t n_src = (size_t)first->w * first->h * first->channels;
  for (uint32_t i = 0; i < n; ++i) {  // the frames into the pinned staging buffers (host work, before anything is queued)
    ImgBow *ib = reinterpret_cast<ImgBow *>(ibs[i]);
    if (ib->staged_pending) SFM_HIP(hipEventSynchronize(ib->staged));
    memcpy(ib->h_src, images[i], n_src);
  }
  hipStream_t s0 = akaze_stream_now(reinterpret_cast<Akaze *>(first->ak));  // (the session's stream: the first extractor's)
  int rc = n > 1 ? gang_open(ms, (int)n) : SFMLOC_OK;
  for (uint32_t i = 0; i < n && rc == SFMLOC_OK; ++i) {
    ImgBow *ib = reinterpret_cast<ImgBow *>(ibs[i]);
    Akaze *a = reinterpret_cast<Akaze *>(ib->ak);
    const BofModel *b = reinterpret_cast<const BofModel *>(ib->bof);
    // the upload depends on nothing recorded so far: queued at once, ahead of the session's launches
    hipStream_t su = ms[i]->stream.unordered();
    SFM_HIP(hipMemcpyAsync(ib->d_src, ib->h_src, n_src, hipMemcpyHostToDevice, su));
    SFM_HIP(hipEventRecord(ib->staged, su));
    ib->staged_pending = true;
    rc = dense_gray_enqueue_member(&ib->plan, ms[i], ib->d_src, akaze_gray_dev(a));
    if (!rc) rc = akaze_compute_resident(a, ib->d_grid, ib->n_grid, 4);
    if (!rc) rc = launch_bof_member(b, ms[i], ib->d_kxy, (int)ib->n_grid, ib->d_counts, ib->d_out, ib->d_out_f32, akaze_desc_dev(a));
    ib->last_stream = s0;
  }
  const int rc_close = n > 1 ? gang_close(ms[0]) : SFMLOC_OK;
  return rc ? rc : rc_close;
}

}  // extern "C"
