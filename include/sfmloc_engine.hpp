// sfmloc::LocalizeEngine -- the reference's in-process localisation class (VisionLocalizeServer/src/LocalizeEngine.h:42-83,
// LocalizeEngine.cc:84-661) as a header-only C++ wrapper over the C ABI of sfmloc.h.
//
// Same constructor arguments and the same localize() contract: returns [t(3), R(9 row-major)] in the frame of the A
// matrix (LocalizeEngine.cc:593-625), an EMPTY vector on failure (:453,481,579); points2D / points3D are the 2D-3D
// correspondences of the resection (n x 2, n x 3, the latter A-transformed, :520-553), pointsInlier the indices of
// the inliers among them, times the six stage durations of :651-657 (selectBeacon = 0: no iBeacon stage; selectBow,
// putMatch, geoMatch, PnP from the query's per-stage HIP events, extFeat measured around the extraction by the image
// entry points; lastTotalSeconds() is the whole call).
// What cannot be mirrored without OpenCV: cv::Mat arguments become plain buffers (8-bit gray image or precomputed
// descriptors + keypoints); iBeacon pre-selection (beaconKnnNum) is out of scope and must be 0.  guidedMatching is
// honoured (sfmloc_params.guided_matching).  Not re-entrant, like the original: one engine per concurrent user
// (localizeImage.cc:71-74) -- or use sfmloc_context directly to share one map.
#ifndef SFMLOC_ENGINE_HPP
#define SFMLOC_ENGINE_HPP

#include <sys/stat.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sfmloc.h"

namespace sfmloc {

// The per-user undistortion the server applies before LocalizeEngine::localize (localizeImage.cc:149-177):
// getOptimalNewCameraMatrix(K, dist, size, 1.0, size, &validRoi) once, then cv::undistort + crop per image (GPU).
class Undistorter {
 public:
  Undistorter(const double K[9], const std::vector<double> &dist, int width, int height, int device = 0) {
    if (sfmloc_undistorter_create(device, K, dist.empty() ? nullptr : dist.data(), (uint32_t)dist.size(), (uint32_t)width,
                                  (uint32_t)height, &mU))
      throw std::runtime_error(sfmloc_last_error());
    sfmloc_undistorter_info(mU, mNewCamera, mRoi);
  }
  Undistorter(const Undistorter &) = delete;
  Undistorter &operator=(const Undistorter &) = delete;
  ~Undistorter() { sfmloc_undistorter_destroy(mU); }
  const double *newCameraMatrix() const { return mNewCamera; }  // 3x3 row-major
  int roiX() const { return mRoi[0]; }
  int roiY() const { return mRoi[1]; }
  int roiWidth() const { return mRoi[2]; }
  int roiHeight() const { return mRoi[3]; }
  // src: height x width x channels (1 or 3) -> the valid region, roiHeight() x roiWidth() x channels
  std::vector<uint8_t> apply(const uint8_t *src, int channels) {
    std::vector<uint8_t> out((size_t)mRoi[2] * mRoi[3] * channels);
    if (sfmloc_undistorter_apply(mU, src, (uint32_t)channels, out.data(), out.size()))
      throw std::runtime_error(sfmloc_last_error());
    return out;
  }

 private:
  sfmloc_undistorter *mU = nullptr;
  double mNewCamera[9];
  int32_t mRoi[4];
};

class LocalizeEngine {
 public:
  LocalizeEngine(const std::string &sfmDataDir, const std::string &matchDir, const std::string &AmatFile,
                 double secondTestRatio, int ransacRound, double ransacPrecision, bool guidedMatching,
                 int beaconKnnNum = 0, int bowKnnNum = 0, int device = 0)
      : mMatchDir(matchDir), mBowKnnNum(bowKnnNum), mDevice(device) {
    if (beaconKnnNum) throw std::invalid_argument("iBeacon view pre-selection is out of scope");
    sfmloc_params p;
    sfmloc_default_params(&p);
    p.dist_ratio = (float)secondTestRatio;
    p.ransac_round = ransacRound;
    p.geom_precision = ransacPrecision;
    p.bow_knn = bowKnnNum;
    p.device = device;
    p.guided_matching = guidedMatching ? 1 : 0;  // mGuidedMatching -> geometricMatch (LocalizeEngine.cc:459)
    // sfmDataDir may also name a packed map file written by sfmloc_pack
    struct stat st;
    const bool packed = ::stat(sfmDataDir.c_str(), &st) == 0 && S_ISREG(st.st_mode);
    if (packed ? sfmloc_open_packed(sfmDataDir.c_str(), &p, &mMap)
               : sfmloc_open(sfmDataDir.c_str(), matchDir.c_str(), &p, &mMap))
      throw std::runtime_error(sfmloc_last_error());
    sfmloc_map_info info;
    sfmloc_map_get_info(mMap, &info);
    mViews = info.n_views;
    mCenters.resize(3 * (size_t)mViews);
    if (sfmloc_map_views(mMap, nullptr, nullptr, mCenters.data())) mCenters.clear();
    if (!AmatFile.empty()) readAmat(AmatFile);  // cv::FileStorage "A": 3x4 (LocalizeEngine.cc:116-118)
    readImageDescriber(matchDir + "/image_describer.txt");
  }
  LocalizeEngine(const LocalizeEngine &) = delete;
  LocalizeEngine &operator=(const LocalizeEngine &) = delete;
  ~LocalizeEngine() {
    for (auto &kv : mAkaze)
      if (kv.second) sfmloc_akaze_destroy(kv.second);
    if (mMap) sfmloc_map_destroy(mMap);
  }

  // LocalizeEngine::localize on an 8-bit gray image (extraction on the GPU with the map's image_describer.txt options)
  std::vector<double> localize(const uint8_t *gray, int width, int height, bool bReturnKeypoints,
                               std::vector<double> &points2D, std::vector<double> &points3D,
                               std::vector<int> &pointsInlier, bool bReturnTime, std::vector<double> &times,
                               const std::vector<double> &center = std::vector<double>(), double radius = -1.0,
                               const std::vector<float> *bow = nullptr) {
    sfmloc_akaze *&ak = mAkaze[std::make_pair(width, height)];
    if (!ak && sfmloc_akaze_create(mDevice, width, height, mNOct, mNOctLay, mThres, &ak))
      throw std::runtime_error(sfmloc_last_error());
    const uint32_t cap = 65536;
    std::vector<float> kp((size_t)cap * 6);
    std::vector<uint8_t> desc((size_t)cap * 64);
    uint32_t n = 0;
    const auto tFeat = std::chrono::steady_clock::now();
    if (sfmloc_akaze_detect_and_compute(ak, gray, kp.data(), desc.data(), cap, &n))
      throw std::runtime_error(sfmloc_last_error());
    const double featSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tFeat).count();
    std::vector<float> xy((size_t)n * 2);
    for (uint32_t i = 0; i < n; ++i) {  // locFeat takes KeyPoint::pt as is (LocalizeEngine.cc:228-231); only the
      xy[2 * i] = kp[6 * i];            // command-line tool reads the 6-digit .feat text back
      xy[2 * i + 1] = kp[6 * i + 1];
    }
    std::vector<double> r = localizeFeatures(desc.data(), xy.data(), n, width, height, bReturnKeypoints, points2D,
                                             points3D, pointsInlier, bReturnTime, times, center, radius, bow);
    if (bReturnTime && times.size() == 6) times[2] = featSeconds;  // "Extract feature from query image"
    return r;
  }
  double lastTotalSeconds() const { return mLastTotal; }

  // the server's calls: a decoded colour image (B G R, what cv::imdecode / cv::imread(IMREAD_COLOR) hand to
  // LocalizeEngine::localize, localizeImage.cc:393/463); cv::AKAZE::detectAndCompute converts it with BGR2GRAY
  std::vector<double> localizeBGR(const uint8_t *bgr, int width, int height, bool bReturnKeypoints,
                                  std::vector<double> &points2D, std::vector<double> &points3D,
                                  std::vector<int> &pointsInlier, bool bReturnTime, std::vector<double> &times,
                                  const std::vector<double> &center = std::vector<double>(), double radius = -1.0,
                                  const std::vector<float> *bow = nullptr) {
    std::vector<uint8_t> gray((size_t)width * height);
    for (size_t i = 0; i < gray.size(); ++i)
      gray[i] = (uint8_t)((bgr[3 * i + 2] * 4899 + bgr[3 * i + 1] * 9617 + bgr[3 * i] * 1868 + 8192) >> 14);
    return localize(gray.data(), width, height, bReturnKeypoints, points2D, points3D, pointsInlier, bReturnTime, times,
                    center, radius, bow);
  }

  // ... and an image file (JPEG / PNG / PGM / PPM) through the library's own imread
  std::vector<double> localizeImageFile(const std::string &path, bool bReturnKeypoints, std::vector<double> &points2D,
                                        std::vector<double> &points3D, std::vector<int> &pointsInlier,
                                        bool bReturnTime, std::vector<double> &times,
                                        const std::vector<double> &center = std::vector<double>(),
                                        double radius = -1.0, const std::vector<float> *bow = nullptr) {
    int32_t w = 0, h = 0;
    if (sfmloc_image_read(path.c_str(), 1, nullptr, 0, &w, &h)) throw std::runtime_error(sfmloc_last_error());
    std::vector<uint8_t> bgr((size_t)w * h * 3);
    if (sfmloc_image_read(path.c_str(), 1, bgr.data(), bgr.size(), &w, &h)) throw std::runtime_error(sfmloc_last_error());
    return localizeBGR(bgr.data(), w, h, bReturnKeypoints, points2D, points3D, pointsInlier, bReturnTime, times, center,
                       radius, bow);
  }

  // the same from the output of extractAKAZESingleImg: desc [n x 64] (.desc rows), keypoints [n x 2]
  std::vector<double> localizeFeatures(const uint8_t *desc, const float *kptXY, uint32_t n, int width, int height,
                                       bool bReturnKeypoints, std::vector<double> &points2D,
                                       std::vector<double> &points3D, std::vector<int> &pointsInlier,
                                       bool bReturnTime, std::vector<double> &times,
                                       const std::vector<double> &center = std::vector<double>(), double radius = -1.0,
                                       const std::vector<float> *bow = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> result;
    points2D.clear();
    points3D.clear();
    pointsInlier.clear();
    std::vector<uint32_t> sel;
    bool useSel = false;
    if (center.size() == 3 && radius > 0 && !mCenters.empty()) {  // getLocalViews (LocalizeEngine.cc:200-; squared vs un-squared)
      useSel = true;
      for (uint32_t v = 0; v < mViews; ++v) {
        double c[3] = {mCenters[3 * v], mCenters[3 * v + 1], mCenters[3 * v + 2]};
        if (mHaveA) applyA(c);
        const double dx = c[0] - center[0], dy = c[1] - center[1], dz = c[2] - center[2];
        if (dx * dx + dy * dy + dz * dz <= radius) sel.push_back(v);
      }
      if (sel.empty()) return result;
    }
    sfmloc_query *q = nullptr;
    if (sfmloc_query_create(mMap, desc, kptXY, n, (uint32_t)width, (uint32_t)height, &q))
      throw std::runtime_error(sfmloc_last_error());
    sfmloc_pose pose;
    std::memset(&pose, 0, sizeof(pose));
    std::vector<uint32_t> pq(65536), pl(65536);  // a query has at most 65 535 features, hence inliers
    if (bReturnTime) sfmloc_set_profile(mMap, 1);  // the reference's `times` come from this query's per-stage events
    const uint32_t *selp = useSel ? sel.data() : nullptr;
    const uint32_t nsel = useSel ? (uint32_t)sel.size() : 0;
    // with a BoW vector the shortlist (applied when more than bowKnnNum views remain, LocalizeEngine.cc:342) and the
    // path run as one call, the shortlist staying on the device
    const int rc = (bow && mBowKnnNum > 0)
                       ? sfmloc_localize_bow(mMap, q, bow->data(), (uint32_t)mBowKnnNum, selp, nsel, &pose, pq.data(),
                                             pl.data(), 65536)
                       : sfmloc_localize(mMap, q, selp, nsel, &pose, pq.data(), pl.data(), 65536);
    if (bReturnTime) sfmloc_set_profile(mMap, mProfile);
    if (rc) {
      sfmloc_query_destroy(q);
      throw std::runtime_error(sfmloc_last_error());
    }
    if (bReturnKeypoints && pose.n_matches_2d3d > 0) {
      const uint32_t cap = (uint32_t)pose.n_matches_2d3d;
      std::vector<uint32_t> qf(cap), lm(cap), inl(65536);
      points2D.assign((size_t)cap * 2, 0.0);
      points3D.assign((size_t)cap * 3, 0.0);
      uint32_t n2 = 0;
      sfmloc_pose tmp;
      if (sfmloc_match_set_read(mMap, &n2, qf.data(), lm.data(), points2D.data(), points3D.data(), cap) ||
          sfmloc_pose_read(mMap, &tmp, nullptr, nullptr, inl.data(), 65536)) {
        sfmloc_query_destroy(q);
        throw std::runtime_error(sfmloc_last_error());
      }
      if (mHaveA)
        for (uint32_t i = 0; i < n2; ++i) applyA(&points3D[3 * (size_t)i]);
      for (int i = 0; i < pose.n_inliers; ++i) pointsInlier.push_back((int)inl[i]);
    }
    sfmloc_query_destroy(q);
    if (bReturnTime) {  // LocalizeEngine.cc:651-657: selectBeacon, selectBow, extFeat (the image entry points fill it),
      times.assign(6, 0.0);  // putMatch, geoMatch, PnP
      times[1] = pose.stage_seconds[1];
      times[3] = pose.stage_seconds[3];
      times[4] = pose.stage_seconds[4];
      times[5] = pose.stage_seconds[5];
      mLastTotal = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!pose.ok) return result;
    double c[3] = {pose.center[0], pose.center[1], pose.center[2]};
    double R[9];
    std::memcpy(R, pose.R, sizeof(R));
    if (mHaveA) {  // localising in the A-transformed map (LocalizeEngine.cc:121-144) = transforming the result
      applyA(c);
      const double det = mA[0] * (mA[5] * mA[10] - mA[6] * mA[9]) - mA[1] * (mA[4] * mA[10] - mA[6] * mA[8]) +
                         mA[2] * (mA[4] * mA[9] - mA[5] * mA[8]);
      const double s = std::cbrt(det);
      double Rn[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          Rn[3 * i + j] = (R[3 * i] * mA[4 * j] + R[3 * i + 1] * mA[4 * j + 1] + R[3 * i + 2] * mA[4 * j + 2]) / s;
      std::memcpy(R, Rn, sizeof(R));
    }
    result.assign(c, c + 3);
    result.insert(result.end(), R, R + 9);
    return result;
  }

  sfmloc_map *map() const { return mMap; }

 private:
  void applyA(double *p) const {
    const double x = p[0], y = p[1], z = p[2];
    for (int i = 0; i < 3; ++i) p[i] = mA[4 * i] * x + mA[4 * i + 1] * y + mA[4 * i + 2] * z + mA[4 * i + 3];
  }
  void readAmat(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("Cannot find A mat file");
    std::string txt;
    char buf[4096];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof(buf), f)) > 0) txt.append(buf, k);
    std::fclose(f);
    const size_t a = txt.find("A:"), d = txt.find("data:", a == std::string::npos ? 0 : a);
    const size_t lb = txt.find('[', d == std::string::npos ? 0 : d), rb = txt.find(']', lb == std::string::npos ? 0 : lb);
    if (a == std::string::npos || d == std::string::npos || lb == std::string::npos || rb == std::string::npos)
      throw std::runtime_error("A mat file: no 3x4 matrix \"A\"");
    std::string body = txt.substr(lb + 1, rb - lb - 1);
    for (char &ch : body)
      if (ch == ',' || ch == '\n') ch = ' ';
    const char *s = body.c_str();
    char *e = nullptr;
    int n = 0;
    for (double v = std::strtod(s, &e); s != e && n < 12; v = std::strtod(s, &e)) {
      mA[n++] = v;
      s = e;
    }
    if (n != 12) throw std::runtime_error("A mat file: \"A\" must hold 12 values");
    mHaveA = true;
  }
  void readImageDescriber(const std::string &path) {  // AKAZEOption.cpp:44-55; defaults AKAZEOption.h:31-34
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return;
    char line[512];
    while (std::fgets(line, sizeof(line), f)) {
      char key[128];
      double v;
      if (std::sscanf(line, " %127[^:]: %lf", key, &v) == 2) {
        if (!std::strcmp(key, "thres")) mThres = (float)v;
        if (!std::strcmp(key, "nOct")) mNOct = (int)v;
        if (!std::strcmp(key, "nOctLay")) mNOctLay = (int)v;
      }
    }
    std::fclose(f);
  }

  std::string mMatchDir;
  sfmloc_map *mMap = nullptr;
  uint32_t mViews = 0;
  std::vector<double> mCenters;
  double mA[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  bool mHaveA = false;
  int mBowKnnNum = 0, mDevice = 0;
  int mProfile = 0;          // params.profile outside a timed call
  double mLastTotal = 0.0;   // wall time of the last localize call with bReturnTime
  float mThres = 0.001f;
  int mNOct = 4, mNOctLay = 4;
  std::map<std::pair<int, int>, sfmloc_akaze *> mAkaze;
};

// One rank of a map sharded by view over N GPUs (include/sfmloc.h "A rank's whole batch per call"): the
// contexts of the two stages and the batch entry points over them, for a C++ host that drives the exchange itself
// (RCCL / MPI all-gathers of `keys` and `packed` between the calls; dist.py is the Python host of the same calls).
//   stage 1 (every rank, every query of the batch): bowKeys -> [all-gather keys] -> stage1Bow -> [all-gather packed]
//   stage 2 (the owner of a query, i mod N):        stage2 -> finish(k) per context
// Every call is asynchronous; ordering against the caller's communication stream is by sfmloc_context_signal / _wait on
// the contexts (context(i) / mergeContext(k)).  Results are those of the unsharded path (tests/test_gpu_gang.py,
// tests/test_gpu_fullsize.py drive the same entry points).
class ShardRank {
 public:
  // map: this rank's shard (opened with its view range); nStage1 contexts in sessions of `gang` (<= 32), nStage2 <= 32
  ShardRank(sfmloc_map *map, uint32_t nStage1 = 32, uint32_t gang = 16, uint32_t nStage2 = 32) : mGang(gang) {
    if (!map || gang < 1 || gang > 32 || nStage1 < gang || nStage1 % gang || nStage2 < 1 || nStage2 > 32)
      throw std::invalid_argument("ShardRank: 1 <= gang <= 32, nStage1 a multiple of gang, 1 <= nStage2 <= 32");
    try {
      for (uint32_t i = 0; i < nStage1; ++i) {  // the first context of a session lends its stream to the others
        sfmloc_context *c = nullptr;
        sfmloc_context *lender = (i % gang) ? mStage1[i - i % gang] : nullptr;
        if (lender ? sfmloc_context_create_sharing(map, lender, &c) : sfmloc_context_create(map, &c))
          throw std::runtime_error(sfmloc_last_error());
        mStage1.push_back(c);
      }
      for (uint32_t k = 0; k < nStage2; ++k) {  // stage 2 needs no matching workspace
        sfmloc_context *c = nullptr;
        if (sfmloc_context_create_merge(map, k ? mStage2[0] : nullptr, &c)) throw std::runtime_error(sfmloc_last_error());
        mStage2.push_back(c);
      }
    } catch (...) {
      destroy();
      throw;
    }
  }
  ShardRank(const ShardRank &) = delete;
  ShardRank &operator=(const ShardRank &) = delete;
  ~ShardRank() { destroy(); }

  sfmloc_context *context(uint32_t i) const { return mStage1.at(i); }
  sfmloc_context *mergeContext(uint32_t k) const { return mStage2.at(k); }
  uint32_t stage2Capacity() const { return (uint32_t)mStage2.size(); }
  static uint64_t packedBytes(uint32_t nQueries, uint32_t budget) { return sfmloc_packed_bytes(nQueries, budget); }

  // every query's knn best views of this shard as 64-bit keys -> keysDev [nQueries][knn] (device memory)
  void bowKeys(sfmloc_query *const *queries, uint32_t nQueries, uint32_t knn, void *keysDev) {
    check(sfmloc_shard_batch_bow_keys(mStage1.data(), (uint32_t)mStage1.size(), mGang, queries, nQueries, knn, keysDev));
  }
  // stage 1 on this shard's part of the global shortlist (keysAllDev: the gathered keys of nParts ranks); the batch's
  // candidates -> packedDev (packedBytes(nQueries, budget))
  void stage1Bow(sfmloc_query *const *queries, uint32_t nQueries, const void *keysAllDev, uint32_t nParts, uint32_t knn,
                 void *packedDev, uint32_t budget) {
    check(sfmloc_shard_batch_begin_bow(mStage1.data(), (uint32_t)mStage1.size(), mGang, queries, nQueries, keysAllDev, nParts,
                                       knn, packedDev, budget));
  }
  // ... without a shortlist: every view of the shard
  void stage1(sfmloc_query *const *queries, uint32_t nQueries, void *packedDev, uint32_t budget) {
    check(sfmloc_shard_batch_begin(mStage1.data(), (uint32_t)mStage1.size(), mGang, queries, nQueries, packedDev, budget));
  }
  // stage 2 of n <= stage2Capacity() queries this rank owns: merge context k takes batch query queryIndex[k], whose
  // candidates lie in the gathered parts (packedAllDev: nParts parts, partStride bytes apart, 0 = back to back)
  void stage2(sfmloc_query *const *queries, const uint32_t *queryIndex, uint32_t n, const void *packedAllDev, uint32_t nParts,
              uint64_t partStride, uint32_t nQueries, uint32_t budget) {
    if (n > mStage2.size()) throw std::invalid_argument("ShardRank::stage2: more queries than merge contexts");
    check(sfmloc_merge_batch_begin(mStage2.data(), n, queries, queryIndex, packedAllDev, nParts, partStride, nQueries, budget));
  }
  // the result of merge context k's query (waits for it): pose + inlier pairs, as sfmloc_localize_end
  void finish(uint32_t k, sfmloc_pose *pose, uint32_t *pairQfeat, uint32_t *pairLandmark, uint32_t cap) {
    check(sfmloc_localize_end(mStage2.at(k), pose, pairQfeat, pairLandmark, cap));
  }

 private:
  static void check(int rc) {
    if (rc) throw std::runtime_error(sfmloc_last_error());
  }
  void destroy() {
    for (size_t k = mStage2.size(); k-- > 0;) sfmloc_context_destroy(mStage2[k]);
    for (size_t i = mStage1.size(); i-- > 0;) sfmloc_context_destroy(mStage1[i]);  // borrowers before their lenders
    mStage2.clear();
    mStage1.clear();
  }
  uint32_t mGang;
  std::vector<sfmloc_context *> mStage1, mStage2;
};

}  // namespace sfmloc
#endif  // SFMLOC_ENGINE_HPP
