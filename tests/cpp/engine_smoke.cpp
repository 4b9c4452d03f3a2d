// Drives sfmloc::LocalizeEngine (include/sfmloc_engine.hpp) the way VisionLocalizeServer drives the reference's class:
//   engine_smoke <sfmDataDir> <matchDir> <AmatFile|-> <query.desc> <query.feat> <width> <height> [cx cy cz radius]
//   engine_smoke <sfmDataDir> <matchDir> <AmatFile|-> --image <file.jpg|png>        (localizeImage.cc:463: imread + localize)
// prints the 12 returned doubles (or "FAIL"), the number of 2D-3D points and the inlier indices.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sfmloc_engine.hpp"

static bool slurp(const char *p, std::vector<uint8_t> *out) {
  FILE *f = fopen(p, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize((size_t)n);
  const bool ok = fread(out->data(), 1, out->size(), f) == out->size();
  fclose(f);
  return ok;
}

int main(int argc, char **argv) {
  const bool image_mode = argc >= 6 && !strcmp(argv[4], "--image");
  if (argc < 8 && !image_mode) return 2;
  try {
    sfmloc::LocalizeEngine eng(argv[1], argv[2], strcmp(argv[3], "-") ? argv[3] : "", 0.6, 25, 4.0, false, 0, 0);
    if (image_mode) {
      std::vector<double> p2, p3, times;
      std::vector<int> inl;
      const std::vector<double> r = eng.localizeImageFile(argv[5], true, p2, p3, inl, true, times);
      if (r.empty()) {
        printf("FAIL\n");
      } else {
        for (double v : r) printf("%.17g ", v);
        printf("\n");
      }
      printf("%zu\n", p2.size() / 2);
      for (int i : inl) printf("%d ", i);
      printf("\n%zu\n", times.size());
      for (double v : times) printf("%.9g ", v);
      printf("%.9g\n", eng.lastTotalSeconds());
      return 0;
    }
    std::vector<uint8_t> raw;
    if (!slurp(argv[4], &raw) || raw.size() < 8) return 3;
    const uint32_t n = (uint32_t)((raw.size() - 8) / 64);
    std::vector<float> xy;
    FILE *f = fopen(argv[5], "r");
    double x, y, s, a;
    while (f && fscanf(f, "%lf %lf %lf %lf", &x, &y, &s, &a) == 4) {
      xy.push_back((float)x);
      xy.push_back((float)y);
    }
    if (f) fclose(f);
    std::vector<double> p2, p3, times, center;
    std::vector<int> inl;
    double radius = -1.0;
    if (argc >= 12) {
      center = {atof(argv[8]), atof(argv[9]), atof(argv[10])};
      radius = atof(argv[11]);
    }
    const std::vector<double> r = eng.localizeFeatures(raw.data() + 8, xy.data(), n, atoi(argv[6]), atoi(argv[7]), true,
                                                       p2, p3, inl, true, times, center, radius);
    if (r.empty()) {
      printf("FAIL\n");
    } else {
      for (double v : r) printf("%.17g ", v);
      printf("\n");
    }
    printf("%zu\n", p2.size() / 2);
    for (int i : inl) printf("%d ", i);
    printf("\n%zu\n", times.size());
    for (double v : times) printf("%.9g ", v);   // the six buckets of LocalizeEngine.cc:651-657, then the call's wall time
    printf("%.9g\n", eng.lastTotalSeconds());
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
