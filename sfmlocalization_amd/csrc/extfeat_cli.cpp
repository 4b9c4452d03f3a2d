// ExtFeatAndMatch, the reference's map-building front end (ExtFeatAndMatch/src/computeFeaturesAndMatches.cpp:49-247),
// as a C++ host program over the C ABI of libsfmloc_hip.so:
//
//   ExtFeatAndMatch <matchDir> [-c=3 -t=0.001 -o=4 -l=4 -f=0.6 -r=4096 -v=0 -p= -mf=0 -mm=60 -g=4.0 -gm -sm]
//
//   <matchDir>/sfm_data.json (views)  ->  image_describer.txt, <base>.feat / <base>.desc per view,
//                                         matches.putative.txt, matches.f.txt
//
// extractAKAZE (AKAZEOpenCV.cpp:116-187) = sfmloc_image_read + sfmloc_akaze_detect_and_compute per image; matchAKAZE /
// trackAKAZE (MatchUtils.cpp:73-277) = sfmloc_match_pairs / sfmloc_track; geometricMatch (MatchUtils.cpp:372-420) =
// sfmloc_geometric_pairs (with -gm: guided matching under each pair's estimated F, the map builder's default,
// reconstructGraph.py:155-163).  Files that already exist are kept, as in the reference.
// sfmlocalization_amd/extfeat.py is the same tool in Python; the
// two write identical files (tests/test_gpu_extfeat.py).
#include <sys/stat.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sfmloc.h"

namespace {

bool is_number(const std::string &s) {
  char *end = nullptr;
  std::strtod(s.c_str(), &end);
  return !s.empty() && end && *end == '\0';
}

// cv::CommandLineParser syntax: positionals, -k=v / --key=v, bare flags
struct Args {
  std::vector<std::string> pos;
  std::map<std::string, std::string> opt;
  std::string get(std::initializer_list<const char *> names, const char *def) const {
    std::string v = def;
    for (const char *n : names) {
      auto it = opt.find(n);
      if (it != opt.end()) v = it->second;
    }
    return v;
  }
  bool flag(std::initializer_list<const char *> names) const {
    std::string v = get(names, "false");
    for (char &c : v) c = (char)tolower(c);
    return v == "1" || v == "true" || v == "yes";
  }
};

Args parse_args(int argc, char **argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    if (s.size() > 1 && s[0] == '-' && !is_number(s)) {
      size_t p = s.find_first_not_of('-');
      std::string kv = s.substr(p == std::string::npos ? s.size() : p);
      size_t eq = kv.find('=');
      if (eq == std::string::npos)
        a.opt[kv] = "true";
      else
        a.opt[kv.substr(0, eq)] = kv.substr(eq + 1);
    } else {
      a.pos.push_back(s);
    }
  }
  return a;
}

std::string join(const std::string &a, const std::string &b) {
  if (a.empty()) return b;
  return a.back() == '/' ? a + b : a + "/" + b;
}
std::string stem_of(const std::string &p) {
  size_t s = p.find_last_of('/');
  std::string b = s == std::string::npos ? p : p.substr(s + 1);
  size_t d = b.find_last_of('.');
  return d == std::string::npos ? b : b.substr(0, d);
}
bool exists(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}

// shortest text that reads back as the same double (what the Python mirror writes into image_describer.txt)
std::string shortest_double(double v) {
  char b[64];
  for (int prec = 1; prec <= 17; ++prec) {
    snprintf(b, sizeof(b), "%.*g", prec, v);
    if (strtod(b, nullptr) == v) break;
  }
  std::string s = b;
  if (s.find_first_of(".en") == std::string::npos) s += ".0";
  return s;
}

struct View {
  uint32_t id = 0, w = 0, h = 0;
  std::string image, feat, desc;
};

// .desc: [u64 N][N x 64 B]; .feat: "x y size angle" per line (AKAZEOpenCV.cpp:80-111, FileUtils.cpp:77-103)
bool write_features(const View &v, const float *kp, const uint8_t *desc, uint32_t n) {
  std::ofstream ff(v.feat);
  FILE *fd = fopen(v.desc.c_str(), "wb");
  if (!ff.is_open() || !fd) {
    if (fd) fclose(fd);
    return false;
  }
  for (uint32_t i = 0; i < n; ++i)
    ff << kp[6 * i] << " " << kp[6 * i + 1] << " " << kp[6 * i + 2] << " " << kp[6 * i + 3] << "\n";
  const uint64_t n64 = n;
  bool ok = fwrite(&n64, 8, 1, fd) == 1 && (n == 0 || fwrite(desc, 64, n, fd) == n);
  fclose(fd);
  return ok && ff.good();
}

bool read_desc(const std::string &p, std::vector<uint8_t> *rows) {
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return false;
  uint64_t n = 0;
  bool ok = fread(&n, 8, 1, f) == 1 && n < (1ull << 32);
  if (ok) {
    const size_t at = rows->size();
    rows->resize(at + (size_t)n * 64);
    ok = n == 0 || fread(rows->data() + at, 64, (size_t)n, f) == (size_t)n;
  }
  fclose(f);
  return ok;
}

bool read_feat(const std::string &p, std::vector<float> *xy, size_t *n_rows) {
  FILE *f = fopen(p.c_str(), "r");
  if (!f) return false;
  double x, y, s, a;
  *n_rows = 0;
  while (fscanf(f, "%lf %lf %lf %lf", &x, &y, &s, &a) == 4) {
    xy->push_back((float)x);
    xy->push_back((float)y);
    ++*n_rows;
  }
  fclose(f);
  return true;
}

typedef std::map<std::pair<uint32_t, uint32_t>, std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> PairMatches;

// PairWiseMatches text: "I J\nN\n" then N lines "i j" (matching/indMatch_utils.hpp PairedIndMatchToStream)
bool write_matches(const std::string &path, const PairMatches &m) {
  FILE *f = fopen(path.c_str(), "w");
  if (!f) return false;
  for (const auto &kv : m) {
    fprintf(f, "%u %u\n%zu\n", kv.first.first, kv.first.second, kv.second.first.size());
    for (size_t k = 0; k < kv.second.first.size(); ++k) fprintf(f, "%u %u\n", kv.second.first[k], kv.second.second[k]);
  }
  return fclose(f) == 0;
}

bool read_matches(const std::string &path, PairMatches *m) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return false;
  unsigned a, b;
  unsigned long n;
  while (fscanf(f, "%u %u %lu", &a, &b, &n) == 3) {
    auto &e = (*m)[std::make_pair(a, b)];
    for (unsigned long k = 0; k < n; ++k) {
      unsigned i, j;
      if (fscanf(f, "%u %u", &i, &j) != 2) break;
      e.first.push_back(i);
      e.second.push_back(j);
    }
  }
  fclose(f);
  return true;
}

// sfmloc_matches (view INDICES) -> PairMatches keyed by view ids
bool take_matches(sfmloc_matches *h, const std::vector<View> &views, PairMatches *out) {
  const uint32_t np = sfmloc_matches_pairs(h);
  bool ok = true;
  for (uint32_t k = 0; k < np && ok; ++k) {
    uint32_t a = 0, b = 0, n = 0;
    ok = sfmloc_matches_pair(h, k, &a, &b, &n) == 0 && a < views.size() && b < views.size();
    if (!ok) break;
    auto &e = (*out)[std::make_pair(views[a].id, views[b].id)];
    e.first.resize(n);
    e.second.resize(n);
    if (n) ok = sfmloc_matches_read(h, k, e.first.data(), e.second.data(), n) == 0;
  }
  sfmloc_matches_destroy(h);
  return ok;
}

typedef std::vector<std::pair<uint32_t, uint32_t>> PairList;

PairList generate_all_pairs(const std::vector<uint32_t> &ids) {  // SfMDataUtils.cpp:128-141
  PairList p;
  for (size_t i = 0; i < ids.size(); ++i)
    for (size_t j = i + 1; j < ids.size(); ++j) p.emplace_back(ids[i], ids[j]);
  return p;
}

PairList generate_video_match_pairs(const std::vector<uint32_t> &ids, int frame) {  // SfMDataUtils.cpp:144-157
  PairList p;
  for (size_t i = 0; i < ids.size(); ++i)
    for (size_t j = i + 1; j < std::min(ids.size(), i + (size_t)frame + 1); ++j) p.emplace_back(ids[i], ids[j]);
  return p;
}

// SfMDataUtils.cpp:168-187, literally: scanning from the back, earlier entries get ordered (first <= second) as a side
// effect, a later duplicate of an earlier pair is dropped
void remove_dup_pairs(PairList *pairs) {
  std::vector<size_t> dup;
  for (size_t i = pairs->size(); i-- > 1;)
    for (size_t j = i; j-- > 0;) {
      auto &pj = (*pairs)[j];
      if (pj.first > pj.second) std::swap(pj.first, pj.second);
      const auto &pi = (*pairs)[i];
      if (pi == pj || (pi.first == pj.second && pi.second == pj.first)) {
        dup.push_back(i);
        break;
      }
    }
  for (size_t i : dup) pairs->erase(pairs->begin() + (long)i);  // descending indices
}

PairList read_pair_file(const std::string &path) {  // FileUtils.cpp:180-194
  PairList p;
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return p;
  char line[512];
  while (fgets(line, sizeof(line), f)) {
    long a, b;
    if (sscanf(line, "%ld %ld", &a, &b) == 2) p.emplace_back((uint32_t)a, (uint32_t)b);
  }
  fclose(f);
  return p;
}

}  // namespace

int main(int argc, char **argv) {
  const Args a = parse_args(argc, argv);
  if (a.pos.empty() || a.opt.count("h") || a.opt.count("help")) {
    printf("usage: ExtFeatAndMatch <matchdir> [-c=3 -t=0.001 -o=4 -l=4 -f=0.6 -r=4096 -v=0 -p= -mf=0 -mm=60 -g=4.0 -gm -sm]\n");
    return 1;
  }
  // computeFeaturesAndMatches.cpp:49-64
  const int akaze_ch = atoi(a.get({"c", "akazeChannel"}, "3").c_str());
  const double akaze_thres = atof(a.get({"t", "akazeThreshold"}, "0.001").c_str());
  const int n_oct = atoi(a.get({"o", "akazeNOctave"}, "4").c_str());
  const int n_oct_lay = atoi(a.get({"l", "akazeOctaveLayer"}, "4").c_str());
  const double ratio = atof(a.get({"f", "fdistratio"}, "0.6").c_str());
  const int ransac_round = atoi(a.get({"r", "ransacround"}, "4096").c_str());
  const int video_frame = atoi(a.get({"v", "videoMatchFrame"}, "0").c_str());
  const std::string pair_file = a.get({"p", "pairfile"}, "");
  const int max_frame_dist = atoi(a.get({"mf", "maxFrameDist"}, "0").c_str());
  const size_t min_match = (size_t)atoi(a.get({"mm", "minMatch"}, "60").c_str());
  const int geom_error = (int)atof(a.get({"g", "geomError"}, "4.0").c_str());  // parsed as int (:92)
  const bool guided = a.flag({"gm", "guidedMatch"}), skip_matching = a.flag({"sm", "skipMathing"});
  const int device = atoi(a.get({"device"}, "0").c_str());
  const std::string match_dir = a.pos[0];
  if ((!pair_file.empty()) + (video_frame > 0) + (max_frame_dist > 0) > 1) {  // CV_Assert at :103-106
    fprintf(stderr, "pair file, video match frames and track length are mutually exclusive\n");
    return 1;
  }
  printf("Matches directory : %s\n", match_dir.c_str());
  const std::string sd_path = join(match_dir, "sfm_data.json");
  sfmloc_view_list *vl = nullptr;
  uint32_t n_views = 0;
  if (sfmloc_view_list_open(sd_path.c_str(), &vl, &n_views)) {
    fprintf(stderr, "Cannot load %s\n", sd_path.c_str());
    return 1;
  }
  std::vector<View> views(n_views);
  std::vector<uint32_t> ids(n_views);
  for (uint32_t k = 0; k < n_views; ++k) {
    const char *img = nullptr;
    sfmloc_view_list_get(vl, k, &views[k].id, &views[k].w, &views[k].h, &img);
    views[k].image = img;
    views[k].feat = join(match_dir, stem_of(views[k].image) + ".feat");
    views[k].desc = join(match_dir, stem_of(views[k].image) + ".desc");
    ids[k] = views[k].id;
  }
  sfmloc_view_list_close(vl);
  if (akaze_ch != 3) {
    fprintf(stderr, "only 3-channel M-LDB (486 bits) is implemented\n");
    return 1;
  }
  {  // AKAZEOption::write (AKAZEOption.cpp:31-41)
    FILE *f = fopen(join(match_dir, "image_describer.txt").c_str(), "w");
    if (f) {
      fprintf(f, "%%YAML:1.0\ndesc_ch: %d\nthres: %s\nnOct: %d\nnOctLay: %d\n", akaze_ch, shortest_double(akaze_thres).c_str(),
              n_oct, n_oct_lay);
      fclose(f);
    }
  }
  // hulo::extractAKAZE (AKAZEOpenCV.cpp:116-187): every view whose .feat or .desc is missing
  {
    std::map<std::pair<int, int>, sfmloc_akaze *> extractors;
    const uint32_t cap = 65536;
    std::vector<float> kp((size_t)cap * 6);
    std::vector<uint8_t> desc((size_t)cap * 64), gray;
    bool failed = false;
    for (const View &v : views) {
      if (exists(v.feat) && exists(v.desc)) continue;
      int32_t w = 0, h = 0;
      if (sfmloc_image_read(v.image.c_str(), 0, nullptr, 0, &w, &h)) {
        fprintf(stderr, "cannot open file to write features for %s\n", v.image.c_str());
        continue;
      }
      gray.resize((size_t)w * h);
      if (sfmloc_image_read(v.image.c_str(), 0, gray.data(), gray.size(), &w, &h)) {
        fprintf(stderr, "cannot open file to write features for %s\n", v.image.c_str());
        continue;
      }
      sfmloc_akaze *&ak = extractors[std::make_pair(w, h)];
      uint32_t n = 0;
      if ((!ak && sfmloc_akaze_create(device, w, h, n_oct, n_oct_lay, (float)akaze_thres, &ak)) ||
          sfmloc_akaze_detect_and_compute(ak, gray.data(), kp.data(), desc.data(), cap, &n)) {
        fprintf(stderr, "%s\n", sfmloc_last_error());
        failed = true;
        break;
      }
      if (!write_features(v, kp.data(), desc.data(), n))
        fprintf(stderr, "cannot open file to write features for %s\n", v.image.c_str());
    }
    for (auto &kv : extractors)
      if (kv.second) sfmloc_akaze_destroy(kv.second);
    if (failed) return 1;
  }
  if (skip_matching) {
    printf("Skip matching option is set. Exit without feature matching.\n");
    return 1;  // the reference returns 1 here (:147)
  }

  // the descriptor bank of the image set, in view order
  std::vector<uint8_t> bank;
  std::vector<float> kxy;
  std::vector<uint32_t> off(n_views + 1, 0), wh(2 * (size_t)n_views);
  for (uint32_t k = 0; k < n_views; ++k) {
    size_t n_feat = 0;
    const size_t before = bank.size();
    if (!read_desc(views[k].desc, &bank) || !read_feat(views[k].feat, &kxy, &n_feat) ||
        (bank.size() - before) / 64 != n_feat) {
      fprintf(stderr, "cannot read the features of view %u (%s)\n", views[k].id, views[k].desc.c_str());
      return 1;
    }
    off[k + 1] = (uint32_t)(bank.size() / 64);
    wh[2 * k] = views[k].w;
    wh[2 * k + 1] = views[k].h;
  }
  std::map<uint32_t, uint32_t> idx_of;
  for (uint32_t k = 0; k < n_views; ++k) idx_of[ids[k]] = k;

  sfmloc_params prm;
  sfmloc_default_params(&prm);
  prm.dist_ratio = (float)ratio;
  prm.ransac_round = ransac_round;
  prm.geom_precision = (double)geom_error;
  prm.device = device;
  prm.guided_matching = guided ? 1 : 0;  // -gm: geometricMatch(..., bGuided_matching) (computeFeaturesAndMatches.cpp:243)
  sfmloc_map_desc md;
  memset(&md, 0, sizeof(md));
  md.n_views = n_views;
  md.view_id = ids.data();
  md.view_off = off.data();
  md.view_wh = wh.data();
  md.n_rows = bank.size() / 64;
  md.desc = bank.data();
  md.kpt_xy = kxy.data();
  sfmloc_map *map = nullptr;
  if (sfmloc_map_create(&md, &prm, &map)) {
    fprintf(stderr, "%s\n", sfmloc_last_error());
    return 1;
  }
  int rc = 0;
  const std::string put_path = join(match_dir, "matches.putative.txt"), f_path = join(match_dir, "matches.f.txt");
  printf("Start Putative Matching...\n");
  if (!exists(put_path)) {
    sfmloc_matches *h = nullptr;
    if (max_frame_dist != 0) {
      rc = sfmloc_track(map, (uint32_t)max_frame_dist, &h);
    } else {
      printf("Generating pairs\n");
      PairList pairs;
      if (!pair_file.empty()) {
        pairs = read_pair_file(pair_file);
      } else if (video_frame > 0) {
        pairs = generate_video_match_pairs(ids, video_frame);
        remove_dup_pairs(&pairs);
      } else {
        pairs = generate_all_pairs(ids);
      }
      for (size_t k = 0; k < pairs.size(); ++k) printf("%s(%u %u)", k ? " " : "", pairs[k].first, pairs[k].second);
      printf("\nTotal number of pairs : %zu\n", pairs.size());
      std::vector<uint32_t> flat;
      for (const auto &p : pairs)
        if (idx_of.count(p.first) && idx_of.count(p.second)) {
          flat.push_back(idx_of[p.first]);
          flat.push_back(idx_of[p.second]);
        }
      uint32_t dummy[2] = {0, 0};
      rc = sfmloc_match_pairs(map, flat.empty() ? dummy : flat.data(), (uint32_t)(flat.size() / 2), &h);
    }
    PairMatches put;
    if (rc || !take_matches(h, views, &put) || !write_matches(put_path, put)) {
      fprintf(stderr, "putative matching failed: %s\n", sfmloc_last_error());
      sfmloc_map_destroy(map);
      return 1;
    }
  }
  printf("Start Geometric Matching...\n");
  if (!exists(f_path)) {
    PairMatches put;
    read_matches(put_path, &put);
    std::vector<uint32_t> pairs, mi, mj;
    std::vector<uint64_t> offsets(1, 0);
    size_t n_kept = 0;
    for (const auto &kv : put) {
      if (kv.second.first.size() < min_match) {  // :211-221
        printf("%u,%u,%zu ", kv.first.first, kv.first.second, kv.second.first.size());
        continue;
      }
      if (!idx_of.count(kv.first.first) || !idx_of.count(kv.first.second)) continue;
      pairs.push_back(idx_of[kv.first.first]);
      pairs.push_back(idx_of[kv.first.second]);
      mi.insert(mi.end(), kv.second.first.begin(), kv.second.first.end());
      mj.insert(mj.end(), kv.second.second.begin(), kv.second.second.end());
      offsets.push_back(mi.size());
      ++n_kept;
    }
    printf("\n");
    sfmloc_matches *h = nullptr;
    uint32_t dummy[2] = {0, 0};
    rc = sfmloc_geometric_pairs(map, pairs.empty() ? dummy : pairs.data(), (uint32_t)n_kept, offsets.data(),
                                mi.empty() ? dummy : mi.data(), mj.empty() ? dummy : mj.data(), &h);
    PairMatches geo;
    if (rc || !take_matches(h, views, &geo)) {
      fprintf(stderr, "geometric matching failed: %s\n", sfmloc_last_error());
      sfmloc_map_destroy(map);
      return 1;
    }
    printf("number of putative matches : %zu\n", n_kept);
    printf("number of geometric matches : %zu\n", geo.size());
    if (!write_matches(f_path, geo)) rc = 1;
  }
  sfmloc_map_destroy(map);
  return rc ? 1 : 0;
}
