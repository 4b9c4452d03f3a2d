// OpenMVGLocalization_AKAZE, the reference's localisation tool (OpenMVGLocalization_AKAZE/src/localization.cpp:64-590),
// as a C++ host program over the C ABI of libsfmloc_hip.so -- the reference's own host language for this path.
//
//   OpenMVGLocalization_AKAZE <queryImage|dir> <sfmDataDir> <matchDir> <outputFolder>
//        [-f=0.6] [-r=200] [-k=0] [-x= -y= -z= -d=-1] [-a=BOWfile.yml] [-p=PCAfile.yml] [-i=1] [-g=4.0] [--featdir=DIR]
//
// Same arguments, console messages and output files (<outputFolder>/<base>.json with the reference's keys and
// Eigen IOFormat(6) matrices; the failure form has the first three keys only, localization.cpp:84-153).
// Images are decoded by the library's own cv::imread (sfmloc_image_read: JPEG, PNG, binary PGM/PPM); when the image
// cannot be decoded the query's features are taken from <featdir>/<base>.desc/.feat (the files extractAKAZESingleImg writes, AKAZEOpenCV.cpp:80-111); the BoW shortlist
// (-k) uses <featdir>/<base>.bow if present, else the dense-feature chain on the decoded colour image; -w writes
// <matchDir>/matches.fQ.txt as the reference does; -gm = guided matching in the geometric stage (:82,183,451).  The Python mirror (sfmlocalization_amd/engine.py) covers the same contract.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sfmloc.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// cv::CommandLineParser syntax: positionals, -k=v / --key=v, bare flags
// ---------------------------------------------------------------------------------------------------------
bool is_number(const std::string &s) {
  char *end = nullptr;
  std::strtod(s.c_str(), &end);
  return !s.empty() && end && *end == '\0';
}

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

// ---------------------------------------------------------------------------------------------------------
// paths and files
// ---------------------------------------------------------------------------------------------------------
std::string join(const std::string &a, const std::string &b) {
  if (a.empty()) return b;
  return a.back() == '/' ? a + b : a + "/" + b;
}
std::string basename_of(const std::string &p) {
  size_t s = p.find_last_of('/');
  return s == std::string::npos ? p : p.substr(s + 1);
}
std::string dirname_of(const std::string &p) {
  size_t s = p.find_last_of('/');
  return s == std::string::npos ? std::string("") : p.substr(0, s);
}
std::string stem_of(const std::string &p) {
  std::string b = basename_of(p);
  size_t d = b.find_last_of('.');
  return d == std::string::npos ? b : b.substr(0, d);
}
std::string ext_of(const std::string &p) {
  std::string b = basename_of(p);
  size_t d = b.find_last_of('.');
  return d == std::string::npos ? std::string("") : b.substr(d + 1);
}
bool is_file(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
bool is_dir(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
bool image_ext(const std::string &e) {  // localization.cpp:204-206
  return e == "jpg" || e == "JPG" || e == "jpeg" || e == "JPEG" || e == "png" || e == "PNG";
}
bool read_all(const std::string &p, std::vector<uint8_t> *out) {
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize(n > 0 ? (size_t)n : 0);
  const bool ok = n >= 0 && fread(out->data(), 1, out->size(), f) == out->size();
  fclose(f);
  return ok;
}

// .desc: [u64 N][N x 64 B]  (FileUtils.cpp:77-103)
bool read_desc(const std::string &p, std::vector<uint8_t> *rows) {
  std::vector<uint8_t> raw;
  if (!read_all(p, &raw) || raw.size() < 8) return false;
  uint64_t n;
  memcpy(&n, raw.data(), 8);
  if (raw.size() != 8 + n * 64) return false;
  rows->assign(raw.begin() + 8, raw.end());
  return true;
}
// .feat: "x y size angle" per line (AKAZEOpenCV.cpp:80-81)
bool read_feat(const std::string &p, std::vector<float> *xy) {
  FILE *f = fopen(p.c_str(), "r");
  if (!f) return false;
  xy->clear();
  double x, y, s, a;
  while (fscanf(f, "%lf %lf %lf %lf", &x, &y, &s, &a) == 4) {
    xy->push_back((float)x);
    xy->push_back((float)y);
  }
  fclose(f);
  return true;
}
// .bow: [i32 rows][i32 cols][i32 type][data], CV_64F or CV_32F (FileUtils.cpp:43-75)
bool read_bow(const std::string &p, std::vector<float> *v) {
  std::vector<uint8_t> raw;
  if (!read_all(p, &raw) || raw.size() < 12) return false;
  int32_t r, c, t;
  memcpy(&r, raw.data(), 4);
  memcpy(&c, raw.data() + 4, 4);
  memcpy(&t, raw.data() + 8, 4);
  const size_t n = (size_t)r * c;
  v->resize(n);
  if (t == 6 && raw.size() == 12 + 8 * n) {
    for (size_t i = 0; i < n; ++i) {
      double d;
      memcpy(&d, raw.data() + 12 + 8 * i, 8);
      (*v)[i] = (float)d;
    }
    return true;
  }
  if (t == 5 && raw.size() == 12 + 4 * n) {
    memcpy(v->data(), raw.data() + 12, 4 * n);
    return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------------------
// image decoding = the library's cv::imread (sfmloc_image_read: JPEG, PNG, binary PGM/PPM)
// ---------------------------------------------------------------------------------------------------------
bool load_image(const std::string &path, bool color, std::vector<uint8_t> *px, int *w, int *h) {
  int32_t iw = 0, ih = 0;
  if (sfmloc_image_read(path.c_str(), color ? 1 : 0, nullptr, 0, &iw, &ih)) return false;
  px->resize((size_t)iw * ih * (color ? 3 : 1));
  if (sfmloc_image_read(path.c_str(), color ? 1 : 0, px->data(), px->size(), &iw, &ih)) return false;
  *w = iw;
  *h = ih;
  return true;
}

// image_describer.txt (cv::FileStorage YAML, AKAZEOption.cpp:44-55; defaults AKAZEOption.h:31-34)
struct AkazeOption {
  int desc_ch = 3, nOct = 4, nOctLay = 4;
  float thres = 0.001f;
};
AkazeOption read_image_describer(const std::string &p) {
  AkazeOption o;
  FILE *f = fopen(p.c_str(), "r");
  if (!f) return o;
  char line[512];
  while (fgets(line, sizeof(line), f)) {
    char key[128];
    double v;
    if (sscanf(line, " %127[^:]: %lf", key, &v) == 2) {
      if (!strcmp(key, "desc_ch")) o.desc_ch = (int)v;
      if (!strcmp(key, "thres")) o.thres = (float)v;
      if (!strcmp(key, "nOct")) o.nOct = (int)v;
      if (!strcmp(key, "nOctLay")) o.nOctLay = (int)v;
    }
  }
  fclose(f);
  return o;
}

// cv::FileStorage YAML as OpenCV writes BOWfile.yml / PCAfile.yml (TrainBoW): scalars, strings, !!opencv-matrix
struct CvYaml {
  std::map<std::string, double> num;
  std::map<std::string, std::string> str;
  struct Mat {
    int rows = 0, cols = 0;
    std::vector<float> data;
  };
  std::map<std::string, Mat> mat;
};
bool read_cv_yaml(const std::string &path, CvYaml *y) {
  std::vector<uint8_t> raw;
  if (!read_all(path, &raw)) return false;
  std::vector<std::string> lines;  // (a matrix may sit on one very long line)
  for (size_t a = 0; a < raw.size();) {
    size_t b = a;
    while (b < raw.size() && raw[b] != '\n') ++b;
    std::string l(raw.begin() + a, raw.begin() + b);
    while (!l.empty() && l.back() == '\r') l.pop_back();
    lines.push_back(l);
    a = b + 1;
  }
  auto trim = [](std::string v) {
    const size_t a = v.find_first_not_of(" \t"), b = v.find_last_not_of(" \t");
    return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
  };
  for (size_t i = 0; i < lines.size();) {
    const std::string ln = lines[i++];
    if (ln.empty() || ln[0] == '%' || ln[0] == ' ' || ln[0] == '\t' || ln.compare(0, 3, "---") == 0) continue;
    const size_t c = ln.find(':');
    if (c == std::string::npos) continue;
    const std::string key = trim(ln.substr(0, c)), rest = trim(ln.substr(c + 1));
    if (rest.compare(0, 15, "!!opencv-matrix") == 0) {
      CvYaml::Mat m;
      std::string body;
      while (i < lines.size() && !lines[i].empty() && (lines[i][0] == ' ' || lines[i][0] == '\t')) {
        const std::string s = trim(lines[i++]);
        if (s.compare(0, 5, "rows:") == 0) m.rows = atoi(s.c_str() + 5);
        else if (s.compare(0, 5, "cols:") == 0) m.cols = atoi(s.c_str() + 5);
        else if (s.compare(0, 3, "dt:") == 0) {}
        else if (s.compare(0, 5, "data:") == 0) body = s.substr(5);
        else body += " " + s;
      }
      for (char &ch : body)
        if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
      const char *p = body.c_str();
      char *e = nullptr;
      for (double v = strtod(p, &e); p != e; v = strtod(p, &e)) {
        m.data.push_back((float)v);
        p = e;
      }
      if ((size_t)m.rows * m.cols != m.data.size()) return false;
      y->mat[key] = m;
    } else if (!rest.empty() && rest[0] == '"') {
      y->str[key] = rest.substr(1, rest.size() >= 2 ? rest.size() - 2 : 0);
    } else {
      y->num[key] = atof(rest.c_str());
      y->str[key] = rest;
    }
  }
  return true;
}

// The query-side BoW vector (DenseLocalFeatureWrapper -> PcaWrapper -> BoFSpatialPyramids, localization.cpp:346-361):
// colour image -> 300x300 gray -> AKAZE descriptors at the dense grid (cv::AKAZE::create() defaults) as floats -> PCA +
// BoF, as ONE resident chain on the device (sfmloc_imgbow: one extractor per image size, nothing but the image and the
// vector crosses PCIe)
struct DenseBow {
  CvYaml b, p;          // the model files' contents (the descriptor below points into them)
  sfmloc_bof_desc d;
  int device = 0;
  bool ready = false;
  std::map<std::pair<int, int>, sfmloc_imgbow *> by_size;
  bool init(const std::string &bow_file, const std::string &pca_file, int dev) {
    device = dev;
    if (!read_cv_yaml(bow_file, &b) || !b.mat.count("Centers")) return false;
    memset(&d, 0, sizeof(d));
    const CvYaml::Mat &cen = b.mat["Centers"];
    d.K = cen.rows;
    d.in_dim = 61;
    d.centers = cen.data.data();
    d.resized_image_size = b.num.count("ResizedImageSize") ? (int)b.num["ResizedImageSize"] : 300;
    d.use_spatial_pyramid = b.num.count("UseSpatialPyramid") ? (int)b.num["UseSpatialPyramid"] : 1;
    d.pyramid_level = b.num.count("PyramidLevel") ? (int)b.num["PyramidLevel"] : 2;
    const std::string norm = b.str.count("NormBofFeatureType") ? b.str["NormBofFeatureType"] : "L1";
    d.norm_type = norm == "NONE" ? 0 : norm == "L2" ? 1 : 2;
    if (!pca_file.empty()) {
      if (!read_cv_yaml(pca_file, &p) || !p.mat.count("EigenVectorsPCA") || !p.mat.count("EigenValuesPCA") ||
          !p.mat.count("MeanPCA"))
        return false;
      d.n_pca = (int)p.num["DimPCA"];
      d.pca_mean = p.mat["MeanPCA"].data.data();
      d.pca_eigvec = p.mat["EigenVectorsPCA"].data.data();  // first n_pca rows of [in_dim x in_dim]
      d.pca_eigval = p.mat["EigenValuesPCA"].data.data();
    }
    ready = true;
    return true;
  }
  bool compute(const std::vector<uint8_t> &bgr, int w, int h, std::vector<float> *out) {
    if (!ready) return false;
    sfmloc_imgbow *&ib = by_size[std::make_pair(w, h)];
    if (!ib && sfmloc_imgbow_create(&d, device, (uint32_t)w, (uint32_t)h, 3, &ib)) return false;
    std::vector<double> bow((size_t)sfmloc_imgbow_dim(ib));
    if (sfmloc_imgbow_compute(ib, bgr.data(), nullptr, bow.data())) return false;
    out->assign(bow.begin(), bow.end());
    return true;
  }
  ~DenseBow() {
    for (auto &kv : by_size)
      if (kv.second) sfmloc_imgbow_destroy(kv.second);
  }
};


// Eigen IOFormat(6, 0, ",", ",\n", rowPrefix, rowSuffix, "[", "]"): 6 significant digits, columns aligned
std::string eigen_format(const double *m, int rows, int cols, const char *row_prefix, const char *row_suffix) {
  std::vector<std::string> cells((size_t)rows * cols);
  size_t width = 0;
  for (int i = 0; i < rows * cols; ++i) {
    char b[64];
    snprintf(b, sizeof(b), "%.6g", m[i]);
    cells[i] = b;
    width = std::max(width, cells[i].size());
  }
  std::string s = "[";
  for (int r = 0; r < rows; ++r) {
    if (r) s += ",\n";
    s += row_prefix;
    for (int c = 0; c < cols; ++c) {
      if (c) s += ",";
      const std::string &cell = cells[(size_t)r * cols + c];
      s += std::string(width - cell.size(), ' ') + cell;
    }
    s += row_suffix;
  }
  return s + "]";
}

// saveResultJson (localization.cpp:84-153)
bool write_result_json(const std::string &out_dir, const std::string &query, const std::string &sfm_json,
                       const std::string &match_dir, const sfmloc_pose *pose, const uint32_t *pq, const uint32_t *pl) {
  const std::string path = join(out_dir, stem_of(query) + ".json");
  FILE *f = fopen(path.c_str(), "w");
  if (!f) return false;
  fprintf(f, "{\n\t\"filename\": \"%s\",\n\t\"sfm_data\": \"%s\",\n", query.c_str(), sfm_json.c_str());
  if (!pose) {
    fprintf(f, "\t\"matches_dir\": \"%s\"\n}\n", match_dir.c_str());
  } else {
    fprintf(f, "\t\"matches_dir\": \"%s\",\n", match_dir.c_str());
    fprintf(f, "\t\"K\": %s,\n", eigen_format(pose->K, 3, 3, "[", "]").c_str());
    fprintf(f, "\t\"R\": %s,\n", eigen_format(pose->R, 3, 3, "[", "]").c_str());
    fprintf(f, "\t\"t\": %s,\n", eigen_format(pose->center, 3, 1, "", "").c_str());
    fprintf(f, "\t\"pair\": [");
    for (int i = 0; i < pose->n_inliers; ++i) fprintf(f, "%s[%u,%u]", i ? "," : "", pq[i], pl[i]);
    fprintf(f, "]\n}\n");
  }
  fclose(f);
  return true;
}

float round6(float v) {  // what the reference reads back from the .feat it wrote (6 significant digits)
  char b[64];
  snprintf(b, sizeof(b), "%.6g", (double)v);
  return strtof(b, nullptr);
}

}  // namespace

int main(int argc, char **argv) {
  const Args a = parse_args(argc, argv);
  if (a.pos.size() < 4 || a.opt.count("h") || a.opt.count("help")) {
    printf("usage: OpenMVGLocalization_AKAZE <queryImage|dir> <sfmDataDir> <matchDir> <outputFolder> [-f=0.6] [-r=200] "
           "[-k=0] [-x= -y= -z= -d=-1] [-i=1] [-g=4.0] [--featdir=DIR]\n");
    return 1;
  }
  const std::string query = a.pos[0], sfm_dir = a.pos[1], match_dir = a.pos[2], out_dir = a.pos[3];
  const double f_ratio = atof(a.get({"f", "fDistRatio"}, "0.6").c_str());
  const int ransac_round = atoi(a.get({"r", "ransacRound"}, "200").c_str());
  const int knn_bow = atoi(a.get({"k", "knnbow"}, "0").c_str());
  const double cx = atof(a.get({"x", "cenLocX"}, "0.0").c_str()), cy = atof(a.get({"y", "cenLocY"}, "0.0").c_str()),
               cz = atof(a.get({"z", "cenLocZ"}, "0.0").c_str()), radius = atof(a.get({"d", "cenRadius"}, "-1.0").c_str());
  const std::string bow_model = a.get({"a", "bowModelFile"}, "");
  const std::string pca_model = a.get({"p", "pcaModelFile"}, "");
  int every = atoi(a.get({"i", "locEvryNFrame"}, "1").c_str());
  const double geom = atof(a.get({"g", "geomLimit"}, "4.0").c_str());
  std::string wm = a.get({"w", "writematch"}, "false");
  for (char &ch : wm) ch = (char)tolower(ch);
  const bool write_match = wm == "1" || wm == "true" || wm == "yes";
  std::string gm = a.get({"gm", "guidedMatch"}, "false");   // localization.cpp:82,183
  for (char &ch : gm) ch = (char)tolower(ch);
  const bool guided = gm == "1" || gm == "true" || gm == "yes";
  const std::string featdir_opt = a.get({"featdir"}, "");
  const int device = atoi(a.get({"device"}, "0").c_str());
  if (every <= 0) every = 1;

  printf("Start localizing input image.\n");
  std::vector<std::string> images;
  if (is_file(query)) {
    if (!image_ext(ext_of(query))) {
      printf("Input image is not JPEG or PNG file\n");
      return 1;
    }
    images.push_back(query);
  } else if (is_dir(query)) {
    if (DIR *d = opendir(query.c_str())) {
      while (dirent *e = readdir(d))
        if (image_ext(ext_of(e->d_name))) images.push_back(join(query, e->d_name));
      closedir(d);
    }
    std::sort(images.begin(), images.end());
    if (images.empty()) {
      printf("JPEG or PNG file is not found in input image directory\n");
      return 1;
    }
  } else {
    images.push_back(query);  // with precomputed features the image itself may be absent
  }

  const std::string sfm_json = is_file(sfm_dir) ? sfm_dir : join(sfm_dir, "sfm_data.json");
  sfmloc_params prm;
  sfmloc_default_params(&prm);
  prm.dist_ratio = (float)f_ratio;
  prm.ransac_round = ransac_round;
  prm.geom_precision = geom;
  prm.bow_knn = knn_bow;
  prm.device = device;
  prm.guided_matching = guided ? 1 : 0;
  sfmloc_map *map = nullptr;
  // <sfmDataDir> may also name a packed map file written by sfmloc_pack (one binary instead of sfm_data.json and the
  // per-view files); the result JSON then cites that file as "sfm_data"
  if (is_file(sfm_dir) ? sfmloc_open_packed(sfm_dir.c_str(), &prm, &map)
                       : sfmloc_open(sfm_dir.c_str(), match_dir.c_str(), &prm, &map)) {
    fprintf(stderr, "%s\n", sfmloc_last_error());
    return 1;
  }
  sfmloc_map_info info;
  sfmloc_map_get_info(map, &info);
  std::vector<uint32_t> view_wh(2 * (size_t)info.n_views), view_id(info.n_views), view_off(info.n_views + 1);
  std::vector<double> centers(3 * (size_t)info.n_views);
  sfmloc_map_view_sizes(map, view_wh.data());
  sfmloc_map_views(map, view_id.data(), view_off.data(), centers.data());
  const AkazeOption ak_opt = read_image_describer(join(match_dir, "image_describer.txt"));
  mkdir(out_dir.c_str(), 0777);
  // the query's view index in the reference's match files: id of the LAST view of sfm_data (posed or not) + 1
  // (localization.cpp:371); a packed map only knows its posed views
  uint32_t ind_query_file = info.n_views ? view_id[info.n_views - 1] + 1 : 0;
  if (!is_file(sfm_dir)) {
    sfmloc_view_list *vl = nullptr;
    uint32_t nv_all = 0;
    if (sfmloc_view_list_open(sfm_json.c_str(), &vl, &nv_all) == 0) {
      uint32_t last = 0;
      if (nv_all && sfmloc_view_list_get(vl, nv_all - 1, &last, nullptr, nullptr, nullptr) == 0) ind_query_file = last + 1;
      sfmloc_view_list_close(vl);
    }
  }

  std::map<std::pair<int, int>, sfmloc_akaze *> extractors;
  DenseBow dense;
  int n_img = 0, match_next = 0, rc_all = 0;
  for (const std::string &img : images) {
    ++n_img;
    if (n_img % every == 0) {  // localization.cpp:289-298
    } else if (match_next <= 0) {
      continue;
    } else {
      --match_next;
    }
    const std::string base = stem_of(img);
    const std::string fdir = featdir_opt.empty() ? dirname_of(img) : featdir_opt;
    int w = info.n_views ? (int)view_wh[0] : 0, h = info.n_views ? (int)view_wh[1] : 0;
    std::vector<uint8_t> desc;
    std::vector<float> xy;
    std::vector<uint8_t> gray, bgr;
    int gw = 0, gh = 0;
    const bool have_img = load_image(img, false, &gray, &gw, &gh);  // imread(IMREAD_GRAYSCALE), AKAZEOpenCV.cpp:60
    if (have_img) {
      w = gw;
      h = gh;
    }
    if (!(read_desc(join(fdir, base + ".desc"), &desc) && read_feat(join(fdir, base + ".feat"), &xy) &&
          xy.size() == desc.size() / 64 * 2)) {
      if (!have_img) {
        fprintf(stderr, "cannot decode %s (PNG and PGM/PPM only in this build) nor read precomputed features for it\n",
                img.c_str());
        write_result_json(out_dir, img, sfm_json, match_dir, nullptr, nullptr, nullptr);
        continue;
      }
      if (ak_opt.desc_ch != 3) {
        fprintf(stderr, "only the 3-channel M-LDB descriptor the reference uses is implemented\n");
        rc_all = 1;
        break;
      }
      printf("Extract features from query image\n");  // localization.cpp:313
      sfmloc_akaze *&ak = extractors[{w, h}];
      if (!ak && sfmloc_akaze_create(device, w, h, ak_opt.nOct, ak_opt.nOctLay, ak_opt.thres, &ak)) {
        fprintf(stderr, "%s\n", sfmloc_last_error());
        rc_all = 1;
        break;
      }
      const uint32_t cap = 65536;
      std::vector<float> kp((size_t)cap * 6);
      desc.assign((size_t)cap * 64, 0);
      uint32_t n = 0;
      if (sfmloc_akaze_detect_and_compute(ak, gray.data(), kp.data(), desc.data(), cap, &n)) {
        fprintf(stderr, "%s\n", sfmloc_last_error());
        rc_all = 1;
        break;
      }
      desc.resize((size_t)n * 64);
      xy.resize((size_t)n * 2);
      for (uint32_t i = 0; i < n; ++i) {
        xy[2 * i] = round6(kp[6 * i]);
        xy[2 * i + 1] = round6(kp[6 * i + 1]);
      }
    }
    const uint32_t nq = (uint32_t)(desc.size() / 64);

    // view pre-selection (localization.cpp:332-361): getLocalViews -- squared distance against the UN-squared radius,
    // as the reference does (SfMDataUtils.cpp:210-227) --, then the BoW shortlist when more views than knn remain
    std::vector<uint32_t> sel;
    bool use_sel = false;
    if (radius > 0) {
      use_sel = true;
      for (uint32_t v = 0; v < info.n_views; ++v) {
        const double dx = centers[3 * v] - cx, dy = centers[3 * v + 1] - cy, dz = centers[3 * v + 2] - cz;
        if (dx * dx + dy * dy + dz * dz <= radius) sel.push_back(v);
      }
    }
    sfmloc_pose pose;
    memset(&pose, 0, sizeof(pose));
    std::vector<uint32_t> pq(65536), pl(65536);  // a query has at most 65 535 features, hence inliers
    bool attempted = false;
    if (!(use_sel && sel.empty())) {
      std::vector<float> bow;
      bool have_bow = false;
      if (knn_bow > 0 && !bow_model.empty()) {
        // a precomputed <base>.bow next to the features if there is one, else from the colour image as the reference does
        have_bow = read_bow(join(fdir, base + ".bow"), &bow);
        if (!have_bow && have_img) {
          if (!dense.ready && !dense.init(bow_model, pca_model, device)) {
            fprintf(stderr, "cannot load the BoW / PCA model (%s)\n", sfmloc_last_error());
            rc_all = 1;
            break;
          }
          int cw = 0, ch = 0;  // imread(IMREAD_COLOR), DenseLocalFeatureWrapper.cpp:85
          have_bow = load_image(img, true, &bgr, &cw, &ch) && dense.compute(bgr, cw, ch, &bow);
        }
      }
      sfmloc_query *q = nullptr;
      if (sfmloc_query_create(map, desc.data(), xy.data(), nq, (uint32_t)w, (uint32_t)h, &q)) {
        // as for any error on ONE image: this image gets the failure form, the run goes on (the reference writes a
        // result file for every image, localization.cpp:441,530)
        fprintf(stderr, "%s: %s\n", img.c_str(), sfmloc_last_error());
        rc_all = 1;
        write_result_json(out_dir, img, sfm_json, match_dir, nullptr, nullptr, nullptr);
        continue;
      }
      // with a BoW vector: shortlist (when more than knn views remain, localization.cpp:346) + path in one call, the
      // shortlist staying on the device
      const uint32_t *selp = use_sel ? sel.data() : nullptr;
      const uint32_t nsel = use_sel ? (uint32_t)sel.size() : 0;
      const int rc = have_bow ? sfmloc_localize_bow(map, q, bow.data(), (uint32_t)knn_bow, selp, nsel, &pose, pq.data(),
                                                    pl.data(), 65536)
                              : sfmloc_localize(map, q, selp, nsel, &pose, pq.data(), pl.data(), 65536);
      sfmloc_query_destroy(q);
      if (rc) {
        // an error on ONE image does not end the run: the reference writes a result file for every image
        // (localization.cpp:441,530); this one gets the failure form and the exit status remembers it
        fprintf(stderr, "%s: %s\n", img.c_str(), sfmloc_last_error());
        rc_all = 1;
        write_result_json(out_dir, img, sfm_json, match_dir, nullptr, nullptr, nullptr);
        continue;
      }
      attempted = true;
    }
    if (!attempted) {  // no view near the given position: nothing was matched
      printf("Not enough putative matches\n");
      write_result_json(out_dir, img, sfm_json, match_dir, nullptr, nullptr, nullptr);
      continue;
    }
    printf("number of putative matches : %d\n", pose.n_putative_views);  // localization.cpp:416
    if (pose.n_putative_views == 0) {
      printf("Not enough putative matches\n");  // :420
      write_result_json(out_dir, img, sfm_json, match_dir, nullptr, nullptr, nullptr);
      continue;
    }
    if (write_match) {
      // -w: exportPairWiseMatches(map_geometricMatches, <matchDir>/matches.fQ.txt) (localization.cpp:452-455); the
      // putative list goes to a per-query folder the reference deletes again (:399-403, :585), so only this file stays.
      // Pair = (view id, id of the last view of sfm_data + 1), matches in AC-RANSAC's inlier order.
      const uint64_t nr = view_off[info.n_views];
      // (with -gm the matches are the guided ones, in ascending map-feature order)
      std::vector<uint32_t> gc(info.n_views), gi(nr), gj(nr);
      if (sfmloc_geometric_read_pairs(map, gc.data(), gi.data(), gj.data(), nr)) {
        fprintf(stderr, "%s\n", sfmloc_last_error());
        rc_all = 1;
        break;
      }
      FILE *fm = fopen(join(match_dir, "matches.fQ.txt").c_str(), "w");
      if (!fm) {
        fprintf(stderr, "Cannot write geometric matches file%s\n", join(match_dir, "matches.fQ.txt").c_str());
      } else {
        for (uint32_t v = 0; v < info.n_views; ++v) {
          if (!gc[v]) continue;
          fprintf(fm, "%u %u\n%u\n", view_id[v], ind_query_file, gc[v]);
          for (uint32_t k = 0; k < gc[v]; ++k) fprintf(fm, "%u %u\n", gi[view_off[v] + k], gj[view_off[v] + k]);
        }
        fclose(fm);
      }
    }
    printf("number of geometric matches : %d\n", pose.n_geometric_views);  // :458
    printf("mapFeatTo3DFeat size = %d\n", pose.n_matches_2d3d);             // :476
    printf("cpt = %d\n", pose.n_matches_2d3d);                              // :502
    if (!pose.ok) {
      printf("Fail to estimate camera matrix\n");  // :512
      printf("#inliers = %d\n", pose.n_inliers);
      write_result_json(out_dir, img, sfm_json, match_dir, nullptr, nullptr, nullptr);
      continue;
    }
    printf("#inliers = %d\n", pose.n_inliers);
    write_result_json(out_dir, img, sfm_json, match_dir, &pose, pq.data(), pl.data());
    match_next = every - 1;
    printf("complete\n");
  }
  for (auto &kv : extractors)
    if (kv.second) sfmloc_akaze_destroy(kv.second);
  sfmloc_map_destroy(map);
  return rc_all;
}
