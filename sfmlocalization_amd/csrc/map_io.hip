// Native loader of the reference's on-disk map (host code only): what OpenMVGLocalization_AKAZE does once at
// start-up (localization.cpp:238-280): Load(sfm_data.json) [cereal JSON], structureToMapViewFeatTo3D
// (SfMDataUtils.cpp:33-46), HuloSfMRegionsProvider::load (all <base>.feat / <base>.desc), plus the optional
// per-view <base>.bow vectors (BoFUtils.cpp:33-42).  The arrays it builds feed sfmloc_map_create.
#include <errno.h>
#include <sys/stat.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <new>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "sfmloc_internal.h"

namespace sfmloc {
namespace {

// ---- a small JSON reader: DOM for small subtrees, streamed over the big top-level arrays ----
struct JVal {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  double num = 0.0;
  bool is_int = false;
  long long inum = 0;
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal *get(const char *k) const {
    for (auto &kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

struct JParser {
  const char *p, *end;
  std::string err;
  int depth = 0;  // value() recurses once per nesting level: a hostile file must not overflow the stack
  static constexpr int kMaxDepth = 64;  // sfm_data.json nests 7 deep
  explicit JParser(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
  }
  bool fail(const char *m) {
    if (err.empty()) err = m;
    return false;
  }
  bool expect(char c) {
    ws();
    if (p < end && *p == c) {
      ++p;
      return true;
    }
    return fail("unexpected character");
  }
  bool peek(char c) {
    ws();
    return p < end && *p == c;
  }
  bool string(std::string &out) {
    ws();
    if (p >= end || *p != '"') return fail("string expected");
    ++p;
    out.clear();
    while (p < end && *p != '"') {
      if (*p == '\\' && p + 1 < end) {
        ++p;
        switch (*p) {
          case 'n': out.push_back('\n'); break;
          case 't': out.push_back('\t'); break;
          case 'r': out.push_back('\r'); break;
          case 'b': out.push_back('\b'); break;
          case 'f': out.push_back('\f'); break;
          case 'u':  // keep the escape verbatim (paths in sfm_data are ASCII in practice)
            out += "\\u";
            break;
          default: out.push_back(*p);
        }
        ++p;
      } else {
        out.push_back(*p++);
      }
    }
    if (p >= end) return fail("unterminated string");
    ++p;
    return true;
  }
  bool value(JVal &v) {
    struct Depth {
      int &d;
      explicit Depth(int &x) : d(x) { ++d; }
      ~Depth() { --d; }
    } guard(depth);
    if (depth > kMaxDepth) return fail("nested too deeply");
    ws();
    if (p >= end) return fail("unexpected end");
    if (*p == '{') {
      ++p;
      v.kind = JVal::Obj;
      if (peek('}')) {
        ++p;
        return true;
      }
      for (;;) {
        std::string k;
        if (!string(k) || !expect(':')) return false;
        v.obj.emplace_back(std::move(k), JVal());
        if (!value(v.obj.back().second)) return false;
        ws();
        if (p < end && *p == ',') {
          ++p;
          continue;
        }
        return expect('}');
      }
    }
    if (*p == '[') {
      ++p;
      v.kind = JVal::Arr;
      if (peek(']')) {
        ++p;
        return true;
      }
      for (;;) {
        v.arr.emplace_back();
        if (!value(v.arr.back())) return false;
        ws();
        if (p < end && *p == ',') {
          ++p;
          continue;
        }
        return expect(']');
      }
    }
    if (*p == '"') {
      v.kind = JVal::Str;
      return string(v.str);
    }
    if (!strncmp(p, "true", 4) && end - p >= 4) {
      v.kind = JVal::Bool;
      v.num = 1;
      p += 4;
      return true;
    }
    if (!strncmp(p, "false", 5) && end - p >= 5) {
      v.kind = JVal::Bool;
      p += 5;
      return true;
    }
    if (!strncmp(p, "null", 4) && end - p >= 4) {
      p += 4;
      return true;
    }
    char *e = nullptr;
    errno = 0;
    const double d = strtod(p, &e);
    if (e == p) return fail("value expected");
    v.kind = JVal::Num;
    v.num = d;
    v.is_int = true;
    for (const char *c = p; c < e; ++c)
      if (*c == '.' || *c == 'e' || *c == 'E') v.is_int = false;
    if (v.is_int) v.inum = strtoll(p, nullptr, 10);
    p = e;
    return true;
  }
};

bool read_file(const std::string &path, std::string &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
  fclose(f);
  return got == out.size();
}

std::string join(const std::string &dir, const std::string &name) {
  if (dir.empty() || dir.back() == '/') return dir + name;
  return dir + "/" + name;
}

std::string basename_part(const std::string &filename) {  // stlplus::basename_part: no directory, no extension
  size_t s = filename.find_last_of("/\\");
  std::string b = s == std::string::npos ? filename : filename.substr(s + 1);
  size_t d = b.find_last_of('.');
  return d == std::string::npos ? b : b.substr(0, d);
}

long long jint(const JVal *v, long long dflt = -1) {
  if (!v || v->kind != JVal::Num) return dflt;
  return v->is_int ? v->inum : (long long)v->num;
}
double jnum(const JVal *v, double dflt = 0.0) { return (v && v->kind == JVal::Num) ? v->num : dflt; }

struct ViewRec {
  uint32_t id;
  std::string filename;
  uint32_t w, h, id_intrinsic, id_pose;
};

}  // namespace

struct Scene {
  std::vector<uint32_t> view_id, view_off, view_wh;
  std::vector<std::string> view_file;
  std::vector<uint8_t> desc;
  std::vector<float> kpt;
  std::vector<int32_t> row_landmark;
  std::vector<uint32_t> landmark_id;
  std::vector<double> landmark_X;
  std::vector<double> view_center;  // [n_views*3] camera centres (getLocalViews, SfMDataUtils.cpp:210-227)
  double focal = 0, ppx = 0, ppy = 0, k1 = 0, k2 = 0, k3 = 0;
  uint32_t intrinsic_type = 0;  // 0 pinhole, 3 pinhole_radial_k3
  uint32_t bow_dim = 0;
  std::vector<float> bow;
  uint32_t n_views_total = 0, n_landmarks_total = 0, n_obs_used = 0;
};

// returns SFMLOC_OK or SFMLOC_EIO with the message set
int load_scene(const char *sfm_dir, const char *match_dir, Scene &S) {
  const std::string sfm_path = join(sfm_dir, "sfm_data.json");
  std::string text;
  SFM_CHECK(read_file(sfm_path, text), SFMLOC_EIO, "The input sfm_data.json file \"%s\" cannot be read.",
            sfm_path.c_str());
  JParser P(text);
  std::vector<ViewRec> views;
  std::map<uint32_t, std::vector<double>> pose_center;  // id_pose -> centre
  bool have_intrinsic0 = false;
  struct Lm {
    uint32_t id;
    double X[3];
    std::vector<std::pair<uint32_t, uint32_t>> obs;  // (view id, id_feat)
  };
  std::vector<Lm> lms;

  if (!P.expect('{')) goto bad;
  if (!P.peek('}')) {
    for (;;) {
      std::string key;
      if (!P.string(key) || !P.expect(':')) goto bad;
      const bool streamed = key == "views" || key == "intrinsics" || key == "extrinsics" || key == "structure";
      if (streamed && P.peek('[')) {
        P.expect('[');
        if (!P.peek(']')) {
          for (;;) {
            JVal e;
            if (!P.value(e)) goto bad;
            const JVal *k = e.get("key");
            const JVal *val = e.get("value");
            if (!val) val = e.get("values");  // OpenMVG < 0.3 (reconstructGraph.py:130-134)
            if (key == "views" && val) {
              const JVal *pw = val->get("ptr_wrapper");
              const JVal *d = pw ? pw->get("data") : nullptr;
              if (d) {
                ViewRec r;
                r.id = (uint32_t)jint(d->get("id_view"), jint(k, 0));
                const JVal *fn = d->get("filename");
                r.filename = fn ? fn->str : "";
                r.w = (uint32_t)jint(d->get("width"), 0);
                r.h = (uint32_t)jint(d->get("height"), 0);
                r.id_intrinsic = (uint32_t)jint(d->get("id_intrinsic"), 0);
                r.id_pose = (uint32_t)jint(d->get("id_pose"), r.id);
                views.push_back(r);
              }
            } else if (key == "intrinsics" && val && jint(k, -1) == 0) {
              const JVal *pw = val->get("ptr_wrapper");
              const JVal *d = pw ? pw->get("data") : nullptr;
              if (d) {
                const JVal *base = d->get("value0") ? d->get("value0") : d;  // pinhole_radial_k3 nests its base class
                S.focal = jnum(base->get("focal_length"));
                const JVal *pp = base->get("principal_point");
                if (pp && pp->arr.size() == 2) {
                  S.ppx = pp->arr[0].num;
                  S.ppy = pp->arr[1].num;
                }
                // other OpenMVG camera models (radial_k1, brown_t2, fisheye) would need their own get_ud_pixel: refuse
                // them rather than treat them as an undistorted pinhole
                for (const char *other : {"disto_k1", "disto_t2", "fisheye"})
                  if (d->get(other)) {
                    set_error("%s: intrinsic 0 carries \"%s\": only pinhole and pinhole_radial_k3 cameras are supported",
                              sfm_path.c_str(), other);
                    return SFMLOC_EIO;
                  }
                const JVal *dk = d->get("disto_k3");
                if (dk && dk->arr.size() == 3) {
                  S.intrinsic_type = 3;
                  S.k1 = dk->arr[0].num;
                  S.k2 = dk->arr[1].num;
                  S.k3 = dk->arr[2].num;
                }
                have_intrinsic0 = S.focal > 0;
              }
            } else if (key == "extrinsics" && val) {
              const JVal *c = val->get("center");
              if (c && c->arr.size() == 3)
                pose_center[(uint32_t)jint(k, 0)] = {c->arr[0].num, c->arr[1].num, c->arr[2].num};
            } else if (key == "structure" && val) {
              Lm L;
              L.id = (uint32_t)jint(k, 0);
              const JVal *X = val->get("X");
              if (X && X->arr.size() == 3) {
                L.X[0] = X->arr[0].num;
                L.X[1] = X->arr[1].num;
                L.X[2] = X->arr[2].num;
                const JVal *obs = val->get("observations");
                if (obs)
                  for (auto &o : obs->arr) {
                    const JVal *ov = o.get("value");
                    if (ov) L.obs.emplace_back((uint32_t)jint(o.get("key"), 0), (uint32_t)jint(ov->get("id_feat"), 0));
                  }
                lms.push_back(std::move(L));
              }
            }
            P.ws();
            if (P.p < P.end && *P.p == ',') {
              ++P.p;
              continue;
            }
            if (!P.expect(']')) goto bad;
            break;
          }
        } else {
          P.expect(']');
        }
      } else {
        JVal skip;
        if (!P.value(skip)) goto bad;
      }
      P.ws();
      if (P.p < P.end && *P.p == ',') {
        ++P.p;
        continue;
      }
      if (!P.expect('}')) goto bad;
      break;
    }
  }
  {
    SFM_CHECK(!views.empty(), SFMLOC_EIO, "%s: no views", sfm_path.c_str());
    SFM_CHECK(have_intrinsic0, SFMLOC_EIO, "%s: intrinsic id 0 missing (localization.cpp:484-487 uses it)",
              sfm_path.c_str());
    S.n_views_total = (uint32_t)views.size();
    S.n_landmarks_total = (uint32_t)lms.size();
    std::sort(views.begin(), views.end(), [](const ViewRec &a, const ViewRec &b) { return a.id < b.id; });
    // posed views only (localization.cpp:337-341), ascending id (std::map order)
    std::map<uint32_t, uint32_t> slot_of_view;
    S.view_off.push_back(0);
    for (const ViewRec &v : views) {
      auto pc = pose_center.find(v.id_pose);
      if (pc == pose_center.end()) continue;
      const std::string base = basename_part(v.filename);
      const std::string fdesc = join(match_dir, base + ".desc");
      const std::string ffeat = join(match_dir, base + ".feat");
      // .desc : u64 count + count x 64 bytes
      FILE *f = fopen(fdesc.c_str(), "rb");
      SFM_CHECK(f, SFMLOC_EIO, "cannot open %s", fdesc.c_str());
      uint64_t n = 0;
      bool ok = fread(&n, sizeof(n), 1, f) == 1;
      const size_t at = S.desc.size();
      if (ok) {  // the count comes from the file: it must fit what the file holds before anything is sized by it
        struct stat fs;
        ok = fstat(fileno(f), &fs) == 0 && fs.st_size >= 8 && n <= ((uint64_t)fs.st_size - 8) / 64;
      }
      if (ok) {
        S.desc.resize(at + (size_t)n * 64);
        ok = n == 0 || fread(&S.desc[at], 64, (size_t)n, f) == (size_t)n;
      }
      fclose(f);
      SFM_CHECK(ok, SFMLOC_EIO, "%s: truncated (expected %llu descriptors)", fdesc.c_str(), (unsigned long long)n);
      // .feat : "x y size angle" per line
      std::string ft;
      SFM_CHECK(read_file(ffeat, ft), SFMLOC_EIO, "cannot open %s", ffeat.c_str());
      uint64_t nf = 0;
      const char *c = ft.c_str();
      for (;;) {
        char *e1, *e2, *e3, *e4;
        const float x = strtof(c, &e1);
        if (e1 == c) break;
        const float y = strtof(e1, &e2);
        if (e2 == e1) break;
        strtof(e2, &e3);
        if (e3 == e2) break;
        strtof(e3, &e4);
        if (e4 == e3) break;
        c = e4;
        S.kpt.push_back(x);
        S.kpt.push_back(y);
        ++nf;
      }
      SFM_CHECK(nf == n, SFMLOC_EIO, "%s has %llu keypoints but %s has %llu descriptors", ffeat.c_str(),
                (unsigned long long)nf, fdesc.c_str(), (unsigned long long)n);
      slot_of_view[v.id] = (uint32_t)S.view_id.size();
      S.view_id.push_back(v.id);
      S.view_file.push_back(v.filename);
      S.view_wh.push_back(v.w);
      S.view_wh.push_back(v.h);
      S.view_center.insert(S.view_center.end(), pc->second.begin(), pc->second.end());
      S.view_off.push_back((uint32_t)(S.desc.size() / 64));
      SFM_CHECK(S.desc.size() / 64 < (1ull << 32) - 64, SFMLOC_EIO, "more than 2^32 descriptors");
    }
    SFM_CHECK(!S.view_id.empty(), SFMLOC_EIO, "%s: no view has a pose", sfm_path.c_str());
    // (view, feat) -> landmark (structureToMapViewFeatTo3D; later landmarks overwrite earlier ones like map[][]=)
    S.row_landmark.assign(S.desc.size() / 64, -1);
    std::sort(lms.begin(), lms.end(), [](const Lm &a, const Lm &b) { return a.id < b.id; });
    for (const Lm &L : lms) {
      const int32_t slot = (int32_t)S.landmark_id.size();
      S.landmark_id.push_back(L.id);
      S.landmark_X.push_back(L.X[0]);
      S.landmark_X.push_back(L.X[1]);
      S.landmark_X.push_back(L.X[2]);
      for (auto &o : L.obs) {
        auto it = slot_of_view.find(o.first);
        if (it == slot_of_view.end()) continue;
        const uint32_t v = it->second;
        const uint32_t nv = S.view_off[v + 1] - S.view_off[v];
        SFM_CHECK(o.second < nv, SFMLOC_EIO, "landmark %u observes feature %u of view %u, which has %u features",
                  L.id, o.second, o.first, nv);
        S.row_landmark[S.view_off[v] + o.second] = slot;
        ++S.n_obs_used;
      }
    }
    // optional <base>.bow per view: int32 rows, cols, cvType + data; TrainBoW writes 500 x 1 CV_64F
    bool all_bow = true;
    std::vector<float> bow;
    uint32_t dim = 0;
    for (size_t v = 0; v < S.view_id.size() && all_bow; ++v) {
      const std::string fb = join(match_dir, basename_part(S.view_file[v]) + ".bow");
      FILE *f = fopen(fb.c_str(), "rb");
      if (!f) {
        all_bow = false;
        break;
      }
      int32_t hdr[3] = {0, 0, 0};
      bool ok = fread(hdr, sizeof(int32_t), 3, f) == 3 && hdr[0] > 0 && hdr[1] > 0;
      const uint32_t n = ok ? (uint32_t)(hdr[0] * hdr[1]) : 0;
      if (ok && dim == 0) dim = n;
      ok = ok && n == dim && (hdr[2] == 6 || hdr[2] == 5);
      if (ok) {
        if (hdr[2] == 6) {
          std::vector<double> t(n);
          ok = fread(t.data(), sizeof(double), n, f) == n;
          for (double x : t) bow.push_back((float)x);  // BoFUtils.cpp:43-45 converts to CV_32F
        } else {
          std::vector<float> t(n);
          ok = fread(t.data(), sizeof(float), n, f) == n;
          bow.insert(bow.end(), t.begin(), t.end());
        }
      }
      fclose(f);
      if (!ok) {
        set_error("%s: not a %u-element CV_64F/CV_32F matrix", fb.c_str(), dim);
        return SFMLOC_EIO;
      }
    }
    if (all_bow && dim) {
      S.bow_dim = dim;
      S.bow.swap(bow);
    }
    return SFMLOC_OK;
  }
bad:
  set_error("%s: JSON error near byte %lld: %s", sfm_path.c_str(), (long long)(P.p - text.data()),
            P.err.empty() ? "syntax" : P.err.c_str());
  return SFMLOC_EIO;
}

}  // namespace sfmloc

using namespace sfmloc;

// ---- packed map file: everything load_scene produces, as one little-endian binary.  Opening a
//      10 000-view map from the reference's files means parsing a large JSON and 20 000 small files (the .feat files
//      are text); the packed file is read at disk speed.  Layout: "SFMLOCM1", then the scalars, then each array as
//      u64 count + raw elements, in the order of write_packed below. ----
namespace {
const char kPackMagic[8] = {'S', 'F', 'M', 'L', 'O', 'C', 'M', '1'};

template <typename T>
bool put_vec(FILE *f, const std::vector<T> &v) {
  const uint64_t n = v.size();
  return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <typename T>
bool get_vec(FILE *f, std::vector<T> &v, uint64_t max_elems) {
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1 || n > max_elems) return false;
  // the count comes from the file: it must fit the bytes that are left before anything is sized by it
  struct stat fs;
  const long pos = ftell(f);
  if (pos < 0 || fstat(fileno(f), &fs) != 0 || (uint64_t)fs.st_size < (uint64_t)pos ||
      n > ((uint64_t)fs.st_size - (uint64_t)pos) / sizeof(T))
    return false;
  v.resize((size_t)n);
  return n == 0 || fread(v.data(), sizeof(T), (size_t)n, f) == n;
}

bool write_packed(const char *path, const Scene &S) {
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  const double sc[6] = {S.focal, S.ppx, S.ppy, S.k1, S.k2, S.k3};
  const uint32_t u[5] = {S.intrinsic_type, S.bow_dim, S.n_views_total, S.n_landmarks_total, S.n_obs_used};
  bool ok = fwrite(kPackMagic, 8, 1, f) == 1 && fwrite(sc, sizeof(sc), 1, f) == 1 && fwrite(u, sizeof(u), 1, f) == 1;
  ok = ok && put_vec(f, S.view_id) && put_vec(f, S.view_off) && put_vec(f, S.view_wh) && put_vec(f, S.desc) &&
       put_vec(f, S.kpt) && put_vec(f, S.row_landmark) && put_vec(f, S.landmark_id) && put_vec(f, S.landmark_X) &&
       put_vec(f, S.view_center) && put_vec(f, S.bow);
  uint64_t nf = S.view_file.size();
  ok = ok && fwrite(&nf, 8, 1, f) == 1;
  for (const std::string &name : S.view_file) {
    const uint64_t len = name.size();
    ok = ok && fwrite(&len, 8, 1, f) == 1 && (len == 0 || fwrite(name.data(), 1, len, f) == len);
  }
  return (fclose(f) == 0) && ok;
}

int read_packed(const char *path, Scene &S) {
  FILE *f = fopen(path, "rb");
  SFM_CHECK(f != nullptr, SFMLOC_EIO, "packed map \"%s\" cannot be read", path);
  char magic[8];
  double sc[6];
  uint32_t u[5];
  const uint64_t kMax = 1ull << 36;
  bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, kPackMagic, 8) == 0 && fread(sc, sizeof(sc), 1, f) == 1 &&
            fread(u, sizeof(u), 1, f) == 1;
  ok = ok && get_vec(f, S.view_id, kMax) && get_vec(f, S.view_off, kMax) && get_vec(f, S.view_wh, kMax) &&
       get_vec(f, S.desc, kMax) && get_vec(f, S.kpt, kMax) && get_vec(f, S.row_landmark, kMax) &&
       get_vec(f, S.landmark_id, kMax) && get_vec(f, S.landmark_X, kMax) && get_vec(f, S.view_center, kMax) &&
       get_vec(f, S.bow, kMax);
  uint64_t nf = 0;
  ok = ok && fread(&nf, 8, 1, f) == 1 && nf <= (1ull << 28);
  for (uint64_t i = 0; ok && i < nf; ++i) {
    uint64_t len = 0;
    ok = fread(&len, 8, 1, f) == 1 && len <= 65536;
    if (!ok) break;
    std::string name((size_t)len, '\0');
    ok = len == 0 || fread(&name[0], 1, (size_t)len, f) == len;
    S.view_file.push_back(std::move(name));
  }
  fclose(f);
  // the invariants sfmloc_map_create relies on
  const size_t nv = S.view_id.size();
  ok = ok && S.view_off.size() == nv + 1 && S.view_wh.size() == 2 * nv && S.desc.size() % 64 == 0 &&
       (nv == 0 || S.view_off[nv] == S.desc.size() / 64) && S.kpt.size() == 2 * (S.desc.size() / 64) &&
       S.row_landmark.size() == S.desc.size() / 64 && S.landmark_X.size() == 3 * S.landmark_id.size() &&
       (S.view_center.empty() || S.view_center.size() == 3 * nv) && (S.bow.empty() || S.bow.size() == (size_t)u[1] * nv);
  SFM_CHECK(ok, SFMLOC_EIO, "packed map \"%s\" is truncated, corrupt or of another version", path);
  S.focal = sc[0], S.ppx = sc[1], S.ppy = sc[2], S.k1 = sc[3], S.k2 = sc[4], S.k3 = sc[5];
  S.intrinsic_type = u[0], S.bow_dim = u[1], S.n_views_total = u[2], S.n_landmarks_total = u[3], S.n_obs_used = u[4];
  return SFMLOC_OK;
}

void scene_info(const Scene &S, sfmloc_scan_info *info) {
  memset(info, 0, sizeof(*info));
  info->n_views_total = S.n_views_total;
  info->n_views_posed = (uint32_t)S.view_id.size();
  info->n_rows = S.desc.size() / 64;
  info->n_landmarks = (uint32_t)S.landmark_id.size();
  info->n_observations = S.n_obs_used;
  info->bow_dim = S.bow_dim;
  info->focal = S.focal;
  info->ppx = S.ppx;
  info->ppy = S.ppy;
  info->k1 = S.k1;
  info->k2 = S.k2;
  info->k3 = S.k3;
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the descriptor bytes: lets a CPU test pin the payload
  for (uint8_t b : S.desc) {
    h ^= b;
    h *= 1099511628211ull;
  }
  info->desc_fnv1a = h;
  double ks = 0;
  for (float x : S.kpt) ks += (double)x;
  info->kpt_sum = ks;
  long long ls = 0;
  for (int32_t x : S.row_landmark) ls += x;
  info->row_landmark_sum = ls;
}

int scene_to_map(const Scene &S, const sfmloc_params *params, sfmloc_map **out) {
  sfmloc_map_desc d;
  memset(&d, 0, sizeof(d));
  d.n_views = (uint32_t)S.view_id.size();
  d.view_id = S.view_id.data();
  d.view_off = S.view_off.data();
  d.view_wh = S.view_wh.data();
  d.n_rows = S.desc.size() / 64;
  d.desc = S.desc.data();
  d.kpt_xy = S.kpt.data();
  d.row_landmark = S.row_landmark.data();
  d.n_landmarks = (uint32_t)S.landmark_id.size();
  d.landmark_id = S.landmark_id.data();
  d.landmark_X = S.landmark_X.data();
  d.focal = S.focal;
  d.ppx = S.ppx;
  d.ppy = S.ppy;
  d.k1 = S.k1;
  d.k2 = S.k2;
  d.k3 = S.k3;
  d.intrinsic_type = S.intrinsic_type;
  d.bow_dim = S.bow_dim;
  d.bow = S.bow.empty() ? nullptr : S.bow.data();
  const int rc = sfmloc_map_create(&d, params, out);
  if (rc) return rc;
  Map *m = reinterpret_cast<Map *>(*out);
  m->h_view_center = S.view_center;
  m->h_view_file = S.view_file;
  return SFMLOC_OK;
}
}  // namespace

extern "C" {

int sfmloc_scan(const char *sfm_dir, const char *match_dir, sfmloc_scan_info *info) {
  try {
    SFM_CHECK(sfm_dir && match_dir && info, SFMLOC_EINVAL, "sfmloc_scan: null argument");
    Scene S;
    int rc = load_scene(sfm_dir, match_dir, S);
    if (rc) return rc;
    scene_info(S, info);
    return SFMLOC_OK;
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_scan: out of host memory");
    return SFMLOC_ENOMEM;
  } catch (const std::exception &e) {
    set_error("sfmloc_scan: %s", e.what());
    return SFMLOC_EIO;
  }
}

int sfmloc_open(const char *sfm_dir, const char *match_dir, const sfmloc_params *params, sfmloc_map **out) {
  try {
    SFM_CHECK(sfm_dir && match_dir && out, SFMLOC_EINVAL, "sfmloc_open: null argument");
    *out = nullptr;
    Scene S;
    int rc = load_scene(sfm_dir, match_dir, S);
    if (rc) return rc;
    return scene_to_map(S, params, out);
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_open: out of host memory");
    return SFMLOC_ENOMEM;
  } catch (const std::exception &e) {
    set_error("sfmloc_open: %s", e.what());
    return SFMLOC_EIO;
  }
}

int sfmloc_pack(const char *sfm_dir, const char *match_dir, const char *out_path) {
  try {
    SFM_CHECK(sfm_dir && match_dir && out_path, SFMLOC_EINVAL, "sfmloc_pack: null argument");
    Scene S;
    int rc = load_scene(sfm_dir, match_dir, S);
    if (rc) return rc;
    SFM_CHECK(write_packed(out_path, S), SFMLOC_EIO, "sfmloc_pack: cannot write \"%s\"", out_path);
    return SFMLOC_OK;
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_pack: out of host memory");
    return SFMLOC_ENOMEM;
  } catch (const std::exception &e) {
    set_error("sfmloc_pack: %s", e.what());
    return SFMLOC_EIO;
  }
}

int sfmloc_scan_packed(const char *path, sfmloc_scan_info *info) {
  try {
    SFM_CHECK(path && info, SFMLOC_EINVAL, "sfmloc_scan_packed: null argument");
    Scene S;
    int rc = read_packed(path, S);
    if (rc) return rc;
    scene_info(S, info);
    return SFMLOC_OK;
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_scan_packed: out of host memory");
    return SFMLOC_ENOMEM;
  } catch (const std::exception &e) {
    set_error("sfmloc_scan_packed: %s", e.what());
    return SFMLOC_EIO;
  }
}

int sfmloc_open_packed(const char *path, const sfmloc_params *params, sfmloc_map **out) {
  try {
    SFM_CHECK(path && out, SFMLOC_EINVAL, "sfmloc_open_packed: null argument");
    *out = nullptr;
    Scene S;
    int rc = read_packed(path, S);
    if (rc) return rc;
    return scene_to_map(S, params, out);
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_open_packed: out of host memory");
    return SFMLOC_ENOMEM;
  } catch (const std::exception &e) {
    set_error("sfmloc_open_packed: %s", e.what());
    return SFMLOC_EIO;
  }
}

int sfmloc_map_views(const sfmloc_map *map, uint32_t *view_id, uint32_t *view_off, double *center) {
  SFM_CHECK(map, SFMLOC_EINVAL, "sfmloc_map_views: null map");
  const Map *m = reinterpret_cast<const Map *>(map);
  if (view_id) memcpy(view_id, m->h_view_id.data(), m->h_view_id.size() * sizeof(uint32_t));
  if (view_off) memcpy(view_off, m->h_view_off.data(), m->h_view_off.size() * sizeof(uint32_t));
  if (center) {
    SFM_CHECK(m->h_view_center.size() == 3 * (size_t)m->n_views, SFMLOC_EINVAL,
              "sfmloc_map_views: this map has no camera centres (not opened from sfm_data.json)");
    memcpy(center, m->h_view_center.data(), m->h_view_center.size() * sizeof(double));
  }
  return SFMLOC_OK;
}

int sfmloc_map_view_sizes(const sfmloc_map *map, uint32_t *wh) {
  SFM_CHECK(map && wh, SFMLOC_EINVAL, "sfmloc_map_view_sizes: null argument");
  const Map *m = reinterpret_cast<const Map *>(map);
  SFM_CHECK(m->h_view_wh.size() == 2 * (size_t)m->n_views, SFMLOC_EINVAL,
            "sfmloc_map_view_sizes: the map was created without view sizes");
  memcpy(wh, m->h_view_wh.data(), m->h_view_wh.size() * sizeof(uint32_t));
  return SFMLOC_OK;
}

// ---- the view table of an sfm_data.json alone (ExtFeatAndMatch works on <matchDir>/sfm_data.json before any
//      reconstruction exists: computeFeaturesAndMatches.cpp:118-126) ----
struct ViewList {
  std::vector<uint32_t> id, w, h;
  std::vector<std::string> image;  // root_path / filename
};

int sfmloc_view_list_open(const char *sfm_data_json, sfmloc_view_list **out, uint32_t *n_views) {
  try {
    SFM_CHECK(sfm_data_json && out, SFMLOC_EINVAL, "sfmloc_view_list_open: null argument");
    std::string text;
    SFM_CHECK(read_file(sfm_data_json, text), SFMLOC_EIO, "The input sfm_data.json file \"%s\" cannot be read.",
              sfm_data_json);
    JParser P(text);
    JVal root;
    SFM_CHECK(P.value(root) && root.kind == JVal::Obj, SFMLOC_EIO, "%s: not a JSON object (%s)", sfm_data_json,
              P.err.c_str());
    const JVal *rp = root.get("root_path");
    const std::string root_path = (rp && rp->kind == JVal::Str) ? rp->str : "";
    const JVal *views = root.get("views");
    SFM_CHECK(views && views->kind == JVal::Arr, SFMLOC_EIO, "%s: no \"views\" array", sfm_data_json);
    std::vector<ViewRec> recs;
    for (const JVal &e : views->arr) {
      const JVal *val = e.get("value");
      if (!val) val = e.get("values");
      const JVal *pw = val ? val->get("ptr_wrapper") : nullptr;
      const JVal *d = pw ? pw->get("data") : nullptr;
      if (!d) continue;
      ViewRec r;
      r.id = (uint32_t)jint(d->get("id_view"), jint(e.get("key"), 0));
      const JVal *fn = d->get("filename");
      r.filename = fn ? fn->str : "";
      r.w = (uint32_t)jint(d->get("width"), 0);
      r.h = (uint32_t)jint(d->get("height"), 0);
      r.id_intrinsic = r.id_pose = 0;
      recs.push_back(r);
    }
    std::stable_sort(recs.begin(), recs.end(), [](const ViewRec &a, const ViewRec &b) { return a.id < b.id; });  // Views is a std::map
    ViewList *L = new ViewList;
    for (const ViewRec &r : recs) {
      L->id.push_back(r.id);
      L->w.push_back(r.w);
      L->h.push_back(r.h);
      const size_t s = r.filename.find_last_of("/\\");
      L->image.push_back(join(root_path, s == std::string::npos ? r.filename : r.filename.substr(s + 1)));
    }
    if (n_views) *n_views = (uint32_t)L->id.size();
    *out = reinterpret_cast<sfmloc_view_list *>(L);
    return SFMLOC_OK;
  } catch (const std::bad_alloc &) {
    set_error("sfmloc_view_list_open: out of host memory");
    return SFMLOC_ENOMEM;
  } catch (const std::exception &e) {
    set_error("sfmloc_view_list_open: %s", e.what());
    return SFMLOC_EIO;
  }
}

int sfmloc_view_list_get(const sfmloc_view_list *list, uint32_t k, uint32_t *view_id, uint32_t *width, uint32_t *height,
                         const char **image_path) {
  SFM_CHECK(list, SFMLOC_EINVAL, "sfmloc_view_list_get: null list");
  const ViewList *L = reinterpret_cast<const ViewList *>(list);
  SFM_CHECK(k < L->id.size(), SFMLOC_EINVAL, "sfmloc_view_list_get: view %u of %zu", k, L->id.size());
  if (view_id) *view_id = L->id[k];
  if (width) *width = L->w[k];
  if (height) *height = L->h[k];
  if (image_path) *image_path = L->image[k].c_str();
  return SFMLOC_OK;
}

void sfmloc_view_list_close(sfmloc_view_list *list) { delete reinterpret_cast<ViewList *>(list); }

}  // extern "C"
