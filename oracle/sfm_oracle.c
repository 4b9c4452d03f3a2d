/*
 * sfm_oracle.c -- CPU restatement of the query-localisation hot path of hulop/SfMLocalization.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (sfmlocalization_amd/) never does and fails
 * loudly when its HIP library is missing.
 *
 * PARITY UNPINNED.  The reference ships no golden vectors, tests or fixtures for this path
 *  and its arithmetic lives in OpenCV 3.0 / OpenMVG 1.1, which are absent from
 * /root/reference and from this image.  Each function below follows the reference's call
 * site (cited) and, where the callee is third-party, the published algorithm.  Where the reference is
 * approximate or non-deterministic (FLANN-LSH search, unseeded RNG) the oracle is the exact /
 * deterministic limit it approximates, and says so.
 *
 * Plain C11, no dependencies beyond libm/OpenMP.  Build: oracle/Makefile -> oracle/_build/liboracle.so
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NOMATCH 0xFFFFFFFFu

int orc_version(void) { return 1; }

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Hamming distance over all 64 stored bytes of two .desc rows.  Bytes 61..63 are zero padding
 * (FileUtils.cpp:77-92) so this equals the 486-bit M-LDB distance that OpenCV's
 * FLANN_DIST_HAMMING computes on the 64-byte rows readAKAZEBin returns (FileUtils.cpp:94-103). */
static inline int hamming64(const uint8_t *a, const uint8_t *b) {
  uint64_t x[8], y[8];
  memcpy(x, a, 64);
  memcpy(y, b, 64);
  int d = 0;
  for (int k = 0; k < 8; ++k) d += __builtin_popcountll(x[k] ^ y[k]);
  return d;
}

/*
 * Exact 2-NN of every bank row among the query rows: what
 *   matchers[..]->knnSearch(desc1, matchesMat, distMat, 2, ...)        MatchUtils.cpp:339-340
 * approximates with an LSH index built on the QUERY descriptors (MatchUtils.cpp:303-311; search set =
 * the map image's descriptors, MatchUtils.cpp:331-332).  Rules fixed here because the reference leaves
 * them to an approximate search: ties go to the lowest query index; with fewer than two
 * query rows there is no second neighbour (j1 = -1, d1 = INT_MAX, as an empty FLANN result slot).
 */
static void hamming_2nn_rows(const uint8_t *query, uint32_t nq, const uint8_t *bank, int64_t r_begin, int64_t r_end,
                             int32_t *j0, int32_t *d0, int32_t *j1, int32_t *d1) {
  for (int64_t r = r_begin; r < r_end; ++r) {
    const uint8_t *row = bank + (uint64_t)r * 64;
    int bj0 = -1, bd0 = INT_MAX, bj1 = -1, bd1 = INT_MAX;
    for (uint32_t j = 0; j < nq; ++j) {
      const int d = hamming64(row, query + (uint64_t)j * 64);
      if (d < bd0) {
        bd1 = bd0;
        bj1 = bj0;
        bd0 = d;
        bj0 = (int)j;
      } else if (d < bd1) {
        bd1 = d;
        bj1 = (int)j;
      }
    }
    j0[r] = bj0;
    d0[r] = bd0;
    j1[r] = bj1;
    d1[r] = bd1;
  }
}

void orc_hamming_2nn(const uint8_t *query, uint32_t nq, const uint8_t *bank, uint64_t n_rows, int32_t *j0,
                     int32_t *d0, int32_t *j1, int32_t *d1, int threads) {
  (void)threads;
  const int64_t chunk = 256;
  const int64_t n_chunks = ((int64_t)n_rows + chunk - 1) / chunk;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
  for (int64_t c = 0; c < n_chunks; ++c) {
    const int64_t b = c * chunk;
    const int64_t e = b + chunk < (int64_t)n_rows ? b + chunk : (int64_t)n_rows;
    hamming_2nn_rows(query, nq, bank, b, e, j0, d0, j1, d1);
  }
}

/* The ratio test exactly as written at MatchUtils.cpp:347-349:
 *   if ((0.0f + distMat.at<int>(i,0)) / distMat.at<int>(i,1) < fDistRatio)
 *     if (distMat.at<int>(i,1) < numeric_limits<int>::max())
 * float32 arithmetic; d1 == 0 gives 0/0 = NaN -> false. */
int orc_ratio_accept(int d0, int d1, float ratio) {
  volatile float num = 0.0f + (float)d0;
  volatile float den = (float)d1;
  volatile float q = num / den;
  return (q < ratio) && (d1 < INT_MAX);
}

/*
 * matchAKAZEToQuery (MatchUtils.cpp:283-367) followed by nothing else: for every selected view, in
 * ascending map-feature index i, emit IndMatch(i, j0) when the ratio test passes.  Output layout mirrors
 * the product's (include/sfmloc.h sfmloc_putative_read): view v's list starts at view_off[v].
 *   view_sel == NULL  -> all views (localization.cpp:386-392 with localViews = every posed view)
 * view_count is written for every view (0 when not selected).  The ">= 16" filter
 * (localization.cpp:408-415) is left to the caller so that both sides of it can be compared.
 */
void orc_match_to_query(const uint8_t *query, uint32_t nq, const uint8_t *bank, const uint32_t *view_off,
                        uint32_t n_views, const uint32_t *view_sel, uint32_t n_sel, float ratio,
                        uint32_t *view_count, uint32_t *match_i, uint32_t *match_j, uint32_t *match_d,
                        int threads) {
  for (uint32_t v = 0; v < n_views; ++v) view_count[v] = 0;
  if (nq < 1) return; /* MatchUtils.cpp:299-301 */
  const uint32_t n_iter = view_sel ? n_sel : n_views;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
  for (int64_t k = 0; k < (int64_t)n_iter; ++k) {
    const uint32_t v = view_sel ? view_sel[k] : (uint32_t)k;
    const uint32_t off = view_off[v], end = view_off[v + 1];
    const uint32_t n = end - off;
    if (n == 0) continue;
    int32_t *tmp = (int32_t *)malloc((size_t)n * 4 * sizeof(int32_t));
    int32_t *j0 = tmp, *d0 = tmp + n, *j1 = tmp + 2 * (size_t)n, *d1 = tmp + 3 * (size_t)n;
    hamming_2nn_rows(query, nq, bank + (uint64_t)off * 64, 0, n, j0, d0, j1, d1);
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; ++i) {
      if (orc_ratio_accept(d0[i], d1[i], ratio)) {
        match_i[off + c] = i;
        match_j[off + c] = (uint32_t)j0[i];
        match_d[off + c] = (uint32_t)d0[i];
        ++c;
      }
    }
    view_count[v] = c;
    free(tmp);
  }
}
