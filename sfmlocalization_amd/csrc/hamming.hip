// K1 / K2 of the query-localisation path, written for gfx950 (MI355X, wave64).
//
//   K1  k_hamming_top2          exact 2-NN (Hamming over 64 stored bytes) of every bank row among the
//                               query's descriptors                      [reference: MatchUtils.cpp:331-340,
//                               where an approximate LSH index stands in for this search]
//   K2  k_merge_ratio_compact   Lowe ratio in the reference's float32 expression + per-view ordered
//                               compaction + per-view count              [MatchUtils.cpp:343-355,
//                               localization.cpp:408-415]
//
// Search direction is map -> query: one LANE owns one bank row (16 dwords in VGPRs), the
// query block sits in LDS and every lane of a wave reads the same query row (LDS broadcast).  Per
// (bank row, query row) pair the VALU cost is 16 v_xor_b32 + 16 v_bcnt_u32_b32 (accumulating form) +
// 1 v_lshl_or_b32 (pack (distance<<16)|query index) + v_med3_u32 + v_min_u32 (running top-2 on the
// packed key; lower query index wins ties because it makes the key smaller).  The kernel is VALU bound
// for N_q >~ 10; the bank is read exactly once per launch in fully coalesced 1 KiB pieces
// thanks to the tiled64 layout (sfmloc_internal.h).
#include <stdlib.h>

#include "chain_device.h"
#include "sfmloc_internal.h"

namespace sfmloc {

namespace {

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// keep (b0 <= b1) = two smallest keys seen so far
__device__ __forceinline__ void top2_push(uint32_t &b0, uint32_t &b1, uint32_t key) {
  b1 = umed3(b0, b1, key);
  b0 = min(b0, key);
}

// row-major .desc rows -> tiled64 (see sfmloc_internal.h). One thread per uint4.
__global__ __launch_bounds__(256) void k_tile_bank(const uint4 *__restrict__ rows, uint64_t row0,
                                                   uint64_t n_rows_chunk, uint4 *__restrict__ bank) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index of uint4 inside the chunk
  if (t >= n_rows_chunk * 4) return;
  uint64_t row = row0 + (t >> 2);
  uint32_t plane = (uint32_t)(t & 3);
  uint64_t block = row >> 6;
  uint32_t lane = (uint32_t)(row & 63);
  bank[(block * 4 + plane) * 64 + lane] = rows[t];
}

constexpr uint32_t kNoBlock = 0xFFFFFFFFu;  // block-list entry that names no bank block

template <int R, int WAVES>
struct HammingTop2Body {
  static constexpr int kGangThreads = WAVES * 64;
  static __device__ __forceinline__ void run(
    const uint4 *__restrict__ bank, const uint32_t *__restrict__ block_list, uint32_t n_work_blocks,
    const uint4 *__restrict__ qdesc, uint32_t nq, uint32_t q_chunk, uint32_t lds_rows,
    uint2 *__restrict__ part) {
#include "hamming_top2.body.inc"
  }
};
template <int R, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_hamming_top2(
    const uint4 *__restrict__ bank, const uint32_t *__restrict__ block_list, uint32_t n_work_blocks,
    const uint4 *__restrict__ qdesc, uint32_t nq, uint32_t q_chunk, uint32_t lds_rows,
    uint2 *__restrict__ part) {
#include "hamming_top2.body.inc"
}

// ---------------------------------------------------------------------------------------------------
// Exact screening (large N_q, one split).  A bank row can only pass the ratio test if its nearest query
// descriptor is closer than T = ratio_cnt[d1], and ratio_cnt is non-decreasing, so ANY upper bound s1 >= d1 gives
// a valid (larger) threshold.  The kernel therefore (a) scans the first kScreenHead query rows exactly to get s1,
// (b) for the remaining rows computes the distance over the first NW dwords only -- a lower bound -- and
// finishes the pair only when some lane of the wave is still below its threshold, (c) flags rows that ever see
// a distance below T.  Unflagged rows are provably rejected; flagged rows (the true matches, a fraction of a
// percent) are redone exactly by k_hamming_rows.  Results are bit-identical to k_hamming_top2 + K2; per pair
// the loop issues 2*NW + 1 VALU instructions instead of 35.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kScreenHead = 64;  // measured: 64 and 128 within 1 %, 256+ slower (profiles/r01_k1_screen_sweep.txt)

template <int WAVES, int NW, int kScreenBatch>
struct HammingScreenBody {
  static constexpr int kGangThreads = WAVES * 64;
  static __device__ __forceinline__ void run(
    const uint4 *__restrict__ bank, const uint32_t *__restrict__ block_list, uint32_t n_work_blocks,
    const uint4 *__restrict__ qdesc, uint32_t nq, uint32_t lds_rows, const uint16_t *__restrict__ ratio_cnt,
    unsigned long long *__restrict__ flagmask /*[work block]: rows handed to k_hamming_rows*/,
    uint2 *__restrict__ flagged /*{part index, bank row}*/, uint32_t *__restrict__ n_flagged,
    unsigned long long *__restrict__ counters /*[0] finished wave-pairs, [1] flagged rows*/, uint32_t head,
    uint4 *__restrict__ flagged_desc /*[slot / 64][4][64]: the flagged rows' descriptors, tiled like the bank*/,
    uint32_t flagged_desc_cap /*slots*/,
    const uint2 *__restrict__ head_part /*null, or [work block][lane]: the exact top-2 over the first `head` query rows,
                                  computed once by k_hamming_top2 -- the slices then start from it instead of each
                                  scanning a head of its own*/) {
#include "hamming_screen.body.inc"
  }
};
template <int WAVES, int NW, int kScreenBatch>
__global__ __launch_bounds__(WAVES * 64) void k_hamming_screen(
    const uint4 *__restrict__ bank, const uint32_t *__restrict__ block_list, uint32_t n_work_blocks,
    const uint4 *__restrict__ qdesc, uint32_t nq, uint32_t lds_rows, const uint16_t *__restrict__ ratio_cnt,
    unsigned long long *__restrict__ flagmask /*[work block]: rows handed to k_hamming_rows*/,
    uint2 *__restrict__ flagged /*{part index, bank row}*/, uint32_t *__restrict__ n_flagged,
    unsigned long long *__restrict__ counters /*[0] finished wave-pairs, [1] flagged rows*/, uint32_t head,
    uint4 *__restrict__ flagged_desc /*[slot / 64][4][64]: the flagged rows' descriptors, tiled like the bank*/,
    uint32_t flagged_desc_cap /*slots*/,
    const uint2 *__restrict__ head_part /*null, or [work block][lane]: the exact top-2 over the first `head` query rows,
                                  computed once by k_hamming_top2 -- the slices then start from it instead of each
                                  scanning a head of its own*/) {
#include "hamming_screen.body.inc"
}
// the scan of a view list (a shortlist; every such scan but the short one of a query alone on the GPU, which takes the
// batched-tail form <8, 10, 4>): the text of k_hamming_screen<8, 10, 1> under a name of its own, so that a profile's
// per-kernel statistics keep the full-bank scan (k_hamming_screen<8, 10, 1>, the roofline kernel) and these apart
__global__ __launch_bounds__(8 * 64) void k_hamming_screen_shortlist(
    const uint4 *__restrict__ bank, const uint32_t *__restrict__ block_list, uint32_t n_work_blocks,
    const uint4 *__restrict__ qdesc, uint32_t nq, uint32_t lds_rows, const uint16_t *__restrict__ ratio_cnt,
    unsigned long long *__restrict__ flagmask, uint2 *__restrict__ flagged, uint32_t *__restrict__ n_flagged,
    unsigned long long *__restrict__ counters, uint32_t head, uint4 *__restrict__ flagged_desc, uint32_t flagged_desc_cap,
    const uint2 *__restrict__ head_part) {
  constexpr int WAVES = 8, NW = 10, kScreenBatch = 1;
#include "hamming_screen.body.inc"
}

// Exact top-2 of the flagged rows.  64 flagged rows x all query rows is ~130 k pairs: on ONE compute unit that is
// 25 us of popcounts, on the query's critical path.  So a chunk of 64 rows is spread over SLICES workgroups (one
// query slice each, the slice in LDS, its waves interleaving the slice's rows); each writes its partial top-2 to a
// scratch slot, and the workgroup that arrives last at the chunk's counter merges the SLICES partials (the packed
// key makes the merge order-independent) and resets the counter.  No workgroup ever waits for another.
template <int WAVES, int SLICES>
struct HammingRowsBody {
  static constexpr int kGangThreads = WAVES * 64;
  static __device__ __forceinline__ void run(const uint4 *__restrict__ bank,
                                                     const uint4 *__restrict__ qdesc, uint32_t nq,
                                                     const uint2 *__restrict__ flagged,
                                                     const uint32_t *__restrict__ n_flagged,
                                                     uint2 *__restrict__ scratch /*[chunk][SLICES][64]*/,
                                                     uint32_t *__restrict__ arrivals /*[chunk], zero*/,
                                                     uint32_t chunk_cap, uint32_t lds_rows,
                                                     const uint4 *__restrict__ flagged_desc,
                                                     uint2 *__restrict__ part) {
#include "hamming_rows.body.inc"
  }
};
template <int WAVES, int SLICES>
__global__ __launch_bounds__(WAVES * 64) void k_hamming_rows(const uint4 *__restrict__ bank,
                                                     const uint4 *__restrict__ qdesc, uint32_t nq,
                                                     const uint2 *__restrict__ flagged,
                                                     const uint32_t *__restrict__ n_flagged,
                                                     uint2 *__restrict__ scratch /*[chunk][SLICES][64]*/,
                                                     uint32_t *__restrict__ arrivals /*[chunk], zero*/,
                                                     uint32_t chunk_cap, uint32_t lds_rows,
                                                     const uint4 *__restrict__ flagged_desc,
                                                     uint2 *__restrict__ part) {
#include "hamming_rows.body.inc"
}

// One wave per selected view.  Merges the per-split partial top-2, applies the ratio test through a
// 513-entry table (ratio_cnt[d1] = number of d0 values for which the reference's float expression
// (0.0f + d0) / d1 < ratio holds; the accepted d0 are exactly 0..cnt-1 because IEEE division is
// monotone), and compacts accepted rows in ascending row order.
struct MergeRatioCompactBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(
    const uint2 *__restrict__ part, const unsigned long long *__restrict__ flagmask /*or null: every slot is valid*/,
    uint32_t n_work_blocks, uint32_t split, const uint32_t *__restrict__ view_sel,
    const uint32_t *__restrict__ view_widx0, uint32_t n_sel, const uint32_t *__restrict__ view_off,
    const uint16_t *__restrict__ ratio_cnt, uint32_t *__restrict__ view_count, uint32_t *__restrict__ match_i,
    uint32_t *__restrict__ match_key) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= n_sel) return;
    const uint32_t v = view_sel ? view_sel[gw] : gw;
    const uint32_t off = view_off[v], end = view_off[v + 1];
    // bank row r lives in bank block r/64; a selected view's blocks are consecutive in the block list, so
    // work index = (r/64 - first block of the view) + work index of that first block
    const uint32_t blk0 = off >> 6;
    const uint32_t widx0 = view_sel ? view_widx0[gw] : blk0;
    uint32_t base = 0;
    // the mask words of the view's blocks, fetched 64 at a time up front instead of one dependent load per iteration
    const uint32_t n_blk = (end > off) ? (((end - 1) >> 6) - blk0 + 1) : 0;
    unsigned long long my_mask = ~0ull;
    uint32_t mask_base = 0;  // block (relative to blk0) held by lane 0
    if (flagmask && lane < n_blk) my_mask = flagmask[widx0 + lane];
    for (uint32_t r0 = off; r0 < end; r0 += 64) {
      const uint32_t r = r0 + lane;
      const bool valid = r < end;
      uint32_t b0 = SFMLOC_NOMATCH, b1 = SFMLOC_NOMATCH;
      const uint32_t rel = (r >> 6) - blk0;
      const uint32_t widx = widx0 + rel;
      if (flagmask && ((r0 + 63) >> 6) - blk0 >= mask_base + 64) {  // a view longer than ~4000 rows: slide the window
        mask_base = (r0 >> 6) - blk0;                                // to the first block this step touches
        my_mask = (mask_base + lane < n_blk) ? flagmask[widx0 + mask_base + lane] : 0ull;
      }
      // a 64-row step of an unaligned view touches two blocks: lanes read the word of their own row's block
      const unsigned long long mw = flagmask ? __shfl(my_mask, (int)((valid ? rel : mask_base) - mask_base), 64) : ~0ull;
      if (valid && ((mw >> (r & 63u)) & 1ull)) {
        for (uint32_t s = 0; s < split; ++s) {
          const uint2 p = part[((uint64_t)s * n_work_blocks + widx) * 64 + (r & 63u)];
          top2_push(b0, b1, p.x);
          top2_push(b0, b1, p.y);
        }
      }
      // b1 == NOMATCH <=> fewer than two query descriptors: the reference then has no second
      // neighbour (MatchUtils.cpp:349) -> reject.
      const bool accept = valid && b1 != SFMLOC_NOMATCH && (b0 >> 16) < (uint32_t)ratio_cnt[b1 >> 16];
      const unsigned long long mask = __ballot(accept);
      if (accept) {
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        match_i[off + pos] = r - off;
        match_key[off + pos] = b0;
      }
      base += (uint32_t)__popcll(mask);
    }
    if (lane == 0) view_count[v] = base;
  }
};
__global__ __launch_bounds__(256) void k_merge_ratio_compact(
    const uint2 *__restrict__ part, const unsigned long long *__restrict__ flagmask /*or null: every slot is valid*/,
    uint32_t n_work_blocks, uint32_t split, const uint32_t *__restrict__ view_sel,
    const uint32_t *__restrict__ view_widx0, uint32_t n_sel, const uint32_t *__restrict__ view_off,
    const uint16_t *__restrict__ ratio_cnt, uint32_t *__restrict__ view_count, uint32_t *__restrict__ match_i,
    uint32_t *__restrict__ match_key) {
  MergeRatioCompactBody::run(part, flagmask, n_work_blocks, split, view_sel, view_widx0, n_sel, view_off, ratio_cnt, view_count, match_i, match_key);
}

// K2 for the screened scan: only masked rows can be matches, and they are a fraction of a percent.  One wave per
// view, one LANE per 64-row block of the view: the lane walks the set bits of its mask word, so every partial-result
// load of the view is in flight at once (the row-by-row kernel above pays one dependent load per hit); an exclusive
// wave scan of the per-lane accept counts then gives each lane its place in the view's ascending list.
struct MergeRatioMaskedBody {
  static constexpr int kGangThreads = 256;
  static __device__ __forceinline__ void run(MergeMaskedArgs M, uint32_t n_sel) {
    const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= n_sel) return;
    merge_ratio_masked_view(M, gw, threadIdx.x & 63u);  // chain_device.h
  }
};
__global__ __launch_bounds__(256) void k_merge_ratio_masked(MergeMaskedArgs M, uint32_t n_sel) {
  MergeRatioMaskedBody::run(M, n_sel);
}

template <int R, int WAVES>
int launch_hamming_t(Ctx *c, const Query *q, uint32_t n_work_blocks, bool use_list, uint32_t split,
                     uint32_t lds_rows_cap, uint32_t nq_override = 0) {
  Map *m = c->map;
  const uint32_t nq = nq_override ? nq_override : q->n;  // override: only the first rows (the shared head of a sliced scan)
  const uint32_t q_chunk = (nq + split - 1) / split;
  // LDS slice of the split's query rows (gfx950: 160 KiB per CU; the cap trades slice reloads for occupancy)
  uint32_t lds_rows = q_chunk < lds_rows_cap ? q_chunk : lds_rows_cap;
  if (lds_rows == 0) lds_rows = 1;
  const size_t lds_bytes = (size_t)lds_rows * 64;
  auto kern = k_hamming_top2<R, WAVES>;
  SFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes));
  dim3 grid((n_work_blocks + WAVES * R - 1) / (WAVES * R), split);
  sfm_launch<HammingTop2Body<R, WAVES>>(c, kern, grid, dim3(WAVES * 64), (uint32_t)lds_bytes, m->d_bank,
                                        use_list ? c->d_block_list : nullptr, n_work_blocks, q->d_desc, nq, q_chunk,
                                        lds_rows, c->d_part);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// tuning hook: SFMLOC_K1_GEOM="R,WAVES,LDS_ROWS" overrides the geometry choice (read once)
struct K1Geom {
  int r = 0, waves = 0, lds_rows = 0;
};
const K1Geom &k1_override() {
  static K1Geom g = [] {
    K1Geom x;
    const char *e = getenv("SFMLOC_K1_GEOM");
    if (e) sscanf(e, "%d,%d,%d", &x.r, &x.waves, &x.lds_rows);
    return x;
  }();
  return g;
}

}  // namespace

int launch_tile_bank(const uint4 *d_rows, uint64_t row0, uint64_t n_rows_chunk, uint4 *d_bank, hipStream_t s) {
  if (n_rows_chunk == 0) return SFMLOC_OK;
  const uint64_t n = n_rows_chunk * 4;
  hipLaunchKernelGGL(k_tile_bank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_rows, row0, n_rows_chunk,
                     d_bank);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

static int launch_hamming_screened(Ctx *c, const Query *q, uint32_t n_work_blocks, bool use_list) {
  Map *m = c->map;
  constexpr int WAVES = 8;
  // Prefix length of the lower bound, in dwords.  With 10 of 16 the "some lane still below its threshold" vote
  // fires for ~3 % of the pairs on the bench data at ratio 0.6 (9 is past the cliff: half the lanes pass);
  // SFMLOC_K1_SCREEN_NW={8..13} / SFMLOC_K1_SCREEN_HEAD override the two knobs for tuning.
  static const int nw = [] {
    const char *e = getenv("SFMLOC_K1_SCREEN_NW");
    const int v = e ? atoi(e) : 10;
    return (v >= 8 && v <= 13) ? v : 10;
  }();
  static const uint32_t head = [] {
    const char *e = getenv("SFMLOC_K1_SCREEN_HEAD");
    const int v = e ? atoi(e) : (int)kScreenHead;
    return (uint32_t)((v >= 2 && v <= 4 * (int)kScreenHead) ? v : (int)kScreenHead);
  }();
  const uint32_t lds_rows = 512;
  const size_t lds_bytes = (size_t)lds_rows * 64;
  if (!c->cleared) SFM_HIP(hipMemsetAsync(c->d_n_flagged, 0, sizeof(uint32_t), c->stream));
  // query slices for a short block list: aim at 8 waves per SIMD (3 200 blocks in one slice leave the fullest SIMDs 4
  // waves and the average 3.1: 0.39 ms; four slices: 0.33 ms although each pays its own exact head), every slice at
  // least 6 heads long so that screening still pays
  static const int qsplit_env = [] {
    const char *e = getenv("SFMLOC_K1_QSPLIT");
    return e ? atoi(e) : 0;
  }();
  // The head is shared: one exact pass over the first `head` query rows (k_hamming_top2) seeds every slice, so a slice
  // costs no head of its own.  Four slices remain the optimum all the same (profiles/r02_k1_qsplit_sweep.txt: a
  // shortlisted query takes 0.95 / 0.93 / 0.89 / 1.02 / 1.29 ms with 1 / 2 / 4 / 8 / 16 slices): a workgroup's fixed
  // costs -- its bank rows, its query slice into LDS, two barriers -- stop amortising below ~500 query rows per slice.
  // Round 4: HOW MANY slices is chosen for the launch's last generation of workgroups.  A lone query's 3 150 blocks in four
  // slices are 1 576 workgroups for 768 places (three of the batched form per compute unit): two full generations and 40
  // stragglers that start when the second ends and run a third of a generation alone -- 48 of the scan's 268 us.  With s
  // slices a workgroup takes 1 / s of the time and the launch G(s) generations, the last one -- a fraction r of the places
  // -- about 0.45 + 0.55 r of a full one (fewer waves per SIMD run faster, not proportionally); each slice costs its
  // workgroups' fixed part (~1 %).  Measured on the headline's lone query, putMatch with 4 / 5 / 7 / 8 / 10 / 14 slices:
  // 0.323 / 0.301 / 0.291 / 0.300 / 0.296 / 0.294 ms.
  uint32_t qsplit = 1;
  if ((uint64_t)n_work_blocks < 32ull * (uint64_t)m->n_cu) {
    const uint32_t wg_per_slice = (n_work_blocks + WAVES - 1) / WAVES;
    const bool batched = n_work_blocks < 16u * (uint32_t)m->n_cu;  // (the form chosen below: 68 VGPRs, three per compute unit)
    const double places = (double)((batched ? 3 : 4) * m->n_cu);
    double best = 1e30;
    for (uint32_t s = 1; s <= 8; ++s) {
      if (s > 1 && (q->n - head) / s < 3 * head) break;  // (a slice keeps at least three heads of query rows)
      const double g = (double)wg_per_slice * s / places;
      const double full = floor(g), r = g - full;
      const double cost = (full + (r > 0.0 ? 0.45 + 0.55 * r : 0.0)) / s * (1.0 + 0.012 * s);
      if (cost < best) {
        best = cost;
        qsplit = s;
      }
    }
  }
  if (!c->k1_may_slice) qsplit = 1;  // other queries are queued on the GPU: their scans fill it, slices only add work
  if (qsplit_env >= 1 && qsplit_env <= 16) qsplit = (uint32_t)qsplit_env;
  const uint2 *head_part = nullptr;
  if (qsplit > 1) {
    int rc = launch_hamming_t<1, 8>(c, q, n_work_blocks, use_list, 1, 512, head);
    if (rc) return rc;
    head_part = c->d_part;
  }
  if (qsplit > 1 && !c->flagmask_zeroed)
    SFM_HIP(hipMemsetAsync(c->d_flagmask, 0, (size_t)n_work_blocks * sizeof(unsigned long long), c->stream));
  c->flagmask_zeroed = false;
#define K1_SCREEN(NW)                                                                                              \
  case NW:                                                                                                        \
    sfm_launch<HammingScreenBody<WAVES, NW, 1>>(                                                                  \
        c, k_hamming_screen<WAVES, NW, 1>, dim3((n_work_blocks + WAVES - 1) / WAVES, qsplit), dim3(WAVES * 64),   \
        (uint32_t)lds_bytes, m->d_bank, use_list ? c->d_block_list : nullptr, n_work_blocks, q->d_desc, q->n,     \
        lds_rows, m->d_ratio_cnt, c->d_flagmask, c->d_flagged, c->d_n_flagged, c->d_k1_counters, head,            \
        c->d_flagged_desc, c->rows_chunk_cap * 64, head_part);                                                    \
    break;
  // (the batched-tail form <.., 4> has 68 VGPRs against 48: while the GPU is shared the lean one leaves the other
  // queries' latency-bound stages more of the register file, +1.6 % queries/s; alone the two are equal.
  // SFMLOC_K1_SCREEN_BATCH = 4 / 1 forces one of them.)
  static const int env_batch = [] { const char *e = getenv("SFMLOC_K1_SCREEN_BATCH"); return e ? atoi(e) : 0; }();
  const bool batched_tail = env_batch ? env_batch == 4 : c->k1_may_slice;
  if (nw == 10 && batched_tail && n_work_blocks < 16u * (uint32_t)m->n_cu) {  // fewer than four waves per SIMD: batched tail
    sfm_launch<HammingScreenBody<WAVES, 10, 4>>(
        c, k_hamming_screen<WAVES, 10, 4>, dim3((n_work_blocks + WAVES - 1) / WAVES, qsplit), dim3(WAVES * 64),
        (uint32_t)lds_bytes, m->d_bank, use_list ? c->d_block_list : nullptr, n_work_blocks, q->d_desc, q->n, lds_rows,
        m->d_ratio_cnt, c->d_flagmask, c->d_flagged, c->d_n_flagged, c->d_k1_counters, head, c->d_flagged_desc,
        c->rows_chunk_cap * 64, head_part);
  } else if (nw == 10 && use_list) {  // any other scan of a view list (a shortlist while the GPU is shared, a long list)
    sfm_launch<HammingScreenBody<WAVES, 10, 1>>(
        c, k_hamming_screen_shortlist, dim3((n_work_blocks + WAVES - 1) / WAVES, qsplit), dim3(WAVES * 64),
        (uint32_t)lds_bytes, m->d_bank, use_list ? c->d_block_list : nullptr, n_work_blocks, q->d_desc, q->n, lds_rows,
        m->d_ratio_cnt, c->d_flagmask, c->d_flagged, c->d_n_flagged, c->d_k1_counters, head, c->d_flagged_desc,
        c->rows_chunk_cap * 64, head_part);
  } else {
    switch (nw) {
      K1_SCREEN(8) K1_SCREEN(9) K1_SCREEN(10) K1_SCREEN(11) K1_SCREEN(12) K1_SCREEN(13)
    }
  }
#undef K1_SCREEN
  // executed VALU lane-ops, deterministic part (the finished pairs are counted on the device): exact head 35 per
  // pair, screened tail 2*nw+1, plus (sfmloc_stats_read) 2*(16-nw)+5 per finished pair
  const uint64_t rows = (uint64_t)n_work_blocks * kBlockRows;
  // (a sliced scan shares ONE head)
  c->stats.hamming_lane_ops += rows * head * 35 + rows * (q->n - head) * (uint64_t)(2 * nw + 1);
  c->k1_finish_ops = 2 * (16 - nw) + 5;
  SFM_HIP(hipGetLastError());
  // the exact pass over the flagged rows: chunks of 64 rows x kRowSlices query slices, see k_hamming_rows
  constexpr int RW = 4, RS = 8;
  const uint32_t per = (q->n + RS - 1) / RS;
  const uint32_t rows_lds = per < 1024 ? per : 1024;  // 64 KiB of LDS at most; a longer slice does not fit ...
  const uint32_t chunk_cap = per <= 1024 ? c->rows_chunk_cap : 0;  // ... and then every chunk takes the walking path
  sfm_launch<HammingRowsBody<RW, RS>>(c, k_hamming_rows<RW, RS>, dim3(128 * RS), dim3(RW * 64), rows_lds * 64, m->d_bank,
                                      q->d_desc, q->n, c->d_flagged, c->d_n_flagged, c->d_rows_scratch, c->d_rows_arrivals,
                                      chunk_cap, rows_lds, c->d_flagged_desc, c->d_part);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

// The host loop of ctx_match_putative on the device, for a view selection that never leaves the GPU (the BoW
// shortlist, K8): ascending selected views -> ascending list of the 64-row bank blocks they overlap (a block shared
// by two selected views appears once) + per view the list position of its first block; the list is padded with
// kNoBlock up to `bound`, the launch size the host can know without reading the selection back.
struct BlocksFromViewsBody {
  static constexpr int kGangThreads = 1024;
  static __device__ __forceinline__ void run(const uint32_t *__restrict__ sel, uint32_t n_sel,
                                                    const uint32_t *__restrict__ view_off,
                                                    uint32_t *__restrict__ view_sel_out,
                                                    uint32_t *__restrict__ widx0, uint32_t *__restrict__ block_list,
                                                    uint32_t bound, unsigned long long *__restrict__ flagmask) {
    BlocksArgs B{sel, n_sel, view_off, view_sel_out, widx0, block_list, bound, flagmask};
    blocks_from_views_block(B);  // chain_device.h
  }
};
__global__ __launch_bounds__(1024) void k_blocks_from_views(const uint32_t *__restrict__ sel, uint32_t n_sel,
                                                    const uint32_t *__restrict__ view_off,
                                                    uint32_t *__restrict__ view_sel_out,
                                                    uint32_t *__restrict__ widx0, uint32_t *__restrict__ block_list,
                                                    uint32_t bound, unsigned long long *__restrict__ flagmask) {
  BlocksFromViewsBody::run(sel, n_sel, view_off, view_sel_out, widx0, block_list, bound, flagmask);
}

int launch_blocks_from_views(Ctx *c, const uint32_t *d_sel, uint32_t n_sel, uint32_t bound) {
  Map *m = c->map;
  sfm_launch<BlocksFromViewsBody>(c, k_blocks_from_views, dim3(1), dim3(1024), 0, d_sel, n_sel, m->d_view_off,
                                  c->d_view_sel, c->d_view_widx0, c->d_block_list, bound, c->d_flagmask);
  SFM_HIP(hipGetLastError());
  c->flagmask_zeroed = true;
  return SFMLOC_OK;
}

int launch_hamming_top2(Ctx *c, const Query *q, uint32_t n_work_blocks, bool use_list, uint32_t split) {
  Map *m = c->map;
  if (n_work_blocks == 0 || q->n == 0) return SFMLOC_OK;
  // (below ~12 heads' worth of query rows the exact head and the second launch eat the saving:
  // profiles/r01_k1_screen_on_akaze_descriptors.jsonl)
  c->last_screened = false;
  if (split == 1 && m->params.exact_rows == 0 && q->n >= 12 * kScreenHead && !k1_override().r) {
    c->last_screened = true;
    return launch_hamming_screened(c, q, n_work_blocks, use_list);
  }
  int R, W, L;
  c->stats.hamming_lane_ops += (uint64_t)n_work_blocks * kBlockRows * q->n * 35;
  const K1Geom &o = k1_override();
  if (o.r) {
    R = o.r;
    W = o.waves;
    L = o.lds_rows;
  } else {
    // Measured on MI355X (profiles/r01_valu_rates.jsonl, profiles/r01_k1_lab.txt): v_xor_b32 issues at 2
    // cycles per wave64, v_bcnt_u32_b32 / v_med3 / v_lshl_or at ~4, and one wave alone gets an instruction
    // only every ~4.8 cycles, so the loop reaches the VALU issue rate of its instruction mix only with >= 4
    // waves per SIMD.  LDS vs scalar-cache query reads, register prefetch of the query row and split popcount
    // chains all time within 5 % of each other: the loop is VALU-issue bound.  Geometry: one bank block per wave
    // (42 VGPRs), 8-wave workgroups, 32 KiB LDS slices -> 4 workgroups = 32 waves per CU.
    const uint64_t wave_blocks = (uint64_t)n_work_blocks * split;
    if (wave_blocks >= (uint64_t)m->n_cu * 32) {
      R = 1; W = 8; L = 512;
    } else {
      R = 1; W = 4; L = 512;
    }
  }
#define K1_CASE(r, w)   if (R == r && W == w) return launch_hamming_t<r, w>(c, q, n_work_blocks, use_list, split, (uint32_t)L);
  K1_CASE(4, 16)
  K1_CASE(2, 16)
  K1_CASE(1, 16)
  K1_CASE(4, 8)
  K1_CASE(2, 8)
  K1_CASE(1, 8)
  K1_CASE(4, 4)
  K1_CASE(2, 4)
  K1_CASE(1, 4)
#undef K1_CASE
  set_error("unsupported K1 geometry R=%d WAVES=%d", R, W);
  return SFMLOC_EINVAL;
}

int launch_merge_ratio_compact(Ctx *c, const Query *q, uint32_t n_sel, bool all_views, uint32_t split,
                               uint32_t n_work_blocks) {
  (void)q;
  Map *m = c->map;
  if (n_sel == 0) return SFMLOC_OK;
  if (c->last_screened && split == 1) {
    MergeMaskedArgs M;
    M.enabled = 1;
    M.part = c->d_part;
    M.flagmask = c->d_flagmask;
    M.view_sel = all_views ? nullptr : c->d_view_sel;
    M.view_widx0 = all_views ? nullptr : c->d_view_widx0;
    M.view_off = m->d_view_off;
    M.ratio_cnt = m->d_ratio_cnt;
    M.view_count = c->d_view_count;
    M.match_i = c->d_match_i;
    M.match_key = c->d_match_key;
    if (c->defer_merge) {  // the caller runs K3 next on this context: its per-view workgroups merge their own views
      c->deferred_merge = M;
      c->merge_is_deferred = true;
      return SFMLOC_OK;
    }
    sfm_launch<MergeRatioMaskedBody>(c, k_merge_ratio_masked, dim3((n_sel + 3) / 4), dim3(256), 0, M, n_sel);
    SFM_HIP(hipGetLastError());
    return SFMLOC_OK;
  }
  sfm_launch<MergeRatioCompactBody>(c, k_merge_ratio_compact, dim3((n_sel + 3) / 4), dim3(256), 0, c->d_part,
                     c->last_screened ? c->d_flagmask : nullptr, n_work_blocks, split, all_views ? nullptr : c->d_view_sel,
                     all_views ? nullptr : c->d_view_widx0, n_sel, m->d_view_off, m->d_ratio_cnt, c->d_view_count,
                     c->d_match_i, c->d_match_key);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

int launch_merge_masked_now(Ctx *c, uint32_t n_sel) {
  sfm_launch<MergeRatioMaskedBody>(c, k_merge_ratio_masked, dim3((n_sel + 3) / 4), dim3(256), 0, c->deferred_merge, n_sel);
  SFM_HIP(hipGetLastError());
  return SFMLOC_OK;
}

}  // namespace sfmloc
