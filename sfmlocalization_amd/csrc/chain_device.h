// Device-side pieces shared by the kernels that start a query's chain, so that the single workgroup which finishes the
// view shortlist (k_bow_topk, or k_bow_merge_select on a shard) also clears the query's counters and builds the
// bank-block list: two dependent launches less per query (k_query_reset and k_blocks_from_views remain for callers that
// drive the stages one at a time or pass a host-side view list).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace sfmloc {

constexpr uint32_t kChainNoBlock = 0xFFFFFFFFu;  // = kNoBlock (sfmloc_internal.h)

// Every counter / flag the stages of one query start from.
struct QueryResetArgs {
  uint32_t *view_count, *geo_count;
  uint32_t n_views;
  unsigned long long *best64;
  uint32_t nq;
  uint32_t *n_flagged;
  int *status;
  uint32_t *cand_header, *view_stats, *ms_n;
  uint32_t *fl_count;  // K3's list of views for k_fmatrix_large (maps with views above 2 048 rows), or null
};
__device__ __forceinline__ void query_reset_items(const QueryResetArgs &R, uint32_t first, uint32_t stride) {
  const uint32_t n = (R.n_views + 1 > R.nq) ? R.n_views + 1 : R.nq;
  for (uint32_t t = first; t < n; t += stride) {
    if (t <= R.n_views) {  // <= : the phantom view's slot too
      R.view_count[t] = 0;
      R.geo_count[t] = 0;
    }
    if (t < R.nq) R.best64[t] = ~0ull;
    if (t == 0) {
      *R.n_flagged = 0;
      *R.status = 0;
      if (R.fl_count) *R.fl_count = 0;
      R.cand_header[0] = R.cand_header[1] = R.cand_header[2] = R.cand_header[3] = 0;  // kPartHeaderBytes = 16
      R.view_stats[0] = R.view_stats[1] = R.view_stats[2] = 0;
      *R.ms_n = 0;
    }
  }
}

// The selected views (ascending indices) -> the ascending list of the 64-row bank blocks they overlap, padded with
// "no block" up to `bound`; one 1024-thread workgroup.
struct BlocksArgs {
  const uint32_t *sel;
  uint32_t n_sel;
  const uint32_t *view_off;
  uint32_t *view_sel_out, *widx0, *block_list;
  uint32_t bound;
  unsigned long long *flagmask;
};
// inclusive scan of one value per thread over a 1024-thread workgroup, OP = max or +: shuffles inside a wave, the sixteen
// wave totals through LDS (two barriers; the log-step scan through LDS this replaces took twenty, twice per call -- 10 of
// k_bow_topk's 28 us on a lone query's shortlist)
template <bool kMax>
__device__ __forceinline__ uint32_t chain_block_scan_1024(uint32_t v, uint32_t *wave_tot /*[16] LDS*/) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc = kMax ? max(inc, o) : inc + o;
  }
  __syncthreads();  // wave_tot of an earlier call is no longer read
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; ++w) {
    const uint32_t t = wave_tot[w];
    if (w < wave) before = kMax ? max(before, t) : before + t;
  }
  return kMax ? max(inc, before) : inc + before;
}
__device__ __forceinline__ void blocks_from_views_block(const BlocksArgs &B) {
  constexpr uint32_t kRows = 64;
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry_last, s_carry_cnt;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (tid == 0) s_carry_last = 0, s_carry_cnt = 0;
  __syncthreads();
  for (uint32_t base = 0; base < B.n_sel; base += 1024) {
    const uint32_t k = base + tid;
    uint32_t v = 0, r0 = 0, r1 = 0;
    if (k < B.n_sel) {
      v = B.sel[k];
      r0 = B.view_off[v];
      r1 = B.view_off[v + 1];
    }
    const bool nonempty = r1 > r0;
    const uint32_t b0 = r0 / kRows, b1 = nonempty ? (r1 - 1) / kRows : 0;
    // inclusive running max of (last block + 1) over the non-empty views so far
    const uint32_t last_inc = chain_block_scan_1024<true>(nonempty ? b1 + 1 : 0u, s_wave);
    // last block (+1) of the nearest earlier non-empty selected view: views ascend, so it is the running maximum
    uint32_t last_before = __shfl_up(last_inc, 1, 64);  // (the thread before me; lane 0: the waves before mine)
    if (lane == 0) {
      last_before = 0;
      for (uint32_t w = 0; w < (tid >> 6); ++w) last_before = max(last_before, s_wave[w]);
    }
    const uint32_t prev = max(s_carry_last, last_before);
    const bool share = nonempty && prev != 0 && prev - 1 == b0;
    const uint32_t add = nonempty ? (b1 - b0 + 1 - (share ? 1u : 0u)) : 0u;
    // inclusive scan of the blocks each view adds
    const uint32_t cnt_inc = chain_block_scan_1024<false>(add, s_wave);
    const uint32_t start = s_carry_cnt + cnt_inc - add;
    if (k < B.n_sel) {
      B.view_sel_out[k] = v;
      B.widx0[k] = nonempty ? (share ? start - 1 : start) : 0u;
      for (uint32_t b = b0 + (share ? 1u : 0u), w = start; nonempty && b <= b1; ++b, ++w)
        if (w < B.bound) B.block_list[w] = b;
    }
    __syncthreads();  // (every thread has read the carries)
    if (tid == 1023) {
      s_carry_last = max(s_carry_last, last_inc);
      s_carry_cnt += cnt_inc;
    }
    __syncthreads();
  }
  for (uint32_t w = s_carry_cnt + tid; w < B.bound; w += 1024) B.block_list[w] = kChainNoBlock;
  // a sliced scan ORs its row flags into the per-block masks: clear them here rather than with one more launch
  for (uint32_t w = tid; w < B.bound; w += 1024) B.flagmask[w] = 0ull;
}

// K2 after a screened scan (k_merge_ratio_masked), one wave per selected view: only rows whose mask bit is set can be
// matches; a lane walks the set bits of one 64-row block, so all partial-result loads of the view are in flight at
// once, and a wave scan of the per-lane accept counts places the accepted rows in ascending order.  Shared by the K2
// kernel and by K3's per-view workgroup, which runs it for its own view when the merge was left to it (one launch less).
struct MergeMaskedArgs {
  int enabled;
  const uint2 *part;
  const unsigned long long *flagmask;
  const uint32_t *view_sel, *view_widx0;
  const uint32_t *view_off;
  const uint16_t *ratio_cnt;
  uint32_t *view_count, *match_i, *match_key;
};
__device__ __forceinline__ void merge_ratio_masked_view(const MergeMaskedArgs &M, uint32_t gw, uint32_t lane) {
  constexpr uint32_t kNoMatch = 0xFFFFFFFFu;  // SFMLOC_NOMATCH
  const uint32_t v = M.view_sel ? M.view_sel[gw] : gw;
  const uint32_t off = M.view_off[v], end = M.view_off[v + 1];
  const uint32_t blk0 = off >> 6;
  const uint32_t widx0 = M.view_sel ? M.view_widx0[gw] : blk0;
  const uint32_t n_blk = (end > off) ? (((end - 1) >> 6) - blk0 + 1) : 0;
  uint32_t base = 0;
  for (uint32_t c0 = 0; c0 < n_blk; c0 += 64) {  // 64 blocks (4096 rows) per pass
    const uint32_t rel = c0 + lane;
    unsigned long long w = 0;
    if (rel < n_blk) {
      w = M.flagmask[widx0 + rel];
      // rows of the block that belong to neighbouring views
      const uint64_t r_lo = (uint64_t)(blk0 + rel) << 6;
      if (r_lo < off) w &= ~0ull << (off - r_lo);
      if (r_lo + 64 > end) w &= (end > r_lo) ? (~0ull >> (r_lo + 64 - end)) : 0ull;
    }
    // pass 1: count the accepted rows of my block
    uint32_t mine = 0;
    unsigned long long acc_bits = 0;
    for (unsigned long long t = w; t; t &= t - 1) {
      const uint32_t bit = (uint32_t)__builtin_ctzll(t);
      const uint2 p = M.part[((uint64_t)(widx0 + rel)) * 64 + bit];
      if (p.y != kNoMatch && (p.x >> 16) < (uint32_t)M.ratio_cnt[p.y >> 16]) {
        acc_bits |= 1ull << bit;
        ++mine;
      }
    }
    // exclusive scan over the lanes
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if ((int)lane >= d) incl += o;
    }
    uint32_t pos = base + incl - mine;
    // pass 2: write them in ascending row order
    for (unsigned long long t = acc_bits; t; t &= t - 1) {
      const uint32_t bit = (uint32_t)__builtin_ctzll(t);
      const uint2 p = M.part[((uint64_t)(widx0 + rel)) * 64 + bit];
      M.match_i[off + pos] = (uint32_t)((((uint64_t)(blk0 + rel)) << 6) + bit - off);
      M.match_key[off + pos] = p.x;
      ++pos;
    }
    base += __shfl(incl, 63, 64);
  }
  if (lane == 0) M.view_count[v] = base;
}

// what the shortlist's last workgroup does on top of the shortlist (enabled = 0: nothing); keys_out (k_bow_topk only):
// the selection as sortable keys for the sharded shortlist, padded to k_out entries
struct ChainArgs {
  int enabled;
  QueryResetArgs reset;
  BlocksArgs blocks;
  unsigned long long *keys_out;
  const uint32_t *key_view_id;
  uint32_t k_out;
};
__device__ __forceinline__ void chain_after_shortlist(const ChainArgs &C) {
  if (!C.enabled) return;
  __syncthreads();  // the selection (this workgroup's own global stores) is complete and visible to it
  query_reset_items(C.reset, threadIdx.x, blockDim.x);
  blocks_from_views_block(C.blocks);
}

}  // namespace sfmloc
