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
      R.cand_header[0] = R.cand_header[1] = R.cand_header[2] = R.cand_header[3] = 0;  // kPartHeaderBytes = 16
      R.view_stats[0] = R.view_stats[1] = 0;
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
__device__ __forceinline__ void blocks_from_views_block(const BlocksArgs &B) {
  constexpr uint32_t kRows = 64;
  __shared__ uint32_t s_last[1024];  // inclusive running max of (last block + 1) over the non-empty views so far
  __shared__ uint32_t s_cnt[1024];   // inclusive scan of the blocks each view adds
  __shared__ uint32_t s_carry_last, s_carry_cnt;
  const uint32_t tid = threadIdx.x;
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
    s_last[tid] = nonempty ? b1 + 1 : 0;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // inclusive max-scan
      const uint32_t o = tid >= off ? s_last[tid - off] : 0;
      __syncthreads();
      s_last[tid] = max(s_last[tid], o);
      __syncthreads();
    }
    // last block (+1) of the nearest earlier non-empty selected view: views ascend, so it is the running maximum
    const uint32_t prev = max(s_carry_last, tid ? s_last[tid - 1] : 0u);
    const bool share = nonempty && prev != 0 && prev - 1 == b0;
    const uint32_t add = nonempty ? (b1 - b0 + 1 - (share ? 1u : 0u)) : 0u;
    s_cnt[tid] = add;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // inclusive sum-scan
      const uint32_t o = tid >= off ? s_cnt[tid - off] : 0;
      __syncthreads();
      s_cnt[tid] += o;
      __syncthreads();
    }
    const uint32_t start = s_carry_cnt + s_cnt[tid] - add;
    if (k < B.n_sel) {
      B.view_sel_out[k] = v;
      B.widx0[k] = nonempty ? (share ? start - 1 : start) : 0u;
      for (uint32_t b = b0 + (share ? 1u : 0u), w = start; nonempty && b <= b1; ++b, ++w)
        if (w < B.bound) B.block_list[w] = b;
    }
    __syncthreads();
    if (tid == 1023) {
      s_carry_last = max(s_carry_last, s_last[1023]);
      s_carry_cnt += s_cnt[1023];
    }
    __syncthreads();
  }
  for (uint32_t w = s_carry_cnt + tid; w < B.bound; w += 1024) B.block_list[w] = kChainNoBlock;
  // a sliced scan ORs its row flags into the per-block masks: clear them here rather than with one more launch
  for (uint32_t w = tid; w < B.bound; w += 1024) B.flagmask[w] = 0ull;
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
