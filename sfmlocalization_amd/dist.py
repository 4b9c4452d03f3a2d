"""Sharded localisation over torch.distributed: one process per GPU, the descriptor bank sharded
by contiguous view ranges, ONE all-gather of per-shard candidate parts per query batch (RCCL over xGMI when the
backend is "nccl"; gloo on CPU in the tests), then the 2D-3D selection + P3P of each query on one rank.

The collective layer is independent of what computes the parts: `HipShardCompute` drives the C ABI
(sfmloc_shard_begin / _export / sfmloc_merge_begin); the CPU tests plug in a stand-in built on the oracle.
"""
import os

import numpy as np

# one candidate as the C ABI lays it out (sfmloc_internal.h Candidate; 40 bytes) and the 16-byte part header
CANDIDATE_DTYPE = np.dtype([("order", "<u8"), ("qfeat", "<u4"), ("landmark_id", "<u4"), ("X", "<f8", (3,))])
PART_HEADER_BYTES = 16


def part_bytes(cap):
    return PART_HEADER_BYTES + cap * CANDIDATE_DTYPE.itemsize


def order_key(dist, view_id, pos):
    """dist<<48 | view_id<<24 | position in the view's geometric list: smaller wins, ties resolve as the
    reference's sequential scan would (SfMDataUtils.cpp:109 keeps the first on equal distance)."""
    return (np.uint64(dist) << np.uint64(48)) | (np.uint64(view_id & 0xFFFFFF) << np.uint64(24)) | np.uint64(pos & 0xFFFFFF)


def pack_part(cands, cap):
    """cands: structured array of CANDIDATE_DTYPE -> uint8[part_bytes(cap)] (header carries the true count)."""
    buf = np.zeros(part_bytes(cap), np.uint8)
    buf[:4] = np.frombuffer(np.uint32(len(cands)).tobytes(), np.uint8)
    n = min(len(cands), cap)
    if n:
        buf[PART_HEADER_BYTES:PART_HEADER_BYTES + n * CANDIDATE_DTYPE.itemsize] = np.frombuffer(
            np.ascontiguousarray(cands[:n]).tobytes(), np.uint8)
    return buf


def unpack_part(buf, cap):
    buf = np.ascontiguousarray(buf, np.uint8)
    n = int(np.frombuffer(buf[:4].tobytes(), np.uint32)[0])
    if n > cap:
        raise OverflowError(f"a shard produced {n} candidates, parts hold {cap}")
    return np.frombuffer(buf[PART_HEADER_BYTES:PART_HEADER_BYTES + n * CANDIDATE_DTYPE.itemsize].tobytes(),
                         CANDIDATE_DTYPE)


def reduce_candidates(cands):
    """What a shard materialises of its candidates (k_emit_min / k_emit_win): per query feature only the one with the
    smallest order key -- matchProviderToMatchSet keeps one per feature anyway.  Host restatement for the tests."""
    cands = np.asarray(cands, CANDIDATE_DTYPE)
    if len(cands) == 0:
        return cands
    o = np.lexsort((cands["order"], cands["qfeat"]))
    c = cands[o]
    first = np.ones(len(c), bool)
    first[1:] = c["qfeat"][1:] != c["qfeat"][:-1]
    return c[first]


def shard_views(view_off, world):
    """Contiguous view ranges [v0, v1) per rank, balanced by descriptor count; a view is never split
    (so the per-view >=16 filter and F-matrix RANSAC stay shard-local)."""
    view_off = np.asarray(view_off, dtype=np.int64)
    nv = len(view_off) - 1
    total = int(view_off[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        v = int(np.searchsorted(view_off, target, side="left"))
        v = min(max(v, cuts[-1]), nv)
        cuts.append(v)
    cuts.append(nv)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _k_best(dist, ids, k):
    """Indices of the k smallest (distance, view id) pairs, in that order, without sorting all of them."""
    n = len(dist)
    if k >= n:
        return np.lexsort((ids, dist))[:k]
    t = np.partition(dist, k - 1)[k - 1]
    less = np.nonzero(dist < t)[0]
    eq = np.nonzero(dist == t)[0]                      # ascending index = ascending view id
    sel = np.concatenate([less, eq[:k - len(less)]])
    return sel[np.lexsort((ids[sel], dist[sel]))]


def merge_bow_shortlists(local_dist, local_view_id, k, world, all_gather):
    """Sharded BoW shortlist: `local_dist[i]` = this rank's distance for its view `local_view_id[i]`
    (ascending ids).  Every rank contributes its k best (distance, view id) pairs; the global k best -- ties to the
    lower view id, as sfmloc_bow_select does on one GPU -- are kept, and the function returns the ascending LOCAL
    indices among them (what sfmloc_shard_begin takes as view_sel).  `all_gather(arr [k, 2] f64) -> [world, k, 2]`."""
    local_dist = np.asarray(local_dist, np.float32)
    local_view_id = np.asarray(local_view_id, np.int64)
    order = _k_best(local_dist, local_view_id, k)
    mine = np.full((k, 2), np.inf, np.float64)
    mine[:len(order), 0] = local_dist[order]
    mine[:len(order), 1] = local_view_id[order]
    allp = np.asarray(all_gather(mine)).reshape(world * k, 2)
    allp = allp[np.isfinite(allp[:, 0])]
    best = allp[np.lexsort((allp[:, 1], allp[:, 0]))[:k]]
    return np.nonzero(np.isin(local_view_id, best[:, 1].astype(np.int64)))[0].astype(np.uint32)


def bow_key(dist_f32, view_id):
    """The sortable key a shard publishes per shortlisted view: float32 distance bits << 32 | view id (non-negative
    floats order like their bit patterns; ties fall to the lower view id, as in the unsharded selection)."""
    bits = np.asarray(dist_f32, np.float32).view(np.uint32).astype(np.uint64)
    return (bits << np.uint64(32)) | np.asarray(view_id, np.uint64)


BOW_KEY_PAD = np.uint64(0xFFFFFFFFFFFFFFFF)


def select_from_keys(gathered_keys, k, local_view_id):
    """gathered_keys [world, k] u64 (one query) -> ascending LOCAL indices of this shard's views among the global k
    best.  Host restatement of k_bow_merge_select (tests, CPU stand-ins)."""
    allk = np.asarray(gathered_keys, np.uint64).ravel()
    allk = np.sort(allk[allk != BOW_KEY_PAD])[:k]
    ids = (allk & np.uint64(0xFFFFFFFF)).astype(np.int64)
    return np.nonzero(np.isin(np.asarray(local_view_id, np.int64), ids))[0].astype(np.uint32)


def packed_cands_offset(n_queries):
    return (16 + 8 * n_queries + 15) // 16 * 16


def packed_bytes(n_queries, budget):
    """One shard's exchange buffer for a batch (the C ABI's sfmloc_packed_bytes): header {u32 total, n_queries, budget,
    flags}, u32 count[B], u32 offset[B], then the candidates of all B queries back to back."""
    return packed_cands_offset(n_queries) + budget * CANDIDATE_DTYPE.itemsize


def pack_batch(cands_list, budget):
    """Host restatement of sfmloc_shard_export_packed over a whole batch (tests, CPU stand-ins)."""
    B = len(cands_list)
    buf = np.zeros(packed_bytes(B, budget), np.uint8)
    hdr = buf[:16 + 8 * B].view(np.uint32)
    base = packed_cands_offset(B)
    off = 0
    for i, c in enumerate(cands_list):
        n = len(c)
        fits = off + n <= budget
        if fits:
            hdr[4 + i], hdr[4 + B + i] = n, off
            if n:
                buf[base + off * 40: base + (off + n) * 40] = np.frombuffer(np.ascontiguousarray(c).tobytes(), np.uint8)
        else:
            hdr[3] |= 1
        off += n
    hdr[0], hdr[1], hdr[2] = off, B, budget
    return buf


def unpack_batch(buf, qi):
    """Candidates of query qi in one shard's packed part."""
    buf = np.ascontiguousarray(buf, np.uint8)
    B = int(buf[4:8].view(np.uint32)[0])
    hdr = buf[:16 + 8 * B].view(np.uint32)
    n, off = int(hdr[4 + qi]), int(hdr[4 + B + qi])
    base = packed_cands_offset(B)
    return np.frombuffer(buf[base + off * 40: base + (off + n) * 40].tobytes(), CANDIDATE_DTYPE)


# ----- images in on several ranks: the extracted features of a batch, exchanged like the candidates ------------------
# The rank that owns query i (i mod world) extracts it; ONE all-gather per batch hands every rank every query as a
# fixed-capacity block {u32 n, width, height, 0 | desc [cap x 64] (rows beyond n zero) | kpt [cap x 2 f32] |
# kpt after the .feat round trip [cap x 2 f32] | BoW vector [bow_dim f32]}, and the queries of the batch are VIEWS
# into the gathered buffer (sfmloc_query_create_view): nothing is copied or allocated per query.
FEATURE_HEADER_BYTES = 16


def feature_block_layout(cap, bow_dim):
    """-> (offset of desc, kpt, kpt6, bow, block bytes); cap must be a multiple of 64 (the bank's row blocks)"""
    assert cap % 64 == 0 and cap > 0
    o_desc = FEATURE_HEADER_BYTES
    o_kpt = o_desc + cap * 64
    o_kpt6 = o_kpt + cap * 8
    o_bow = o_kpt6 + cap * 8
    total = (o_bow + bow_dim * 4 + 15) // 16 * 16
    return o_desc, o_kpt, o_kpt6, o_bow, total


def pack_features(desc, kpt_xy, kpt6_xy, width, height, bow, cap, bow_dim):
    """one query's block (uint8 [block bytes]); kpt6_xy = capi.feat_round_trip(kpt_xy)"""
    o_desc, o_kpt, o_kpt6, o_bow, total = feature_block_layout(cap, bow_dim)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 64)
    n = desc.shape[0]
    if n > cap:
        raise OverflowError(f"{n} features > the exchange's capacity of {cap} per query")
    b = np.zeros(total, np.uint8)
    b[:16].view(np.uint32)[:3] = (n, width, height)
    b[o_desc:o_desc + n * 64] = desc.ravel()
    b[o_kpt:o_kpt + n * 8] = np.ascontiguousarray(kpt_xy, np.float32).reshape(-1).view(np.uint8)
    b[o_kpt6:o_kpt6 + n * 8] = np.ascontiguousarray(kpt6_xy, np.float32).reshape(-1).view(np.uint8)
    if bow_dim:
        b[o_bow:o_bow + bow_dim * 4] = np.ascontiguousarray(bow, np.float32).reshape(-1)[:bow_dim].view(np.uint8)
    return b


def unpack_features(block, cap, bow_dim):
    """-> (desc [n, 64], kpt [n, 2], kpt6 [n, 2], width, height, bow [bow_dim])  (host side of the layout: tests,
    stand-in compute objects)"""
    o_desc, o_kpt, o_kpt6, o_bow, total = feature_block_layout(cap, bow_dim)
    block = np.ascontiguousarray(block, np.uint8)
    n, w, h = (int(x) for x in block[:16].view(np.uint32)[:3])
    return (block[o_desc:o_desc + n * 64].reshape(n, 64).copy(),
            block[o_kpt:o_kpt + n * 8].view(np.float32).reshape(n, 2).copy(),
            block[o_kpt6:o_kpt6 + n * 8].view(np.float32).reshape(n, 2).copy(), w, h,
            block[o_bow:o_bow + bow_dim * 4].view(np.float32).copy())


class ShardedLocalizer:
    """One batch = (optional) sharded BoW shortlist -> stage 1 on every shard -> ONE all-gather of the shards' packed
    candidate parts -> stage 2 (2D-3D selection + P3P) of query i on rank i mod world.

    A shard sends what it found: one packed part per batch holding up to `budget` = B x budget_per_query candidates
    (a few MB for 256 queries) instead of B fixed-capacity parts.  Every rank sees every part header after the gather,
    so all ranks agree -- without another collective -- when some shard needed more, and the batch is exchanged again
    with a budget that fits (and the larger budget is kept for the batches that follow).

    compute protocol (HipShardCompute below; the CPU tests plug in a stand-in built on the oracle):
      stage1(queries, slot, budget) -> uint8 [packed_bytes(B, budget)] on compute.device (possibly still being written)
      bow_keys(queries, knn, slot) -> int64 [B, knn] (u64 bit patterns)        } only for bow_knn > 0
      stage1_bow(queries, gathered_keys [world, B, knn], knn, slot, budget)     }
      before_collective(slot) / after_collective(slot): order the slot's streams against the collective's (optional)
      stage2(indices, gathered [world, packed_bytes(B, budget)], slot, budget) -> {index: result} for the queries owned here.
    A compute object with `n_slots >= 2` lets two batches overlap (localize_stream)."""

    def __init__(self, compute, budget_per_query=256, rank=None, world=None, group=None, always_gather=False,
                 n_views_global=None):
        import torch.distributed as dist
        self.dist = dist
        self.always_gather = always_gather   # run the collective even on one rank (rehearsal of the N>1 path)
        self.compute = compute
        self.budget_per_query = int(budget_per_query)
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_slots = int(getattr(compute, "n_slots", 1))
        self.n_views_global = n_views_global
        self.reset_counters()

    def reset_counters(self):
        self._coll_events = []
        self._coll_stats = {}
        self._n_batches = self._n_queries = self._n_redo = 0
        self._bytes_parts = self._bytes_keys = self._bytes_feats = 0
        self._max_total = 0

    def counters(self):
        nb = max(1, self._n_batches)
        self._fold_coll_events()
        coll_ms = {k: {"n": n, "mean_ms": tot / n, "max_ms": mx} for k, (n, tot, mx) in self._coll_stats.items() if n}
        return {"batches": self._n_batches, "queries": self._n_queries,
                "process_group_size": int(self.dist.get_world_size(self.group)) if self.dist.is_initialized() else 1,
                "collective_ms_on_the_comm_stream": coll_ms,
                "candidate_allgather_bytes_per_batch_per_rank": self._bytes_parts / nb,
                "bow_key_allgather_bytes_per_batch_per_rank": self._bytes_keys / nb,
                **({"feature_allgather_bytes_per_batch_per_rank": self._bytes_feats / nb} if self._bytes_feats else {}),
                "budget_candidates_per_query": self.budget_per_query,
                "max_candidates_of_one_shard_for_one_batch": int(self._max_total),
                "batches_exchanged_again_with_a_larger_budget": self._n_redo}

    _COLL_EVENTS_CAP = 256   # pending (start, end) event pairs; folded into running statistics beyond that

    def _fold_coll_events(self, keep=0):
        """Elapsed times of the recorded collectives -> running {what: (n, sum, max)}; the events are dropped, so a
        service that never resets its counters holds at most _COLL_EVENTS_CAP pairs (oldest first: those have long
        completed; synchronising on them does not stall the pipeline)."""
        ev, self._coll_events = self._coll_events[:len(self._coll_events) - keep], self._coll_events[len(self._coll_events) - keep:]
        for what, e0, e1 in ev:
            try:
                e1.synchronize()
                ms = float(e0.elapsed_time(e1))
            except Exception:  # noqa: BLE001
                continue
            n, tot, mx = self._coll_stats.get(what, (0, 0.0, 0.0))
            self._coll_stats[what] = (n + 1, tot + ms, max(mx, ms))

    def owner(self, i):
        return i % self.world

    # ----- collectives --------------------------------------------------------------------------------------------
    def _comm_stream(self):
        return getattr(self.compute, "comm", None)

    def _all_gather(self, send, what="candidates"):
        """send [n, ...] -> ([world, n, ...] on send's device, event) -- on the compute object's collective stream (if
        any); the event marks the gather's completion on that stream (None without one)."""
        import contextlib
        import torch
        out = torch.empty((self.world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        comm = self._comm_stream()
        scope = torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()
        timed = comm is not None and send.is_cuda
        with scope:
            if timed:
                t_start = torch.cuda.Event(enable_timing=True)
                t_start.record(comm)
            if self.world > 1 or self.always_gather:
                flat = out.view((self.world * send.shape[0],) + tuple(send.shape[1:]))
                if send.is_cuda and self.dist.get_backend(self.group) == "gloo":
                    # rehearsal of the N>1 path without RCCL (e.g. two ranks sharing one GPU): stage through the host
                    if comm is not None:
                        comm.synchronize()
                    host = torch.empty(flat.shape, dtype=send.dtype)
                    self.dist.all_gather_into_tensor(host, send.cpu().contiguous(), group=self.group)
                    flat.copy_(host)
                else:
                    self.dist.all_gather_into_tensor(flat, send.contiguous(), group=self.group)
            else:
                out[0].copy_(send)
            if timed:   # the collective's own duration on its stream (read in counters(), after the events completed)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(comm)
                self._coll_events.append((what, t_start, ev))
                if len(self._coll_events) > self._COLL_EVENTS_CAP:
                    self._fold_coll_events(keep=self._COLL_EVENTS_CAP // 2)
            else:
                ev = comm.record_event() if comm is not None else None
        return out, ev

    def _before(self, slot):
        if hasattr(self.compute, "before_collective"):
            self.compute.before_collective(slot)

    def _after(self, slot):
        if hasattr(self.compute, "after_collective"):
            self.compute.after_collective(slot)

    # ----- images in: the batch's features ------------------------------------------------------------------------
    def gather_queries(self, own_blocks, n_queries, cap, bow_dim, slot=0):
        """own_blocks {i: uint8 block (pack_features)} for the queries i of the batch that this rank owns and has
        extracted -> the batch's n_queries queries, as the compute object makes them over the gathered buffer
        (compute.query_views(gathered [world, per_rank, block], cap, bow_dim, n_queries)); every rank gets the same
        list.  One all-gather of per_rank x block bytes per rank."""
        import torch
        per = -(-n_queries // self.world)
        total = feature_block_layout(cap, bow_dim)[4]
        send = np.zeros((per, total), np.uint8)
        for i, b in own_blocks.items():
            assert self.owner(i) == self.rank and 0 <= i < n_queries
            send[i // self.world] = b
        import contextlib
        dev = getattr(self.compute, "device", None)
        t = torch.from_numpy(send)
        if dev is not None and dev.type == "cuda":
            comm = self._comm_stream()     # (the upload on the stream the collective runs on: ordered before it)
            with (torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()):
                t = t.pin_memory().to(dev, non_blocking=True)
        gathered, ev = self._all_gather(t, "features")
        if ev is not None:
            ev.synchronize()
        self._bytes_feats += t.numel()
        return self.compute.query_views(gathered, cap, bow_dim, n_queries, slot)

    # ----- one batch ----------------------------------------------------------------------------------------------
    def _use_bow(self, bow_knn):
        """localization.cpp:346: the shortlist applies only when more than knn candidate views remain."""
        if not bow_knn:
            return False
        return self.n_views_global is None or self.n_views_global > bow_knn

    def _stage1(self, queries, slot, budget, bow_knn=0, keys_all=None):
        if self._use_bow(bow_knn):
            if keys_all is None:
                keys = self.compute.bow_keys(queries, bow_knn, slot)
                self._before(slot)
                keys_all, _ = self._all_gather(keys, "bow_keys")
                self._after(slot)
                self._bytes_keys += keys.numel() * keys.element_size()
            part = self.compute.stage1_bow(queries, keys_all, bow_knn, slot, budget)
        else:
            part = self.compute.stage1(queries, slot, budget)
        self._before(slot)
        gathered, ev = self._all_gather(part)
        # the slot's contexts do nothing more until stage 2 of THIS batch: they wait for this gather right away, so that
        # a later batch's collectives on the same stream never stand between a batch and its own stage 2
        self._after(slot)
        self._bytes_parts += part.numel() * part.element_size()
        return gathered, keys_all, ev

    def _begin(self, queries, slot, bow_knn=0):
        budget = len(queries) * self.budget_per_query
        gathered, keys_all, ev = self._stage1(queries, slot, budget, bow_knn)
        return (queries, gathered, keys_all, slot, bow_knn, ev, budget)

    def _finish(self, state, gather_results, split=False, turn=0):
        import torch
        queries, gathered, keys_all, slot, bow_knn, ev, budget = state
        B = len(queries)
        assert gathered.dtype == torch.uint8 and tuple(gathered.shape) == (self.world, packed_bytes(B, budget))
        while True:
            # every rank holds every header: the same decision everywhere, no extra collective
            if ev is not None:
                ev.synchronize()
            hdr = gathered[:, :16].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            mx = int(hdr[:, 0].max()) if hdr.size else 0
            self._max_total = max(self._max_total, mx)
            if int(np.bitwise_or.reduce(hdr[:, 3])) & 2:
                raise OverflowError("a shard found more 2D-3D candidates for one query than a context's part holds")
            if mx <= budget:
                break
            per = -(-mx * 5 // (4 * B))                       # 25 % head room, the larger budget is kept
            self.budget_per_query = max(self.budget_per_query, (per + 63) // 64 * 64)
            budget = B * self.budget_per_query
            self._n_redo += 1
            gathered, _, ev = self._stage1(queries, slot, budget, bow_knn, keys_all)
        self._n_batches += 1
        self._n_queries += B
        mine = [i for i in range(B) if self.owner(i) == self.rank]
        if split and hasattr(self.compute, "stage2_begin"):
            # queued, not awaited: the caller collects the results later (localize_stream), so that the NEXT batch's stage 1
            # can be queued while this batch's P3P rounds run
            return ("pending", self.compute.stage2_begin(mine, gathered, slot, budget, turn), gather_results)
        local = self.compute.stage2(mine, gathered, slot, budget)
        return self._collect(local, gather_results)

    def _collect(self, local, gather_results):
        if not gather_results or self.world == 1:
            return local
        allres = [None] * self.world
        self.dist.all_gather_object(allres, local, group=self.group)
        out = {}
        for d in allres:
            out.update(d)
        return out

    def _finish_end(self, token):
        _, pending, gather_results = token
        return self._collect(self.compute.stage2_end(pending), gather_results)

    def localize_batch(self, queries, gather_results=True, bow_knn=0):
        return self._finish(self._begin(queries, 0, bow_knn), gather_results)

    def localize_stream(self, batches, gather_results=False, bow_knn=0):
        """Generator over batches (lists of queries), yielding each batch's {index: result} in order.  With a
        two-slot compute object the shard-local stage of batch i+1 (and its collectives) is already queued while batch
        i goes through its P3P stage, so the exchange and the latency-bound tail hide under the next batch's Hamming
        scans.  Every rank must iterate the same batches."""
        prev = None
        slot = 0
        # Three batches deep when the compute object can queue stage 2 without waiting for it (HipShardCompute): while the
        # host waits for batch b - 1's poses the GPU already holds stage 1 of batch b + 1 and stage 2 of batch b.  (Two deep,
        # the host sat in batch b's stage 2 for as long as its P3P rounds took and stage 1 of batch b + 2 was queued only
        # then: 17 ms per 256-query batch for a rank of 8 whose scans take 8.)
        deep = hasattr(self.compute, "stage2_begin") and os.environ.get("SFMLOC_SHARD_PIPELINE", "3") != "2"
        queued = None      # batch b: stage 2 queued, results not collected
        turn = 0
        for batch in batches:
            if self.n_slots < 2:
                yield self.localize_batch(batch, gather_results, bow_knn)
                continue
            cur = self._begin(batch, slot, bow_knn)
            if prev is not None:
                if deep:
                    token = self._finish(prev, gather_results, split=True, turn=turn)
                    turn ^= 1
                    if queued is not None:
                        yield self._finish_end(queued)
                    if isinstance(token, tuple) and token and token[0] == "pending":
                        queued = token
                    else:                   # (nothing of this batch is owned here, or the compute object did not split)
                        queued = None
                        yield token
                else:
                    yield self._finish(prev, gather_results)
            prev = cur
            slot ^= 1
        if prev is not None:
            if deep:
                token = self._finish(prev, gather_results, split=True, turn=turn)
                if queued is not None:
                    yield self._finish_end(queued)
                yield self._finish_end(token) if isinstance(token, tuple) and token and token[0] == "pending" else token
            else:
                yield self._finish(prev, gather_results)


class HipShardCompute:
    """Stage 1 / stage 2 on one MI355X through the C ABI.  `shard_map` is a capi.Map holding this rank's views
    (with the FULL landmark table); queries are capi.Query objects created on it (with a resident BoW vector when the
    shortlist is used).  Two slots of `n_contexts` contexts each, so that ShardedLocalizer.localize_stream can overlap
    consecutive batches.  Nothing here blocks the host between stage 1 and the collective: the contexts' streams and
    the collective's stream (`comm`) are ordered by events (sfmloc_context_signal / _wait)."""

    n_slots = 2

    def __init__(self, shard_map, cap=None, n_contexts=4, device=None, gang=None):
        import os
        import torch
        from . import capi
        # stage 1 of `gang` queries per launch (sfmloc_gang_begin/_end): a rank has 1/N of the scan per query but every
        # query's launches, so with more ranks the launches are what is left.  SFMLOC_GANG=1 turns it off.
        g = int(os.environ.get("SFMLOC_GANG", "0")) if gang is None else int(gang)
        self.gang = max(1, min(capi.GANG_MAX, g if g > 0 else 1))
        self.n_stage2 = 8 if self.gang == 1 else capi.GANG_MAX   # queries in stage 2 at a time
        self.map = shard_map
        self.device = torch.device("cuda", shard_map.params.device) if device is None else device
        self.comm = torch.cuda.Stream(self.device)
        self.ctxs = []
        for _ in range(self.n_slots):
            cs = []
            for k in range(n_contexts):     # a gang's first context owns the stream the whole gang works on
                cs.append(shard_map.context(share=None if k % self.gang == 0 else cs[k - k % self.gang]))
            self.ctxs.append(cs)
        # stage 2 is one chain of launches per query: it wants a stream per query in flight, which the gangs' members
        # do not have -- contexts of its own then, shared by the slots (a batch's stage 2 ends before the next one's begins)
        self.ctx2 = None
        if self.gang > 1:
            self.ctx2 = []
            for _ in range(2):
                lead = shard_map.context(merge_only=True)
                self.ctx2.append([lead] + [shard_map.context(share=lead, merge_only=True)
                                           for _ in range(self.n_stage2 - 1)])
        self._buf = {}
        self._views = {}       # slot -> the query views of its last image batch, and the buffer they point into
        self._feat_keep = {}
        self._queries = [None] * self.n_slots

    def close(self):
        for qs in self._views.values():
            for q in qs:
                q.close()
        self._views = {}
        self._feat_keep = {}
        for cs in self.ctxs:
            for c in reversed(cs):          # (a gang's members before the context whose stream they borrow)
                c.close()
        for group in self.ctx2 or []:
            for c in reversed(group):
                c.close()

    def _tensor(self, name, slot, shape, dtype):
        """persistent per (name, slot, shape): the C ABI writes every byte that is read back, so no clearing (and a
        fill on torch's stream would race with the writes on the contexts' streams)"""
        import torch
        key = (name, slot, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
            self._buf[key] = t
        return t

    def _streams(self, slot):
        """one context per stream of the slot (a gang's members work on its first context's stream)"""
        return self.ctxs[slot][::self.gang]

    def before_collective(self, slot=0):
        for c in self._streams(slot):
            c.signal(self.comm.cuda_stream)       # the collective waits for the slot's queued work

    def after_collective(self, slot=0):
        for c in self._streams(slot):
            c.wait(self.comm.cuda_stream)         # the slot's later work waits for the collective

    def _rounds(self, slot, n_queries):
        """(session, [(context, query index)]) in the order they are queued: query i goes to context i mod n as ever; the
        contexts are cut into gangs of `self.gang`, a gang takes its share of every n consecutive queries in ONE session,
        and the gangs take turns so that their streams stay equally loaded"""
        from . import capi
        cs = self.ctxs[slot]
        n = len(cs)
        for base in range(0, n_queries, n):
            for g0 in range(0, n, self.gang):
                work = [(cs[k], base + k) for k in range(g0, min(n, g0 + self.gang)) if base + k < n_queries]
                if work:
                    yield capi.gang([c for c, _ in work]), work

    def bow_keys(self, queries, knn, slot=0):
        import torch
        B = len(queries)
        keys = self._tensor("keys", slot, (B, knn), torch.int64)
        # ONE foreign call for the batch (sfmloc_shard_batch_bow_keys runs the sessions of _rounds in C)
        from . import capi
        capi.shard_batch_bow_keys(self.ctxs[slot], self.gang, queries, knn, keys.data_ptr())
        return keys

    def query_views(self, gathered, cap, bow_dim, n_queries, slot=0):
        """the batch's queries as views into the gathered feature blocks (query i: rank i mod world, block i // world);
        the buffer is kept until the slot's next batch"""
        import torch
        world, per, total = gathered.shape
        o_desc, o_kpt, o_kpt6, o_bow, tot = feature_block_layout(cap, bow_dim)
        assert tot == total and gathered.is_contiguous() and gathered.data_ptr() % 16 == 0
        hdr = gathered[:, :, :16].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        for q in self._views.get(slot, ()):
            q.close()
        self._views[slot] = []
        self._feat_keep[slot] = gathered
        base = gathered.data_ptr()
        for i in range(n_queries):
            r, j = i % world, i // world
            p = base + (r * per + j) * total
            n, w, h = (int(x) for x in hdr[r, j, :3])
            self._views[slot].append(self.map.query_view(p + o_desc, p + o_kpt, p + o_kpt6, p + o_bow if bow_dim else 0,
                                                         n, w, h))
        return list(self._views[slot])

    def _packed(self, slot, B, budget):
        """the slot's packed part with its header zeroed on the collective's stream, which the slot's contexts then
        wait for (the running total must be zero before the batch's first export)"""
        import torch
        from . import capi
        part = self._tensor("packed", slot, (capi.packed_bytes(B, budget),), torch.uint8)
        with torch.cuda.stream(self.comm):
            part[:16].zero_()
        self.after_collective(slot)
        return part

    def stage1_bow(self, queries, keys_all, knn, slot=0, budget=0):
        B = len(queries)
        world = keys_all.shape[0]
        part = self._packed(slot, B, budget)
        # query i's key lists: keys_all[r, i, :] for r in range(world) -> one part every B*knn keys; one foreign call
        from . import capi
        assert keys_all.is_contiguous() and tuple(keys_all.shape) == (world, B, knn)
        capi.shard_batch_begin_bow(self.ctxs[slot], self.gang, queries, keys_all.data_ptr(), world, knn, part.data_ptr(), budget)
        self._queries[slot] = queries
        return part

    def stage1(self, queries, slot=0, budget=0, view_sels=None):
        B = len(queries)
        part = self._packed(slot, B, budget)
        base = part.data_ptr()
        if view_sels is None:
            from . import capi
            capi.shard_batch_begin(self.ctxs[slot], self.gang, queries, base, budget)
            self._queries[slot] = queries
            return part
        for sess, work in self._rounds(slot, B):
            with sess:
                for c, i in work:
                    # K1..K3 + candidate emission, asynchronous; view_sels[i]: this shard's views to scan
                    c.shard_begin(queries[i], None if view_sels is None else view_sels[i])
                    c.shard_export_packed(base, B, budget, i)  # on the same stream
        self._queries[slot] = queries
        return part

    def stage2(self, indices, gathered, slot=0, budget=0):
        world, pb = gathered.shape
        base = gathered.data_ptr()
        out = {}
        queries = self._queries[slot]
        B = len(queries)
        if self.ctx2 is not None and len(indices) <= len(self.ctx2[0]) and len(self.ctx2) >= 2:
            return self.stage2_end(self.stage2_begin(indices, gathered, slot, budget, 0))
        if self.ctx2 is not None:
            # gang sessions: the 2D-3D selection and every P3P round of up to len(ctx2) queries per launch
            # (two groups of contexts take turns, so that a session is queued while the one before it is awaited)
            from . import capi
            n = len(self.ctx2[0])
            pending = None
            for t, k0 in enumerate(range(0, len(indices), n)):
                cs = self.ctx2[t % len(self.ctx2)]
                chunk = indices[k0:k0 + n]
                capi.merge_batch_begin(cs[:len(chunk)], [queries[i] for i in chunk], chunk, base, world, pb, B, budget)
                if pending is not None:
                    for c, i in pending:
                        out[i] = _pose_tuple(c.end())
                pending = list(zip(cs, chunk))
            for c, i in pending or []:
                out[i] = _pose_tuple(c.end())
            return out
        cs = self.ctxs[slot]
        n = len(cs)
        pending = []
        for k, i in enumerate(indices):
            c = cs[k % n]
            if k >= n:
                j, cj = pending.pop(0)
                out[j] = _pose_tuple(cj.end())
            c.merge_begin_packed(queries[i], base, world, B, budget, i, part_stride=pb)
            pending.append((i, c))
        for j, cj in pending:
            out[j] = _pose_tuple(cj.end())
        return out


def _hip_stage2_begin(self, indices, gathered, slot=0, budget=0, turn=0):
    """Stage 2 of a batch's own queries QUEUED (one gang session per group of contexts), nothing awaited: -> the list the
    caller hands to stage2_end.  `turn` picks the group of contexts (consecutive batches alternate, so a batch's session is
    queued while the previous batch's is still running).  Falls back to the blocking form when the batch's share does not
    fit one group."""
    from . import capi
    world, pb = gathered.shape
    queries = self._queries[slot]
    B = len(queries)
    if self.ctx2 is None:
        return ("done", self.stage2(indices, gathered, slot, budget))
    if not indices:
        return ("done", {})
    # a batch's share larger than a group (a rank of 2 owns 128 of a batch's 256 queries, a group holds GANG_MAX): one
    # session per group of contexts, as many groups as it takes -- made when first needed --, all queued, none awaited.
    # (Until round 4 such a share went through the blocking form: the host sat in stage 2 for 42 of a rank of 2's 47 ms
    # per batch and the next batch's stage 1 was queued only then.)  Group 2 j + (turn & 1) takes the j-th session.
    n = len(self.ctx2[0])
    need = -(-len(indices) // n)
    while len(self.ctx2) < 2 * need:
        lead = self.map.context(merge_only=True)
        self.ctx2.append([lead] + [self.map.context(share=lead, merge_only=True) for _ in range(n - 1)])
    pending = []
    for j in range(need):
        cs = self.ctx2[2 * j + (turn & 1)]
        chunk = indices[j * n:(j + 1) * n]
        capi.merge_batch_begin(cs[:len(chunk)], [queries[i] for i in chunk], chunk, gathered.data_ptr(), world, pb, B, budget)
        pending += list(zip(cs, chunk))
    return ("queued", pending, gathered)      # (the gathered buffer must outlive the sessions)


def _hip_stage2_end(self, pending):
    if pending[0] == "done":
        return pending[1]
    return {i: _pose_tuple(c.end()) for c, i in pending[1]}


HipShardCompute.stage2_begin = _hip_stage2_begin
HipShardCompute.stage2_end = _hip_stage2_end


def _pose_tuple(res):
    pose, pq, pl = res
    return {"ok": bool(pose.ok), "n_inliers": int(pose.n_inliers), "R": np.array(pose.R).reshape(3, 3),
            "center": np.array(pose.center), "P": np.array(pose.P).reshape(3, 4), "pair_qfeat": pq, "pair_landmark": pl,
            "fingerprint": _fingerprint(pose, pq, pl)}


def _fingerprint(pose, pq, pl):
    from . import capi
    return capi.result_fingerprint(pose, pq, pl, view_counts=False)
