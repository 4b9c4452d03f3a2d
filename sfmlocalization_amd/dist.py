"""Sharded localisation over torch.distributed: one process per GPU, the descriptor bank sharded
by contiguous view ranges, ONE all-gather of per-shard candidate parts per query batch (RCCL over xGMI when the
backend is "nccl"; gloo on CPU in the tests), then the 2D-3D selection + P3P of each query on one rank.

The collective layer is independent of what computes the parts: `HipShardCompute` drives the C ABI
(sfmloc_shard_begin / _export / sfmloc_merge_begin); the CPU tests plug in a stand-in built on the oracle.
"""
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


class ShardedLocalizer:
    """compute.stage1(queries[, slot]) -> torch.uint8 [B, part_bytes] on compute.device (this shard's parts, possibly
    still being written: compute.stage1_wait(slot) blocks until they are final);
    compute.stage2(indices, gathered[world, B, part_bytes][, slot]) -> {index: result} for the queries this rank owns.
    A compute object with `n_slots >= 2` lets two batches overlap (localize_stream)."""

    def __init__(self, compute, cap, rank=None, world=None, group=None, always_gather=False):
        import torch.distributed as dist
        self.dist = dist
        self.always_gather = always_gather   # run the collective even on one rank (rehearsal of the N>1 path)
        self.compute = compute
        self.cap = cap
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_slots = int(getattr(compute, "n_slots", 1))

    def owner(self, i):
        return i % self.world

    def _stage1(self, queries, slot, view_sels=None):
        kw = {} if view_sels is None else {"view_sels": view_sels}
        if self.n_slots > 1:
            return self.compute.stage1(queries, slot, **kw)
        return self.compute.stage1(queries, **kw)

    def _finish(self, queries, parts, slot, gather_results):
        import torch
        B = len(queries)
        assert parts.dtype == torch.uint8 and tuple(parts.shape) == (B, part_bytes(self.cap))
        if hasattr(self.compute, "stage1_wait"):
            self.compute.stage1_wait(slot)     # the parts are written on the compute object's own streams
        # concatenated along dim 0 (the layout every backend accepts), viewed as [world, B, part_bytes]
        flat = torch.empty((self.world * B, parts.shape[1]), dtype=torch.uint8, device=parts.device)
        if self.world > 1 or self.always_gather:
            if parts.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # rehearsal of the N>1 path without RCCL (e.g. two ranks sharing one GPU): stage through the host
                host = torch.empty(flat.shape, dtype=torch.uint8)
                self.dist.all_gather_into_tensor(host, parts.cpu().contiguous(), group=self.group)
                flat.copy_(host)
            else:
                self.dist.all_gather_into_tensor(flat, parts.contiguous(), group=self.group)
        else:
            flat.copy_(parts)
        if flat.is_cuda:
            torch.cuda.current_stream(flat.device).synchronize()   # stage 2 reads `flat` on other streams
        gathered = flat.view(self.world, B, parts.shape[1])
        mine = [i for i in range(B) if self.owner(i) == self.rank]
        if self.n_slots > 1:
            local = self.compute.stage2(mine, gathered, slot)
        else:
            local = self.compute.stage2(mine, gathered)
        if not gather_results or self.world == 1:
            return local
        allres = [None] * self.world
        self.dist.all_gather_object(allres, local, group=self.group)
        out = {}
        for d in allres:
            out.update(d)
        return out

    def bow_shortlists(self, local_map, query_bows, k):
        """Per query: the LOCAL view indices of the global k-nearest .bow vectors (exact, equal to the unsharded
        sfmloc_bow_select).  One small all-gather for the whole batch."""
        import torch
        ids = np.asarray(local_map.view_id, np.int64)
        B = len(query_bows)
        mine = np.full((B, k, 2), np.inf, np.float64)
        for b, qb in enumerate(query_bows):
            d = local_map.bow_distances(qb)
            order = _k_best(d, ids, k)
            mine[b, :len(order), 0] = d[order]
            mine[b, :len(order), 1] = ids[order]
        if self.world > 1:
            t = torch.from_numpy(mine)
            dev = getattr(self.compute, "device", torch.device("cpu"))
            t = t.to(dev)
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=dev)
            self.dist.all_gather_into_tensor(out.view(self.world * B, k, 2), t, group=self.group)
            allp = out.cpu().numpy()
        else:
            allp = mine[None]
        sels = []
        for b in range(B):
            p = allp[:, b].reshape(-1, 2)
            p = p[np.isfinite(p[:, 0])]
            best = p[np.lexsort((p[:, 1], p[:, 0]))[:k]]
            sels.append(np.nonzero(np.isin(ids, best[:, 1].astype(np.int64)))[0].astype(np.uint32))
        return sels

    def localize_batch(self, queries, gather_results=True, view_sels=None):
        return self._finish(queries, self._stage1(queries, 0, view_sels), 0, gather_results)

    def localize_stream(self, batches, gather_results=False, view_sels=None):
        """Generator over batches (lists of queries), yielding each batch's {index: result} in order.  With a
        two-slot compute object the shard-local stage of batch i+1 is already queued on the GPU while batch i goes
        through the collective and its P3P stage, so the exchange and the latency-bound tail hide under the next
        batch's Hamming scans.  Every rank must iterate the same batches.  view_sels: optional iterable, per batch the
        list of this shard's view selections (e.g. from bow_shortlists), consumed lazily batch by batch."""
        prev = None
        slot = 0
        sels_it = iter(view_sels) if view_sels is not None else None   # per batch: this shard's view selections
        for batch in batches:
            sels = next(sels_it) if sels_it is not None else None
            if self.n_slots < 2:
                yield self.localize_batch(batch, gather_results, view_sels=sels)
                continue
            cur = (batch, self._stage1(batch, slot, sels), slot)
            if prev is not None:
                yield self._finish(*prev, gather_results)
            prev = cur
            slot ^= 1
        if prev is not None:
            yield self._finish(*prev, gather_results)


class HipShardCompute:
    """Stage 1 / stage 2 on one MI355X through the C ABI.  `shard_map` is a capi.Map holding this rank's views
    (with the FULL landmark table); queries are capi.Query objects created on it.  Two slots of `n_contexts`
    contexts each, so that ShardedLocalizer.localize_stream can overlap consecutive batches."""

    n_slots = 2

    def __init__(self, shard_map, cap, n_contexts=4, device=None):
        import torch
        self.map = shard_map
        self.cap = cap
        self.device = torch.device("cuda", shard_map.params.device) if device is None else device
        self.ctxs = [[shard_map.context() for _ in range(n_contexts)] for _ in range(self.n_slots)]
        self._parts = [None] * self.n_slots
        self._queries = [None] * self.n_slots

    def close(self):
        for cs in self.ctxs:
            for c in cs:
                c.close()

    def stage1(self, queries, slot=0, view_sels=None):
        import torch
        B = len(queries)
        pb = part_bytes(self.cap)
        parts = self._parts[slot]
        if parts is None or parts.shape[0] != B:
            # persistent per slot; sfmloc_shard_export writes every byte of a part, so no clearing is needed (and a
            # fill on torch's stream would race with the exports on the contexts' streams)
            parts = torch.empty((B, pb), dtype=torch.uint8, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
            self._parts[slot] = parts
        base = parts.data_ptr()
        cs = self.ctxs[slot]
        for i, q in enumerate(queries):
            c = cs[i % len(cs)]
            # K1..K3 + candidate emission, asynchronous; view_sels[i]: this shard's views to scan (BoW shortlist)
            c.shard_begin(q, None if view_sels is None else view_sels[i])
            c.shard_export(base + i * pb, self.cap)  # device-to-device copy on the same stream
        self._queries[slot] = queries
        return parts

    def stage1_wait(self, slot=0):
        for c in self.ctxs[slot]:
            c.sync()                               # the collective runs on torch's stream

    def stage2(self, indices, gathered, slot=0):
        world, B, pb = gathered.shape
        base = gathered.data_ptr()
        out = {}
        cs = self.ctxs[slot]
        n = len(cs)
        queries = self._queries[slot]
        pending = []
        for k, i in enumerate(indices):
            c = cs[k % n]
            if k >= n:
                j, cj = pending.pop(0)
                out[j] = _pose_tuple(cj.end())
            # query i's parts: gathered[r, i, :] for r in range(world) -> stride B*pb
            c.merge_begin(queries[i], base + i * pb, world, self.cap, part_stride=B * pb)
            pending.append((i, c))
        for j, cj in pending:
            out[j] = _pose_tuple(cj.end())
        return out


def _pose_tuple(res):
    pose, pq, pl = res
    return {"ok": bool(pose.ok), "n_inliers": int(pose.n_inliers), "R": np.array(pose.R).reshape(3, 3),
            "center": np.array(pose.center), "P": np.array(pose.P).reshape(3, 4), "pair_qfeat": pq, "pair_landmark": pl}
