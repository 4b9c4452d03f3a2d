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


class ShardedLocalizer:
    """compute.stage1(queries) -> torch.uint8 [B, part_bytes] on compute.device (this shard's parts);
    compute.stage2(indices, gathered[world, B, part_bytes]) -> {index: result} for the queries this rank owns."""

    def __init__(self, compute, cap, rank=None, world=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.compute = compute
        self.cap = cap
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world

    def owner(self, i):
        return i % self.world

    def localize_batch(self, queries, gather_results=True):
        import torch
        B = len(queries)
        parts = self.compute.stage1(queries)
        assert parts.dtype == torch.uint8 and tuple(parts.shape) == (B, part_bytes(self.cap))
        # concatenated along dim 0 (the layout every backend accepts), viewed as [world, B, part_bytes]
        flat = torch.empty((self.world * B, parts.shape[1]), dtype=torch.uint8, device=parts.device)
        if self.world > 1:
            self.dist.all_gather_into_tensor(flat, parts.contiguous(), group=self.group)
        else:
            flat.copy_(parts)
        gathered = flat.view(self.world, B, parts.shape[1])
        mine = [i for i in range(B) if self.owner(i) == self.rank]
        local = self.compute.stage2(mine, gathered)
        if not gather_results or self.world == 1:
            return local
        allres = [None] * self.world
        self.dist.all_gather_object(allres, local, group=self.group)
        out = {}
        for d in allres:
            out.update(d)
        return out


class HipShardCompute:
    """Stage 1 / stage 2 on one MI355X through the C ABI.  `shard_map` is a capi.Map holding this rank's views
    (with the FULL landmark table); queries are capi.Query objects created on it."""

    def __init__(self, shard_map, cap, n_contexts=4, device=None):
        import torch
        self.map = shard_map
        self.cap = cap
        self.device = torch.device("cuda", shard_map.params.device) if device is None else device
        self.ctxs = [shard_map.context() for _ in range(n_contexts)]

    def close(self):
        for c in self.ctxs:
            c.close()

    def stage1(self, queries):
        import torch
        B = len(queries)
        pb = part_bytes(self.cap)
        parts = torch.zeros((B, pb), dtype=torch.uint8, device=self.device)
        base = parts.data_ptr()
        for i, q in enumerate(queries):
            c = self.ctxs[i % len(self.ctxs)]
            c.shard_begin(q)                       # K1..K3 + candidate emission, asynchronous
            c.shard_export(base + i * pb, self.cap)  # device-to-device copy on the same stream
        for c in self.ctxs:
            c.sync()                               # the collective runs on torch's stream
        self._queries = queries
        return parts

    def stage2(self, indices, gathered):
        world, B, pb = gathered.shape
        base = gathered.data_ptr()
        out = {}
        n = len(self.ctxs)
        pending = []
        for k, i in enumerate(indices):
            c = self.ctxs[k % n]
            if k >= n:
                j, cj = pending.pop(0)
                out[j] = _pose_tuple(cj.end())
            # query i's parts: gathered[r, i, :] for r in range(world) -> stride B*pb
            c.merge_begin(self._queries[i], base + i * pb, world, self.cap, part_stride=B * pb)
            pending.append((i, c))
        for j, cj in pending:
            out[j] = _pose_tuple(cj.end())
        return out


def _pose_tuple(res):
    pose, pq, pl = res
    return {"ok": bool(pose.ok), "n_inliers": int(pose.n_inliers), "R": np.array(pose.R).reshape(3, 3),
            "center": np.array(pose.center), "P": np.array(pose.P).reshape(3, 4), "pair_qfeat": pq, "pair_landmark": pl}
