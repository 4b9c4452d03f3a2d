"""CPU, world_size 2 over gloo: the sharded-localisation protocol of sfmlocalization_amd/dist.py (view sharding,
part layout, one all-gather per batch, query ownership, result gather) with the oracle standing in for the HIP
stages.  The merged result must equal the unsharded oracle pipeline's exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sfmlocalization_amd import dist as D
import synthdata as synth


def test_shard_views_balanced_and_contiguous():
    off = np.concatenate([[0], np.cumsum([5, 0, 100, 100, 3, 50, 50, 200, 1, 1])])
    for world in (1, 2, 3, 4, 8):
        r = D.shard_views(off, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == 10
        assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(a <= b for a, b in r)
    assert D.shard_views(off, 2) == [(0, 6), (6, 10)]


def test_packed_batch_layout_roundtrip():
    """The packed exchange buffer (one per shard and batch): counts / offsets / candidates; a budget that does not
    suffice leaves the query empty, sets the flag and keeps counting the total."""
    rng = np.random.Generator(np.random.PCG64(3))
    lists = []
    for n in (3, 0, 5, 1):
        c = np.zeros(n, D.CANDIDATE_DTYPE)
        c["order"] = rng.integers(0, 1 << 60, n)
        c["qfeat"] = rng.integers(0, 2000, n)
        c["X"] = rng.normal(size=(n, 3))
        lists.append(c)
    buf = D.pack_batch(lists, 16)
    assert len(buf) == D.packed_bytes(4, 16) == 48 + 16 * 40 and buf[:16].view(np.uint32).tolist() == [9, 4, 16, 0]
    for i, c in enumerate(lists):
        assert (D.unpack_batch(buf, i) == c).all()
    small = D.pack_batch(lists, 7)
    assert small[:16].view(np.uint32).tolist() == [9, 4, 7, 1]
    assert (D.unpack_batch(small, 0) == lists[0]).all() and len(D.unpack_batch(small, 2)) == 0
    assert (D.unpack_batch(small, 3) == lists[3])[:0].all()


def test_part_layout_roundtrip():
    c = np.zeros(3, D.CANDIDATE_DTYPE)
    c["order"] = [D.order_key(12, 7, 3), D.order_key(0, 0xFFFFFF, 0), D.order_key(512, 1, 2)]
    c["qfeat"] = [5, 6, 7]
    c["landmark_id"] = [100, 200, 300]
    c["X"] = np.arange(9).reshape(3, 3)
    assert D.CANDIDATE_DTYPE.itemsize == 40 and D.part_bytes(10) == 16 + 400
    buf = D.pack_part(c, 10)
    back = D.unpack_part(buf, 10)
    assert (back == c).all()
    assert int(c["order"][0]) == (12 << 48) | (7 << 24) | 3
    with pytest.raises(OverflowError):
        D.unpack_part(D.pack_part(c, 2), 2)          # header keeps the true count -> overflow is detected


def test_merge_bow_shortlists_equals_global_selection():
    """Sharded BoW shortlist: per-shard k best (distance, view id) pairs merged to the global k best, ties to the lower
    view id -- equal to ranking all views at once, for every split of the views into shards."""
    rng = np.random.Generator(np.random.PCG64(6))
    V, k = 57, 9
    dist = rng.integers(0, 12, V).astype(np.float32)            # many exact ties
    ids = np.arange(V) * 3 + 1
    glob = set(int(v) for v in ids[np.lexsort((ids, dist))[:k]])
    for world in (1, 2, 3, 8):
        cuts = [0] + sorted(rng.choice(np.arange(1, V), world - 1, replace=False).tolist()) + [V] if world > 1 else [0, V]
        per_rank = []
        for r in range(world):
            a, b = cuts[r], cuts[r + 1]
            order = np.lexsort((ids[a:b], dist[a:b]))[:k]
            m = np.full((k, 2), np.inf)
            m[:len(order), 0], m[:len(order), 1] = dist[a:b][order], ids[a:b][order]
            per_rank.append(m)
        gathered = np.stack(per_rank)
        got = set()
        for r in range(world):
            a, b = cuts[r], cuts[r + 1]
            sel = D.merge_bow_shortlists(dist[a:b], ids[a:b], k, world, lambda mine: gathered)
            assert list(sel) == sorted(sel)
            got |= set(int(ids[a + i]) for i in sel)
        assert got == glob


def test_gang_sessions_cover_every_query_once_on_its_context():
    """HipShardCompute._rounds (host logic only): query i stays on context i mod n, a session holds distinct contexts of
    ONE gang, and consecutive sessions alternate between the gangs."""
    class FakeCtx:
        _h = None

    for n_ctx, gang, n_q in ((8, 1, 13), (8, 4, 13), (32, 16, 256), (6, 4, 5), (30, 15, 31)):
        comp = object.__new__(D.HipShardCompute)
        comp.ctxs = [[FakeCtx() for _ in range(n_ctx)]]
        comp.gang = gang
        seen = {}
        last_gang = None
        for sess, work in comp._rounds(0, n_q):
            cs = [c for c, _ in work]
            assert 1 <= len(cs) <= gang and len(set(map(id, cs))) == len(cs) and sess.ctxs == cs
            gangs = {comp.ctxs[0].index(c) // gang for c in cs}
            assert len(gangs) == 1
            if n_ctx // gang > 1 and n_q > n_ctx:
                assert gangs != last_gang or len(cs) < gang       # the gangs take turns
            last_gang = gangs
            for c, i in work:
                assert i not in seen and comp.ctxs[0].index(c) == i % n_ctx
                seen[i] = c
        assert sorted(seen) == list(range(n_q))


def test_feature_block_roundtrip_and_capacity():
    rng = np.random.Generator(np.random.PCG64(3))
    for n, cap, bow_dim in ((0, 64, 0), (1, 64, 4), (64, 64, 500), (200, 256, 500)):
        desc = rng.integers(0, 256, (n, 64), dtype=np.uint8)
        kpt = rng.uniform(0, 2000, (n, 2)).astype(np.float32)
        k6 = kpt + 0.5
        bow = rng.random(bow_dim).astype(np.float32)
        b = D.pack_features(desc, kpt, k6, 1920, 1080, bow, cap, bow_dim)
        o_desc, o_kpt, o_kpt6, o_bow, total = D.feature_block_layout(cap, bow_dim)
        assert b.shape == (total,) and total % 16 == 0 and o_desc % 16 == 0 and o_kpt % 8 == 0 and o_bow % 4 == 0
        assert not b[o_desc + n * 64:o_kpt].any()                  # the rows beyond n are zero (the kernels rely on it)
        d2, k2, k62, w, h, bow2 = D.unpack_features(b, cap, bow_dim)
        assert np.array_equal(d2, desc) and np.array_equal(k2, kpt) and np.array_equal(k62, k6)
        assert (w, h) == (1920, 1080) and np.array_equal(bow2, bow)
    with pytest.raises(OverflowError):
        D.pack_features(np.zeros((65, 64), np.uint8), np.zeros((65, 2)), np.zeros((65, 2)), 1, 1, None, 64, 0)


def test_k_best_equals_full_sort():
    """dist._k_best (partition + tie handling) against the full (distance, id) sort it replaces."""
    rng = np.random.Generator(np.random.PCG64(16))
    for n in (1, 2, 9, 100, 1000):
        d = rng.integers(0, max(2, n // 7), n).astype(np.float32)
        ids = np.arange(n, dtype=np.int64) * 2 + 5
        for k in sorted({1, 2, n // 3 + 1, n - 1 if n > 1 else 1, n, n + 3}):
            np.testing.assert_array_equal(D._k_best(d, ids, k), np.lexsort((ids, d))[:k])


class OracleShardCompute:
    """stand-in for HipShardCompute built on the oracle (tests only)"""

    def __init__(self, m, v0, v1, cap):
        self.m, self.v0, self.v1, self.cap = m, v0, v1, cap
        self.device = torch.device("cpu")

    n_slots = 2      # so that ShardedLocalizer.localize_stream takes its pipelined branch

    def _parts(self, queries, slot, cap, sels=None):
        from oracle import pipeline as opipe
        if not hasattr(self, "queries"):
            self.queries = {}
        self.queries[slot] = queries
        cands = []
        for i, q in enumerate(queries):
            kw = {} if sels is None else {"view_sel": sels[i]}
            cands.append(D.reduce_candidates(
                opipe.shard_candidates(self.m, q.desc, q.kpt_xy, (q.width, q.height), self.v0, self.v1, **kw)))
        return torch.from_numpy(D.pack_batch(cands, cap))

    def stage1(self, queries, slot=0, budget=0):
        return self._parts(queries, slot, budget)

    def bow_keys(self, queries, knn, slot=0):
        """this shard's knn best (float32 L2 distance, view id) keys per query, padded with ~0"""
        ids = self.m.view_id[self.v0:self.v1]
        out = np.full((len(queries), knn), D.BOW_KEY_PAD, np.uint64)
        for i, q in enumerate(queries):
            d = ((self.m.extra["bow"][self.v0:self.v1] - q.bow[None, :]) ** 2).sum(1, dtype=np.float32)
            keys = np.sort(D.bow_key(d, ids))[:knn]
            out[i, :len(keys)] = keys
        return torch.from_numpy(out.view(np.int64))

    def stage1_bow(self, queries, keys_all, knn, slot=0, budget=0):
        ka = keys_all.numpy().view(np.uint64)
        ids = self.m.view_id[self.v0:self.v1]
        sels = [self.v0 + D.select_from_keys(ka[:, i, :], knn, ids) for i in range(len(queries))]
        return self._parts(queries, slot, budget, sels)

    def query_views(self, gathered, cap, bow_dim, n_queries, slot=0):
        """images in: the batch's queries out of the gathered feature blocks (host copies here; HipShardCompute makes
        views into the device buffer)"""
        import types
        g = gathered.numpy()
        world = g.shape[0]
        out = []
        for i in range(n_queries):
            desc, kpt, kpt6, w, h, bow = D.unpack_features(g[i % world, i // world], cap, bow_dim)
            out.append(types.SimpleNamespace(desc=desc, kpt_xy=kpt, kpt6_xy=kpt6, width=w, height=h, bow=bow))
        return out

    def stage2(self, indices, gathered, slot=0, budget=0):
        from oracle import pipeline as opipe
        g = gathered.numpy()
        out = {}
        for i in indices:
            parts = [D.unpack_batch(g[r], i) for r in range(g.shape[0])]
            out[i] = opipe.merge_candidates(parts, self.queries[slot][i].kpt_xy, self.m.intrinsic)
        return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_c
        oracle_c.build()
        m = synth.make_map(31, n_views=24, desc_per_view=220, views_per_place=8, landmarks_per_place=200,
                           obs_per_view=90, ragged=True)
        queries = [synth.make_query(m, 700 + k, n_feat=260, n_copies=110, outlier_frac=0.2, place=k % 3) for k in range(3)]
        v0, v1 = D.shard_views(m.view_off, world)[rank]
        loc = D.ShardedLocalizer(OracleShardCompute(m, v0, v1, cap=2048), budget_per_query=2048)
        res = loc.localize_batch(queries)
        assert loc.counters()["batches_exchanged_again_with_a_larger_budget"] == 0
        # a budget too small for some shard: every rank sees that shard's total in the gathered headers and the batch is
        # exchanged again with a budget that fits -- same results
        small = D.ShardedLocalizer(OracleShardCompute(m, v0, v1, cap=2048), budget_per_query=4)
        res_small = small.localize_batch(queries)
        assert small.counters()["batches_exchanged_again_with_a_larger_budget"] == 1 and small.budget_per_query > 4
        for k in res:
            assert res_small[k]["ok"] == res[k]["ok"] and np.array_equal(res_small[k]["ms_qfeat"], res[k]["ms_qfeat"])
        # sharded BoW shortlist (two collectives per batch): equal to shortlisting over the whole map
        rng = np.random.Generator(np.random.PCG64(77))
        m.extra["bow"] = rng.integers(0, 4, (m.n_views, 12)).astype(np.float32)     # many tied distances
        knn = 7
        for k, qq in enumerate(queries):
            qq.bow = m.extra["bow"][(5 * k) % m.n_views] + rng.integers(0, 2, 12).astype(np.float32)
        bowloc = D.ShardedLocalizer(OracleShardCompute(m, v0, v1, cap=2048), n_views_global=m.n_views)
        res_bow = bowloc.localize_batch(queries, bow_knn=knn)
        assert bowloc.counters()["bow_key_allgather_bytes_per_batch_per_rank"] == len(queries) * knn * 8
        # the pipelined form over three uneven batches must give the same per-query results
        batches = [[queries[0], queries[1]], [queries[2]], [queries[2], queries[0]]]
        for b, out in zip(batches, loc.localize_stream(batches, gather_results=True)):
            for i, qq in enumerate(b):
                k = [id(x) for x in queries].index(id(qq))
                assert out[i]["ok"] == res[k]["ok"], "stream vs batch: ok"
                if out[i]["ok"]:
                    assert np.array_equal(np.asarray(out[i]["pair_qfeat"]), np.asarray(res[k]["pair_qfeat"]))
                    assert np.array_equal(np.asarray(out[i]["P"]), np.asarray(res[k]["P"]))
        # images in: every rank "extracts" only the queries it owns, one all-gather of feature blocks hands every rank
        # the whole batch, and the batch localises as before
        from sfmlocalization_amd import capi
        cap, bow_dim = 320, 12
        own = {i: D.pack_features(qq.desc, qq.kpt_xy, capi.feat_round_trip(qq.kpt_xy), qq.width, qq.height, qq.bow, cap,
                                  bow_dim)
               for i, qq in enumerate(queries) if i % world == rank}
        imgloc = D.ShardedLocalizer(OracleShardCompute(m, v0, v1, cap=2048), budget_per_query=2048)
        got_q = imgloc.gather_queries(own, len(queries), cap, bow_dim)
        for a, b in zip(got_q, queries):
            assert np.array_equal(a.desc, b.desc) and np.array_equal(a.kpt_xy, b.kpt_xy.astype(np.float32))
            assert np.array_equal(a.kpt6_xy, capi.feat_round_trip(b.kpt_xy)) and (a.width, a.height) == (b.width, b.height)
            assert np.array_equal(a.bow, b.bow)
        res_img = imgloc.localize_batch(got_q)
        assert (imgloc.counters()["feature_allgather_bytes_per_batch_per_rank"]
                == -(-len(queries) // world) * D.feature_block_layout(cap, bow_dim)[4])
        for k in res:
            assert res_img[k]["ok"] == res[k]["ok"] and np.array_equal(res_img[k]["ms_qfeat"], res[k]["ms_qfeat"])
        if rank == 0:
            pack = lambda rr: {i: {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in r.items()}  # noqa: E731
                               for i, r in rr.items()}
            q.put({"plain": pack(res), "bow": pack(res_bow),
                   "qbow": [qq.bow.tolist() for qq in queries], "bow_mat": m.extra["bow"].tolist(), "knn": knn})
    except Exception as e:  # surface the failure instead of letting the parent time out
        import traceback
        q.put({"error": f"rank {rank}: {e}\n{traceback.format_exc()}"})
        raise
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_unsharded():
    from oracle import pipeline as opipe
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    assert "error" not in res, res.get("error")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = synth.make_map(31, n_views=24, desc_per_view=220, views_per_place=8, landmarks_per_place=200,
                       obs_per_view=90, ragged=True)
    n_ok = 0
    bow_mat = np.array(res["bow_mat"], np.float32)
    for k in range(3):
        qq = synth.make_query(m, 700 + k, n_feat=260, n_copies=110, outlier_frac=0.2, place=k % 3)
        # the sharded shortlist + path equals the shortlist over the whole map + the unsharded path
        d = ((bow_mat - np.array(res["qbow"][k], np.float32)[None, :]) ** 2).sum(1, dtype=np.float32)
        sel = np.sort(np.lexsort((m.view_id, d))[:res["knn"]]).astype(np.uint32)
        exp_b = opipe.localize(m, qq.desc, qq.kpt_xy, (qq.width, qq.height), view_sel=sel)
        got_b = res["bow"][str(k)] if str(k) in res["bow"] else res["bow"][k]
        assert got_b["ok"] == exp_b["ok"]
        np.testing.assert_array_equal(got_b.get("ms_qfeat", []), exp_b["ms_qfeat"])
        exp = opipe.localize(m, qq.desc, qq.kpt_xy, (qq.width, qq.height))
        got = res["plain"][k]
        assert got["ok"] == exp["ok"]
        np.testing.assert_array_equal(got["ms_qfeat"], exp["ms_qfeat"])
        np.testing.assert_array_equal(got["ms_landmark"], exp["ms_landmark"])
        if exp["ok"]:
            n_ok += 1
            np.testing.assert_array_equal(got["pair_qfeat"], exp["pair_qfeat"])
            np.testing.assert_array_equal(got["pair_landmark"], exp["pair_landmark"])
            assert (np.array(got["P"]) == exp["P"]).all() and (np.array(got["center"]) == exp["center"]).all()
    assert n_ok >= 2


_TIMEOUT_RANK = r'''
import os, sys, time
from datetime import timedelta
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
import bench
rank = int(sys.argv[1])
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[2]
a = bench.parse(["--gpus", "2", "--steps", "1", "--warmup", "0"])
dist.init_process_group("gloo", rank=rank, world_size=2, timeout=timedelta(seconds=4))
x = torch.zeros(4)
out = torch.zeros(8)
dist.all_gather_into_tensor(out, x)          # both ranks: the group works
if rank == 1:
    time.sleep(30)                           # ... then this rank stops taking part
    os._exit(0)
try:
    dist.all_gather_into_tensor(out, x)      # cannot complete: raises within the group's timeout
except BaseException as e:
    bench.fail(a, rank, f"{{type(e).__name__}}: {{e}}", 3)
print("unreachable")
'''


def test_a_collective_that_cannot_complete_ends_the_run_with_a_line():
    """bench.py at N > 1: the process group is created with a timeout, and a rank whose collective
    cannot complete -- here rank 1 stops taking part -- prints ONE JSON line with the reason (`error`, `value` null) and
    exits non-zero at once instead of sitting out the default ten minutes in silence.  World 2 over gloo; the same
    bench.fail() serves the watchdog of a run that hangs inside a call."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = _TIMEOUT_RANK.format(root=root)
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    out0, err0 = procs[0].communicate(timeout=120)
    took = time.perf_counter() - t0
    procs[1].kill()
    procs[1].communicate()
    assert procs[0].returncode == 3, (procs[0].returncode, out0[-500:], err0[-1500:])
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "unreachable" not in out0
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["error"] and d["metric"] == "query images localized/sec"
    assert took < 60, took


def test_watchdog_ends_a_run_that_makes_no_progress():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys, time; sys.path.insert(0, {root!r}); import bench; a = bench.parse(['--gpus', '2']); "
            "w = bench.Watchdog(1.0, 0, a); time.sleep(20); print('unreachable')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4 and '"error": "watchdog' in r.stdout and "unreachable" not in r.stdout
