#!/usr/bin/env python3
"""What ONE rank of an N-rank run does per batch, measured on one GPU: the builder has a single device, and the N-rank
throughput of the sharded path (dist.py) is, up to the collectives' latency, the rate at which a rank gets through its
batches -- every rank runs stage 1 of every query on its 1/N of the map and stage 2 of every N-th query.

The other ranks' contributions to the two all-gathers (their k-best BoW keys, their packed candidate parts) are computed
ONCE, before the timed region, by running their shards on this same GPU; the timed region then runs rank 0 for real and
`_all_gather` hands it [own live part, the others' stored parts].  The poses of rank 0's queries are compared with the
unsharded path on the whole map first (same inliers, same pose bits), so the emulated exchange is the real one.

Diagnostic: not the metric, no collective inside (its latency hides under the two-slot pipeline or does not -- only a
real N-GPU run tells).  usage: python tools/rank_emulation.py --of 8 [--steps 8 --warmup 2 --in-flight 8]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--of", type=int, default=8, help="N: the emulated world size (this process is rank 0 of N)")
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--views", type=int, default=10000)
    ap.add_argument("--desc-per-view", type=int, default=2000)
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--bow-knn", type=int, default=100)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--in-flight", type=int, default=8, help="contexts per slot (the sharded path keeps two slots)")
    ap.add_argument("--gang", type=int, default=0, help="queries per launch in stage 1 (0 = the library's default)")
    ap.add_argument("--only-stage1", action="store_true", help="diagnosis: no query is owned here (stage 2 never runs)")
    ap.add_argument("--images", choices=["", "vga", "1080p"], default="",
                    help="images in: rank 0 extracts AKAZE + M-LDB features of one synthetic frame per query it owns "
                         "(cost only: the map is synthetic, so the localised features stay the synthetic query's), packs "
                         "them, and the batch's queries are views into the gathered feature blocks (dist.gather_queries)")
    ap.add_argument("--extract-workers", type=int, default=4, help="host threads (one extractor and stream each)")
    a = ap.parse_args()
    N, B = a.of, a.batch
    # (hardware queues as bench.plan() sets them for a sharded run with gang sessions: 16 -- with 24 an emulated rank of
    # 8 drops from 18.4 k to 16.1 k queries/s, with 8 a rank of 2 from 6.0 k to 5.5 k)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if a.gang > 1 else str(min(24, max(8, 2 * a.in_flight + 2))))
    import numpy as np
    import torch
    import bench
    import sfmlocalization_amd as S
    import synthdata as synth
    from sfmlocalization_amd import dist as D

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    m = synth.make_map(2, n_views=a.views, desc_per_view=a.desc_per_view)
    queries = [synth.make_query(m, 1000 + i, n_feat=a.nq) for i in range(a.queries)]
    bow, qbow = bench.synth_bow(m, queries)
    params = S.default_params(device=0, profile=0, ransac_round=25)
    batch_ids = [i % len(queries) for i in range(B)]

    def shard(r, n):
        v0, v1 = (a.views * r) // n, (a.views * (r + 1)) // n
        r0, r1 = int(m.view_off[v0]), int(m.view_off[v1])
        dm = S.Map(m.view_id[v0:v1], m.view_off[v0:v1 + 1] - m.view_off[v0], m.desc[r0:r1], params=params,
                   view_wh=m.view_wh[v0:v1], kpt_xy=m.kpt_xy[r0:r1], row_landmark=m.row_landmark[r0:r1],
                   landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic, bow=bow[v0:v1])
        dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in queries]
        for dq, qb in zip(dqs, qbow):
            dq.set_bow(qb)
        return dm, dqs

    budget = B * 256
    # ---- the other ranks' halves of the two exchanges, once --------------------------------------------------------
    maps = [shard(r, N) for r in range(N)]
    comps = [D.HipShardCompute(dm, n_contexts=(a.in_flight if r == 0 else 2), device=dev, gang=(a.gang or None) if r == 0 else 1)
             for r, (dm, _) in enumerate(maps)]
    keys = []
    for comp, (dm, dqs) in zip(comps, maps):
        k = comp.bow_keys([dqs[i] for i in batch_ids], a.bow_knn, 0)
        for c in comp.ctxs[0]:
            c.sync()
        keys.append(k.clone())
    keys_all = torch.stack(keys)                                   # [N, B, knn]
    parts = [None] * N
    for r in range(1, N):
        dm, dqs = maps[r]
        p = comps[r].stage1_bow([dqs[i] for i in batch_ids], keys_all, a.bow_knn, 0, budget)
        for c in comps[r].ctxs[0]:
            c.sync()
        torch.cuda.synchronize()
        parts[r] = p.clone()
    for r in range(1, N):
        comps[r].close()
        for dq in maps[r][1]:
            dq.close()
        maps[r][0].close()
    dm0, dqs0 = maps[0]
    comp = comps[0]

    class Emulated(D.ShardedLocalizer):
        """rank 0 of N; the collectives replaced by device copies of what the other ranks would have sent"""

        def _all_gather(self, send, what="candidates"):
            stored = keys if send.dtype == torch.int64 else (parts if send.dim() == 1 else feats)
            out = torch.empty((N,) + tuple(send.shape), dtype=send.dtype, device=send.device)
            comm = self._comm_stream()
            with torch.cuda.stream(comm):
                out[0].copy_(send)
                for r in range(1, N):
                    assert stored[r].shape == send.shape, "the stored parts were made for another batch / budget"
                    out[r].copy_(stored[r])
                ev = comm.record_event()
            return out, ev

    # ---- images in: the other ranks' feature blocks, once; extractors for this rank's share --------------------------
    feats = [None] * N
    extract = None
    if a.images:
        from concurrent.futures import ThreadPoolExecutor
        from sfmlocalization_amd import capi
        hw = (480, 640) if a.images == "vga" else (1080, 1920)
        cap_f, bow_dim = (a.nq + 63) // 64 * 64, bow.shape[1]
        per = -(-B // N)
        block = D.feature_block_layout(cap_f, bow_dim)[4]
        packed = {}

        def pack(i):
            k = batch_ids[i]
            if k not in packed:
                q = queries[k]
                packed[k] = D.pack_features(q.desc, q.kpt_xy, capi.feat_round_trip(q.kpt_xy), q.width, q.height, qbow[k],
                                            cap_f, bow_dim)
            return packed[k]
        for r in range(1, N):
            blk = np.zeros((per, block), np.uint8)
            for i in range(r, B, N):
                blk[i // N] = pack(i)
            feats[r] = torch.from_numpy(blk).to(dev)
        frames = [synth.texture_image(900 + k, hw[0], hw[1]) for k in range(4)]
        G_ex = 8                                            # frames per extraction call (detect_and_compute_batch)
        extractors = [[S.Akaze(hw[1], hw[0], device=0) for _ in range(G_ex)] for _ in range(a.extract_workers)]
        lenders = [maps[0][0].context(merge_only=True) for _ in range(a.extract_workers)]
        for es, c in zip(extractors, lenders):              # a worker's extractors on ONE stream (a small context lends it)
            for e in es:
                e.share_stream(c)
        pool = ThreadPoolExecutor(a.extract_workers)
        mine_i = list(range(0, B, N))
        n_kp = [0]

        def extract():
            """this rank's share of a batch: one frame through AKAZE + M-LDB per owned query (W workers), then the blocks"""
            def work(w):
                js = list(range(w, len(mine_i), a.extract_workers))
                for k0 in range(0, len(js), G_ex):
                    part = js[k0:k0 + G_ex]
                    got = S.Akaze.detect_and_compute_batch(extractors[w][:len(part)], [frames[j % len(frames)] for j in part])
                    n_kp[0] = len(got[0][0])
            list(pool.map(work, range(a.extract_workers)))
            return {i: pack(i) for i in mine_i}

    loc = Emulated(comp, rank=0, world=N, n_views_global=a.views)
    host = {}                                   # host seconds inside each of the compute object's calls

    def timed(name):
        f = getattr(comp, name)

        def g(*x, **k):
            t = time.perf_counter()
            r = f(*x, **k)
            host[name] = host.get(name, 0.0) + time.perf_counter() - t
            return r
        setattr(comp, name, g)
    for name in ("bow_keys", "stage1_bow", "stage2", "stage2_begin", "stage2_end"):
        if hasattr(comp, name):
            timed(name)
    batch = [dqs0[i] for i in batch_ids]

    # ---- the emulated exchange is the real one: rank 0's queries against the unsharded path --------------------------
    res = loc.localize_batch(batch, gather_results=False, bow_knn=a.bow_knn)
    assert loc.counters()["batches_exchanged_again_with_a_larger_budget"] == 0, "budget grew: stored parts are stale"
    full, fq = shard(0, 1)
    same = 0
    checked = sorted(res)[:16]
    for i in checked:
        pose, pq, pl = full.localize_bow(fq[batch_ids[i]], qbow[batch_ids[i]], a.bow_knn)
        r = res[i]
        same += int(bool(pose.ok) == r["ok"] and int(pose.n_inliers) == r["n_inliers"]
                    and np.array_equal(np.array(pose.P).reshape(3, 4), r["P"]))
    for dq in fq:
        dq.close()
    full.close()
    n_ok_check = sum(int(r["ok"]) for r in res.values())

    # ---- the rate ---------------------------------------------------------------------------------------------------
    t_extract = [0.0]

    def batches(n):
        """descriptors in: the same resident queries every time; images in: extraction, exchange and views per batch
        (slots alternate as in localize_stream)"""
        for b in range(n):
            if extract is None:
                yield batch
                continue
            t = time.perf_counter()
            own = extract()
            t_extract[0] += time.perf_counter() - t
            yield loc.gather_queries(own, B, cap_f, bow_dim, slot=b % 2)

    def run(n):
        ok = 0
        for r in loc.localize_stream(batches(n), gather_results=False, bow_knn=a.bow_knn):
            ok += sum(int(x["ok"]) for x in r.values())
        torch.cuda.synchronize()
        return ok

    if a.only_stage1:
        loc.owner = lambda i: -1
    run(a.warmup)
    host.clear()
    t_extract[0] = 0.0
    t0 = time.perf_counter()
    ok = run(a.steps)
    dt = time.perf_counter() - t0
    print(json.dumps({
        "emulated": f"rank 0 of {N}", "map_views": a.views, "map_rows": int(m.n_rows), "shard_rows": int(m.view_off[a.views // N]),
        "bow_knn": a.bow_knn, "batch": B, "steps": a.steps, "contexts_per_slot": a.in_flight,
        "queries_per_launch_stage1": getattr(comp, "gang", 1),
        "ms_per_batch": dt / a.steps * 1e3,
        "host_ms_per_batch_in": {k: v / a.steps * 1e3 for k, v in host.items()},
        **({"images_in": {"frame": a.images, "keypoints_per_frame": n_kp[0], "frames_extracted_per_batch": len(mine_i),
                          "extract_workers": a.extract_workers, "extract_ms_per_batch": t_extract[0] / a.steps * 1e3,
                          "feature_block_bytes": block, "feature_allgather_bytes_per_batch_per_rank": per * block}}
           if a.images else {}),
        "rank_rate_queries_per_s": a.steps * B / dt,
        "note": "all ranks work in lock step, so this is also the predicted whole-job rate of N ranks, collectives' "
                "latency excluded",
        "stage2_queries_per_batch": len(res), "localised_of_own": f"{ok}/{a.steps * len(res)}",
        "same_as_unsharded": f"{same}/{len(checked)}", "first_batch_ok": n_ok_check,
        "redo": loc.counters()["batches_exchanged_again_with_a_larger_budget"]}), flush=True)
    comp.close()
    for dq in dqs0:
        dq.close()
    dm0.close()


if __name__ == "__main__":
    main()
