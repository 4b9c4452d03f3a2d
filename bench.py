#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

metric : query images localised per second (BASELINE.json), whole job over all N GPUs
step   : one pass of the hot path over one query (inputs resident in HBM before the timed region)
N = 1  : BASELINE.json configs[1] -- 1 k-image / 2 M-descriptor synthetic map, 2 k feats/query
N > 1  : the same map sharded by view across the N ranks (strong scaling)

Prints ONE JSON line (rank 0).  Adds `roofline` (dominant kernel = K1 Hamming 2-NN) and `cpu_baseline`
(the C oracle timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

# one HW queue per in-flight context (sfmlocalization_amd/_lib.py); the sharded path keeps two slots of contexts
# (so does the image-in mode: a context and an extractor stream per worker)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16" if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or
                                                    os.environ.get("SFMLOC_BENCH_FORCE_SHARDED") == "1" or
                                                    "--from-images" in sys.argv) else "8")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)
VALU_PEAK_TOPS = 39.1  # measured ceiling of the xor+bcnt instruction mix at 8 waves/SIMD (profiles/r01_valu_rates.jsonl)
OPS_PER_PAIR = 35  # 16 xor + 16 bcnt + lshl_or + med3 + min (hamming.hip)


# HBM bytes per K1 launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc runs of this same
# command, corrected as MI355X_MICROARCH.md prescribes; tools/run_profile.sh + tools/pmc_summary.py)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_pmc_summary_screen.json")


def pmc_traffic(default_workload):
    """HBM bytes per launch of the dominant kernel, or None when no PMC pass exists for this workload."""
    if not default_workload or not os.path.exists(PMC_SUMMARY):
        return None, None
    with open(PMC_SUMMARY) as fh:
        d = json.load(fh)
    tot = 0.0
    for k in ("k_hamming_screen", "k_hamming_rows"):
        if k not in d or "hbm_bytes_per_dispatch" not in d[k]:
            return None, None
        tot += d[k]["hbm_bytes_per_dispatch"]["total"]
    return tot, os.path.relpath(PMC_SUMMARY, ROOT)


def hbm_regime():
    """The same kernel family where it IS HBM-bound: few query rows per bank pass, measured by
    tools/hbm_sweep.py on the GPU box and committed; reported next to the headline workload's (VALU-bound) figure."""
    path = os.path.join(ROOT, "profiles", "r01_k1_small_nq_hbm_sweep.jsonl")
    if not os.path.exists(path):
        return None
    rows = [json.loads(ln) for ln in open(path) if ln.strip()]
    rows = [r for r in rows if r.get("nq", 99) <= 8]
    if not rows:
        return None
    best = max(rows, key=lambda r: r["bank_GBps"])
    return {"nq": best["nq"], "bank_rows": best["rows"], "achieved": best["bank_GBps"], "unit": "GB/s of bank bytes",
            "peak": HBM_PEAK_GBS, "frac": best["bank_GBps"] / HBM_PEAK_GBS,
            "note": "plus 12.5 % partial-result writes; 6.3 TB/s is the measured copy peak of this chip",
            "source": os.path.relpath(path, ROOT)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--views", type=int, default=1000)
    ap.add_argument("--desc-per-view", type=int, default=2000)
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--queries", type=int, default=8, help="distinct synthetic queries cycled through")
    ap.add_argument("--in-flight", type=int, default=4, help="queries in flight (contexts); 1 = latency mode")
    ap.add_argument("--batch", type=int, default=0,
                    help="queries per all-gather when --gpus > 1 (a multiple of the world size keeps the P3P stage balanced; "
                         "small batches keep the two-slot pipeline full over a short timed region).  Default: the world "
                         "size, at least 4 -- measured on one rank: 411 queries/s over 40 steps with 4, 383 with 8")
    ap.add_argument("--bow-knn", type=int, default=0,
                    help="> 0: BASELINE configs[2] -- every query first shortlists this many views by BoW distance "
                         "(sfmloc_bow_select over a synthetic .bow matrix) and runs the path on those (1 GPU only); "
                         "use with --views 10000")
    ap.add_argument("--from-images", action="store_true",
                    help="image-in serving mode (1 GPU): every step first extracts AKAZE + M-LDB features from a synthetic "
                         "640x480 image on the GPU (sfmloc_akaze_detect_and_compute, its own stream), then localises the "
                         "step's query; --in-flight worker threads, one extractor and one context each.  The map is "
                         "synthetic, so the localised descriptors are the synthetic query's, not the image's: the point is "
                         "the cost of extraction sharing the GPU and the host with the path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def cpu_baseline(m, queries, seconds):
    """The C oracle's restatement of the whole per-query path on this host's cores (OpenMP), on a bounded
    sample of the SAME workload: exact 2-NN + ratio on a sample of views (scaled to the full bank: that stage is
    linear in rows) plus the query's own place's views, where every later stage (F-matrix AC-RANSAC, 2D-3D set,
    P3P AC-RANSAC) does all its work."""
    from oracle import oracle_c, pipeline as opipe
    threads = max(1, min(16, os.cpu_count() or 1, oracle_c.max_threads()))
    q = queries[0]
    n_cal = max(threads, m.n_views // 50)
    t0 = time.perf_counter()
    oracle_c.match_to_query(q.desc, m.desc, m.view_off, np.arange(n_cal, dtype=np.uint32), 0.6, threads=threads)
    per_view = (time.perf_counter() - t0) / n_cal
    n_sample = int(min(m.n_views, max(n_cal, seconds / max(per_view, 1e-9))))
    place_views = np.nonzero(m.view_place == q.place)[0]
    sel = np.unique(np.concatenate([np.arange(n_sample), place_views])).astype(np.uint32)
    r = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=sel, ransac_round=25, threads=threads)
    rows = int(sum(int(m.view_off[v + 1] - m.view_off[v]) for v in sel))
    t_full = r["t_putative"] * (m.n_rows / rows) + r["t_rest"]
    # two more rows SURVEY 8(d) asks for, on smaller samples: the same port on ONE core, and the reference's I/O
    # pattern -- it re-reads every view's .desc file for every query (MatchUtils.cpp:328-332)
    n1 = max(4, min(len(sel), m.n_views // 40))
    t0 = time.perf_counter()
    oracle_c.match_to_query(q.desc, m.desc, m.view_off, np.arange(n1, dtype=np.uint32), 0.6, threads=1)
    t_one = (time.perf_counter() - t0) / n1 * m.n_views + r["t_rest"]
    import tempfile
    from sfmlocalization_amd import fileio
    n_io = max(8, min(len(sel), m.n_views // 10))
    with tempfile.TemporaryDirectory() as td:
        for v in range(n_io):
            fileio.write_desc(os.path.join(td, f"v{v}.desc"), m.desc[int(m.view_off[v]):int(m.view_off[v + 1])])
        t0 = time.perf_counter()
        for v in range(n_io):
            fileio.read_desc(os.path.join(td, f"v{v}.desc"))
        t_io = (time.perf_counter() - t0) / n_io * m.n_views
    return {"value": 1.0 / t_full, "unit": "queries/s", "cores": threads, "kind": "port",
            "single_core_value": 1.0 / t_one, "with_per_query_desc_reread_value": 1.0 / (t_full + t_io),
            "sample": f"1 query ({q.desc.shape[0]} feats): exact 2-NN + ratio against {len(sel)}/{m.n_views} views "
                      f"({rows} rows, {r['t_putative']:.2f}s, scaled to the full bank) + F-matrix AC-RANSAC, 2D-3D set "
                      f"and P3P AC-RANSAC on the surviving views ({r['t_rest']:.3f}s, localised={r['ok']}); "
                      "C oracle, OpenMP, -O3 -march=x86-64-v3"}


def cpu_baseline_shortlist(m, queries, bow, qbow, knn, seconds):
    """configs[2] on the host: exact L2 shortlist over the .bow matrix (NumPy), then the C oracle's whole path on the
    shortlisted views (OpenMP), whole queries until `seconds` are spent."""
    from oracle import oracle_c, pipeline as opipe
    threads = max(1, min(16, os.cpu_count() or 1, oracle_c.max_threads()))
    t0 = time.perf_counter()
    done, ok, t_bow = 0, 0, 0.0
    while done < len(queries) and (done == 0 or time.perf_counter() - t0 < seconds):
        q = queries[done]
        t1 = time.perf_counter()
        d = ((bow - qbow[done][None, :]) ** 2).sum(1)
        sel = np.sort(np.argsort(d, kind="stable")[:knn]).astype(np.uint32)
        t_bow += time.perf_counter() - t1
        r = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=sel, ransac_round=25, threads=threads)
        ok += int(bool(r["ok"]))
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": f"{done} whole queries ({queries[0].desc.shape[0]} feats): exact BoW shortlist k={knn} of {m.n_views} "
                      f"views in NumPy ({t_bow / done * 1e3:.1f} ms/query) + the C oracle's path on the shortlisted views "
                      f"(exact 2-NN + ratio, F-matrix AC-RANSAC, 2D-3D set, P3P AC-RANSAC; localised {ok}/{done}); "
                      "OpenMP, -O3 -march=x86-64-v3"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.batch <= 0:
        a.batch = world * max(1, (4 + world - 1) // world)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    forced = os.environ.get("SFMLOC_BENCH_FORCE_SHARDED") == "1"
    if world > 1 or (forced and "RANK" in os.environ):
        import torch.distributed as dist_
        dist = dist_
        # SFMLOC_BENCH_BACKEND=gloo: rehearse N ranks on fewer GPUs (ranks share devices, the exchange goes through the
        # host); the measured configuration is always nccl = RCCL, one rank per GPU
        backend = os.environ.get("SFMLOC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)

    import sfmlocalization_amd as S
    from sfmlocalization_amd import synth

    # every rank builds the same seeded map and keeps its shard of views (contiguous view ranges)
    m = synth.make_map(2, n_views=a.views, desc_per_view=a.desc_per_view)
    queries = [synth.make_query(m, 1000 + i, n_feat=a.nq) for i in range(a.queries)]
    bow = qbow = None
    if a.bow_knn > 0:
        # N > 1 (BASELINE configs[3]): the .bow matrix shards with the views; every batch first runs the sharded
        # shortlist (one small all-gather of each rank's k best, dist.py bow_shortlists), then the sharded path
        # one BoW prototype per place + noise per view: the shortlist finds the query's place (TrainBoW's vectors are
        # 500-dimensional, BoFUtils.cpp:43-45)
        rng = np.random.Generator(np.random.PCG64(33))
        place_bow = rng.uniform(0, 1, (len(m.place_center), 500)).astype(np.float32)
        bow = (place_bow[m.view_place] + rng.normal(0, 0.05, (a.views, 500))).astype(np.float32)
        qbow = [(place_bow[q.place] + rng.normal(0, 0.05, 500)).astype(np.float32) for q in queries]
    v0 = (a.views * rank) // world
    v1 = (a.views * (rank + 1)) // world
    r0, r1 = int(m.view_off[v0]), int(m.view_off[v1])
    params = S.default_params(device=local_rank, profile=1, ransac_round=25)
    dev_map = S.Map(m.view_id[v0:v1], m.view_off[v0:v1 + 1] - m.view_off[v0], m.desc[r0:r1], params=params,
                    view_wh=m.view_wh[v0:v1], kpt_xy=m.kpt_xy[r0:r1], row_landmark=m.row_landmark[r0:r1],
                    landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
                    bow=None if bow is None else bow[v0:v1])
    dqs = [dev_map.query(q.desc, q.kpt_xy, q.width, q.height) for q in queries]
    lat = []
    n_ok = [0]
    nctx = max(1, a.in_flight)
    sharded = None
    # SFMLOC_BENCH_FORCE_SHARDED=1: rehearse the N>1 code path (parts, collective, merge) on a single rank
    if world > 1 or forced:
        # bank sharded by view; per batch: stage 1 on every shard, ONE all-gather of candidate parts over RCCL,
        # stage 2 of each query on its owner rank (sfmlocalization_amd/dist.py)
        from sfmlocalization_amd import dist as D
        cap = 4096
        comp = D.HipShardCompute(dev_map, cap, n_contexts=nctx, device=torch.device("cuda", local_rank))
        sharded = D.ShardedLocalizer(comp, cap, rank=rank, world=world, always_gather=forced and dist is not None)
    ctxs = [dev_map.context() for _ in range(nctx)] if sharded is None else []
    t_begin = [0.0] * nctx
    busy = [False] * nctx

    def finish(k):
        pose, _, _ = ctxs[k].end()
        lat.append(time.perf_counter() - t_begin[k])
        n_ok[0] += int(pose.ok)
        busy[k] = False

    def run_sharded(first, count):
        """`count` steps starting at step `first`, in batches of a.batch through the two-slot pipeline: stage 1 of
        batch b+1 is queued before batch b's all-gather and P3P stage."""
        idx = list(range(first, first + count))
        batches = [idx[k:k + a.batch] for k in range(0, len(idx), a.batch)]
        t_mark = [time.perf_counter(), time.perf_counter()]   # batch b was enqueued when batch b-2 was yielded
        sels = None
        if qbow is not None:   # generator: the shortlist of batch b is computed when the pipeline reaches it
            sels = (sharded.bow_shortlists(dev_map, [qbow[i % len(dqs)] for i in b], a.bow_knn) for b in batches)
        stream = sharded.localize_stream([[dqs[i % len(dqs)] for i in b] for b in batches], gather_results=False,
                                         view_sels=sels)
        for b, res in zip(batches, stream):
            now = time.perf_counter()
            lat.extend([now - t_mark[0]] * len(b))          # a query's latency in batch mode = its batch's wall time
            t_mark = [t_mark[1], now]
            n_ok[0] += sum(int(r["ok"]) for r in res.values())

    def step(i):
        # one step = one query through the whole path; up to `nctx` steps overlap on the GPU
        k = i % nctx
        if busy[k]:
            finish(k)
        t_begin[k] = time.perf_counter()
        if qbow is not None:   # configs[2]: shortlist + path in one asynchronous call, the shortlist stays on the device
            ctxs[k].begin_bow(dqs[i % len(dqs)], qbow[i % len(dqs)], a.bow_knn)
        else:
            ctxs[k].begin(dqs[i % len(dqs)])
        busy[k] = True

    def run(first, count):
        if sharded is not None:
            run_sharded(first, count)
        else:
            for i in range(first, first + count):
                step(i)

    def drain():
        for k in range(nctx):
            if sharded is None and busy[k]:
                finish(k)

    def fence():
        drain()
        dev_map.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    img_mode = None
    if a.from_images:
        if sharded is not None:
            raise SystemExit("--from-images is a single-GPU mode")
        import threading
        imgs = [synth.texture_image(900 + k, 480, 640) for k in range(4)]
        extractors = [S.Akaze(640, 480, device=local_rank) for _ in range(nctx)]
        n_feat = [0]
        lock = threading.Lock()

        def worker(k, first, count):
            for i in range(first + k, first + count, nctx):
                t1 = time.perf_counter()
                kp, _ = extractors[k].detect_and_compute(imgs[i % len(imgs)])
                if qbow is not None:
                    ctxs[k].begin_bow(dqs[i % len(dqs)], qbow[i % len(dqs)], a.bow_knn)
                else:
                    ctxs[k].begin(dqs[i % len(dqs)])
                pose, _, _ = ctxs[k].end()
                with lock:
                    lat.append(time.perf_counter() - t1)
                    n_ok[0] += int(pose.ok)
                    n_feat[0] = len(kp)

        def run_images(first, count):
            ts = [threading.Thread(target=worker, args=(k, first, count)) for k in range(nctx)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()

        run = run_images  # noqa: F811
        img_mode = {"extractors": extractors, "n_feat": n_feat}

    run(0, a.warmup)
    fence()
    dev_map.stats_reset()
    lat.clear()
    n_ok[0] = 0
    fence()
    t0 = time.perf_counter()
    run(0, a.steps)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = dev_map.stats()
    k1_ms = st.total_ms[0] / max(1, st.launches[0])
    lat_throughput = list(lat)
    # per-query latency proper: the same queries one at a time (outside the timed region; with several queries in
    # flight a query's wall time is mostly queueing behind the others' Hamming scans)
    lat_single = []
    iso = None
    if sharded is None:
        dev_map.stats_reset()
        dev_map.set_profile(2)   # K1 events only: every bracketed stage costs ~10 us of idle GPU on the critical path
        for i in range(min(a.steps, 32)):
            t1 = time.perf_counter()
            if qbow is not None:
                ctxs[0].begin_bow(dqs[i % len(dqs)], qbow[i % len(dqs)], a.bow_knn)
            else:
                ctxs[0].begin(dqs[i % len(dqs)])
            ctxs[0].end()
            lat_single.append(time.perf_counter() - t1)
        st1 = dev_map.stats()
        iso = (st1.total_ms[0] / max(1, st1.launches[0]), st1.hamming_lane_ops / max(1, st1.launches[0]))
    sel0 = None
    if qbow is not None:   # this rank's part of the first query's shortlist (collective when sharded)
        sel0 = (sharded.bow_shortlists(dev_map, [qbow[0]], a.bow_knn)[0] if sharded is not None
                else dev_map.bow_select(qbow[0], a.bow_knn))
    dev_map.match_putative(dqs[0], sel0)  # outside the timed region: number of emitted matches for the byte count
    n_match = int(dev_map.putative_read()[0].sum())
    if world > 1:
        ok_t = torch.tensor([n_ok[0]], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok_t)
        n_ok[0] = int(ok_t.item())
    rows_rank = r1 - r0
    if sel0 is not None:   # only the shortlisted views are scanned
        rows_rank = int(sum(int(m.view_off[v + 1] - m.view_off[v]) for v in sel0))
    alg_bytes = 64 * rows_rank + 64 * a.nq + 12 * n_match
    achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
    pairs = rows_rank * a.nq
    # VALU lane-instructions K1 actually issued (counted by the library: the screening kernel rejects most pairs on
    # a 10-dword prefix distance, so this is below 35 per pair)
    lane_ops = st.hamming_lane_ops / max(1, st.launches[0])
    valu = lane_ops / (k1_ms * 1e-3) / 1e12

    traffic, traffic_src = pmc_traffic(world == 1 and (a.views, a.desc_per_view, a.nq) == (1000, 2000, 2000))
    if rank == 0:
        out = {
            "metric": "query images localized/sec",
            "value": a.steps / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32 popcount (Hamming)",
            "data": "synthetic",
            "config": {"workload": f"{a.views}-image / {m.n_rows}-descriptor synthetic map, {a.nq} feats/query, "
                                   "whole per-query path: brute-force Hamming 2-NN + Lowe ratio -> >=16 filter -> "
                                   "F-matrix AC-RANSAC (25 rounds) -> 2D-3D set -> P3P AC-RANSAC (4096) -> pose; "
                                   f"{nctx} queries in flight"
                                   + (f"; every query first shortlists {a.bow_knn} views by exact L2 over the "
                                      f"{a.views} x 500 .bow matrix (BASELINE configs[2])" if a.bow_knn > 0 else ""),
                       "views": a.views, "rows": int(m.n_rows), "nq": a.nq,
                       "parallelism": (f"bank sharded by view x{world}, one all-gather of candidate parts per "
                                       f"{a.batch}-query batch" if world > 1 else "1 GPU, whole bank"),
                       "queries_localised": f"{n_ok[0]}/{a.steps}"},
            "latency_ms": {"p50": float(np.percentile(lat_single or lat_throughput, 50) * 1e3),
                           "p95": float(np.percentile(lat_single or lat_throughput, 95) * 1e3),
                           "mode": "one query in flight" if lat_single else f"{a.batch}-query batches",
                           "p50_at_throughput": float(np.percentile(lat_throughput, 50) * 1e3),
                           "p95_at_throughput": float(np.percentile(lat_throughput, 95) * 1e3)},
            "stage_ms": {"putMatch(K1+K2)": (st.total_ms[0] + st.total_ms[1]) / a.steps,
                         "geoMatch(K3)": st.total_ms[2] / a.steps, "matchSet(K4)": st.total_ms[3] / a.steps,
                         "PnP(K5)": st.total_ms[4] / a.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC FETCH_SIZE*2+WRITE_SIZE)", "traffic_source": traffic_src,
                         "kernel": "k_hamming_screen (+k_hamming_rows)", "kernel_ms": k1_ms,
                         "algorithmic_bytes": alg_bytes,
                         "note": "SURVEY F7: at N_q=2000 the kernel is VALU-bound (intensity N_q/2 lane-ops/byte)",
                         "valu": {"achieved": valu, "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s",
                                  "frac": valu / VALU_PEAK_TOPS, "ops_per_pair_exact": OPS_PER_PAIR,
                                  "ops_per_pair_issued": lane_ops / max(1, pairs),
                                  "pairs_per_s": pairs / (k1_ms * 1e-3),
                                  "pairs_finished_frac": st.hamming_pairs_finished / max(1, st.hamming_pairs),
                                  "rows_flagged_per_query": st.hamming_rows_flagged / max(1, st.launches[0])}},
        }
        # with several queries in flight the launches overlap: a launch's event bracket (kernel_ms) then contains the
        # others' work, so also report what the chip did over the whole timed region (all launches' work / wall time)
        n_launch = max(1, st.launches[0])
        out["roofline"]["timed_region_aggregate"] = {
            "achieved": alg_bytes * n_launch / dt / 1e9, "unit": "GB/s",
            "frac": alg_bytes * n_launch / dt / 1e9 / HBM_PEAK_GBS,
            "valu_achieved": st.hamming_lane_ops / dt / 1e12, "valu_frac": st.hamming_lane_ops / dt / 1e12 / VALU_PEAK_TOPS,
            "launches": int(n_launch), "wall_ms": dt * 1e3,
            "note": "K1 launches of the timed region only; dt is the region's wall time (max over ranks)"}
        hr = hbm_regime()
        if hr is not None:
            out["roofline"]["hbm_bound_regime"] = hr
        if img_mode is not None:
            out["config"]["workload"] += ("; image-in mode: each step first runs AKAZE + M-LDB extraction of a 640x480 "
                                          f"synthetic image on the GPU ({img_mode['n_feat'][0]} keypoints)")
            out["latency_ms"]["p50_image_in_at_throughput"] = float(np.percentile(lat_throughput, 50) * 1e3)
        if iso is not None:
            # the same kernel with nothing else on the GPU (the latency phase): what the kernel itself achieves; in
            # the timed region its launches share the chip with the other queries in flight, which stretches them
            out["roofline"]["isolated"] = {
                "kernel_ms": iso[0], "achieved": alg_bytes / (iso[0] * 1e-3) / 1e9,
                "frac": alg_bytes / (iso[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "valu_achieved": iso[1] / (iso[0] * 1e-3) / 1e12,
                "valu_frac": iso[1] / (iso[0] * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                "pairs_per_s": pairs / (iso[0] * 1e-3)}
        if not a.no_cpu_baseline and world == 1:   # the CPU leg is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = (cpu_baseline_shortlist(m, queries, bow, qbow, a.bow_knn, a.cpu_seconds)
                                   if a.bow_knn > 0 else cpu_baseline(m, queries, a.cpu_seconds))
        print(json.dumps(out), flush=True)
    if img_mode is not None:
        for e in img_mode["extractors"]:
            e.close()
    for c in ctxs:
        c.close()
    if sharded is not None:
        sharded.compute.close()
    for dq in dqs:
        dq.close()
    dev_map.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
