#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

metric : query images localised per second (BASELINE.json), whole job over all N GPUs
step   : one pass of the hot path over one BATCH of `--batch` queries (256: the batch BASELINE configs[3] names);
         query descriptors, keypoints and BoW vectors are resident in HBM before the timed region
N = 1  : BASELINE.json configs[2] -- 10 k-image / 20 M-descriptor synthetic map, 2 k feats/query, every query first
         shortlists k = 100 views by exact L2 over the 10 000 x 500 .bow matrix, then runs the whole path on them
N > 1  : the same map and workload with the bank (and the .bow matrix) sharded by view across the N ranks
         (BASELINE configs[3]/[4]): sharded shortlist (one small all-gather of per-shard k best), shard
         local K1..K3, ONE all-gather of candidate parts per batch over RCCL, P3P of query i on rank i mod N; both
         stages in gang sessions (--gang: 16 queries per kernel launch).  Strong scaling on a fixed map.

`python bench.py --gpus N` without a launcher starts the N ranks itself (before any GPU call) through
torch.distributed.run and relays rank 0's JSON line; under a launcher a rank insists on WORLD_SIZE == --gpus.

Prints ONE JSON line (rank 0).  `roofline` = the dominant kernel family (K1 Hamming 2-NN) measured live, one launch
in flight, on the FULL 20 M-row bank of the same map (SURVEY 8d cfg 3); `cpu_baseline` = the C oracle timed on this
host's cores on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0
# Issue ceiling of K1's instruction mix (v_xor_b32 + v_bcnt_u32_b32 1:1): tools/valu_rates.hip at 8 waves per SIMD on an
# exactly balanced grid (LDS-sized so that every SIMD holds exactly 8 waves; census in profiles/r02_valu_rates.jsonl),
# lane-ops over WALL time.  The per-wave cycle stamps of the same run must not be used: all waves start within 0.7 us
# but finish up to 1.4 ms apart (the SIMD's arbitration is not fair), so a wave's own duration divided by the waves per
# SIMD over-reads the SIMD's rate by ~1.6x -- that is where round 1's "51 T from the stamps" came from.
VALU_PEAK_TOPS = 41.3
OPS_PER_PAIR = 35  # 16 xor + 16 bcnt + lshl_or + med3 + min (hamming.hip)
# Issue rates per instruction class at 8 waves per SIMD on the balanced grid (profiles/r02_valu_rates.jsonl, lane-ops over
# wall time): VOP2-encoded integer ops (v_xor_b32 58.96 T; v_cmp / v_min issue at that rate) and the VOP3-only ones
# (v_bcnt_u32_b32 36.70, v_med3_u32 37.16, v_lshl_or_b32 likewise).  A kernel's ceiling is the HARMONIC combination of
# the two for its own class counts -- time = vop2 / R2 + vop3 / R3 -- so a fraction against it cannot exceed 1
# (the 1:1 "mix" figure above lost 8 % to alternation and K1 read 1.03 of it).
VOP2_TOPS, VOP3_TOPS = 58.96, 36.93
# lane-ops per (bank row, query row) pair by class, from hamming.hip's own accounting (hamming.hip:341-347):
#   exact head pair (first 64 query rows): 16 xor + min | 16 bcnt + lshl_or + med3
#   screened pair (10 of 16 dwords):        10 xor + cmp | 10 bcnt
#   a pair that is finished (counted on the device): + 6 xor + min + 2 | + 6 bcnt + lshl_or + med3
K1_CLASS_OPS = {"head": (17, 18), "screened": (11, 10), "finished_extra": (9, 8)}
K1_HEAD_ROWS = 64

# HBM bytes per launch of the roofline kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc
# runs of this command, corrected as MI355X_MICROARCH.md prescribes; tools/run_profile.sh + tools/pmc_summary.py).
# PMC cannot be collected from inside this process, so this one field is read from the committed summary.
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_summary_fullscan.json") for r in (4, 3))
                    if os.path.exists(p)), os.path.join(ROOT, "profiles", "r04_pmc_summary_fullscan.json"))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="queries per step (and per all-gather when --gpus > 1)")
    ap.add_argument("--views", type=int, default=10000)
    ap.add_argument("--desc-per-view", type=int, default=2000)
    ap.add_argument("--nq", type=int, default=2000)
    ap.add_argument("--bow-knn", type=int, default=100,
                    help="views shortlisted per query by BoW distance (BASELINE configs[2]); 0 = every view is "
                         "scanned (configs[1] with --views 1000)")
    ap.add_argument("--queries", type=int, default=64, help="distinct synthetic queries cycled through")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="queries in flight per GPU (contexts); 0 = 12 with a shortlist, 4 for full scans; 1 = latency mode")
    ap.add_argument("--from-images", action="store_true",
                    help="image-in serving mode (1 GPU): every query first extracts AKAZE + M-LDB features from a synthetic "
                         "640x480 image on the GPU (sfmloc_akaze_detect_and_compute, its own stream); --in-flight worker "
                         "threads, one extractor and one context each.  The map is synthetic, so the localised descriptors "
                         "are the synthetic query's, not the image's: the point is the cost of extraction sharing the GPU")
    ap.add_argument("--image-batch", type=int, default=8,
                    help="--from-images: frames a worker takes at a time -- extracted together "
                         "(sfmloc_akaze_detect_and_compute_batch: one launch per kernel for all of them) and localised in one "
                         "gang session on the worker's one stream; 1 = a frame at a time")
    ap.add_argument("--gang", type=int, default=0,
                    help="--gpus > 1: queries per launch in both stages of the sharded path (gang sessions, "
                         "sfmloc_gang_begin/_end); 0 = 32 with more than one rank, 1 = one query per launch")
    ap.add_argument("--queries-per-stream", type=int, default=int(os.environ.get("SFMLOC_BENCH_QUERIES_PER_STREAM", "1")),
                    help="1 GPU: queries a context stream carries per turn, as one gang session (sfmloc_gang_begin/_end: one "
                         "launch per kernel for all of them); in flight = --in-flight x this")
    ap.add_argument("--threads", type=int, default=0,
                    help="host threads driving the contexts (1 GPU, no shortlist collective): 0/1 = one thread round-robins "
                         "all contexts; N > 1 = N threads, each with its share of the contexts (the C ABI calls release the "
                         "GIL, so the kernel launches of different queries are issued in parallel)")
    ap.add_argument("--replicas", action="store_true",
                    help="--gpus N: every rank holds the WHOLE map and takes the queries i mod N of each batch, no collective "
                         "(the map fits one GPU 45 times over: what sharding is compared with).  Without this flag a run on "
                         "more than one GPU measures the sharded path and then this one, and prints it as `replicas`")
    ap.add_argument("--no-replica-leg", action="store_true", help="--gpus N: skip the replicas comparison leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-phase", action="store_true", help="skip the full-bank scan and the N_q sweep")
    ap.add_argument("--no-real-stats", action="store_true",
                    help="skip the side measurement on M-LDB-like descriptors (its scans run the same kernel on another "
                         "bank: leave it out of a rocprofv3 run whose averages are meant for the 20 M-row scan)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-image-in", action="store_true",
                    help="skip the image-in leg (frame -> AKAZE + M-LDB -> BoW vector from the image -> shortlist -> path -> "
                         "pose on a map whose bank was extracted from rendered views)")
    ap.add_argument("--image-in-only", action="store_true", help="run only the image-in leg (development)")
    ap.add_argument("--image-tiles", type=int, default=8,
                    help="image-in leg: the plane is tiles x tiles places of 16 m x 16 m with a texture each")
    ap.add_argument("--image-views", type=int, default=0,
                    help="image-in leg: map views rendered and extracted on the GPU; 0 = all of --views (the default: the whole "
                         "configs[2] map is extracted).  Fewer: the rest of --views are padding views of random descriptors")
    ap.add_argument("--no-image-in-1080p", action="store_true", help="skip the 1920x1080 image-in leg (`image_in_1080p`)")
    ap.add_argument("--image-views-1080p", type=int, default=1000,
                    help="1080p image-in leg: map views (all rendered at 1920x1080 and extracted; no padding views)")
    ap.add_argument("--image-steps-1080p", type=int, default=2, help="1080p image-in leg: timed batches of --batch-1080p frames")
    ap.add_argument("--batch-1080p", type=int, default=128)
    ap.add_argument("--image-oracle-frames", type=int, default=16,
                    help="image-in leg: frames taken through the ORACLE's whole chain after the timed region and compared "
                         "with the device's features, BoW vector, shortlist, pose and inlier set (4 at 1080p)")
    ap.add_argument("--image-steps", type=int, default=6, help="image-in leg: timed batches of --batch frames")
    ap.add_argument("--image-workers", type=int, default=10,
                    help="image-in leg: worker threads (a stream for extraction + path and one for the BoW chains each)")
    ap.add_argument("--image-bow-chain-streams", dest="image_bow_worker_stream", action="store_false",
                    help="image-in leg: one stream per BoW chain (default: a worker's chains in turn on one stream of their own "
                         "-- hardware queues are what the leg runs out of: 549 -> 660 images/s at 4 workers, 815 at 10)")
    ap.add_argument("--image-bow-shared-stream", dest="image_bow_own_stream", action="store_false",
                    help="image-in leg: the BoW-vector chain of a frame behind its feature extraction on the worker's stream "
                         "(default: on a stream of its own beside it, joined by an event: 660 instead of 540 images/s and "
                         "3.2 instead of 4.0 ms)")
    return ap.parse_args(argv)


def spawn_ranks(a):
    """`--gpus N` (N > 1) without a launcher: start N fresh ranks as CHILD processes -- this parent has made no GPU
    call and never will -- relay their output (rank 0 prints the JSON line) and return their exit status."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def pmc_traffic(match):
    """HBM bytes per launch of the roofline kernel, or None when no PMC pass exists for this workload."""
    if not match or not os.path.exists(PMC_SUMMARY):
        return None, None
    with open(PMC_SUMMARY) as fh:
        d = json.load(fh)
    k = "k_hamming_screen<8, 10, 1>"       # the full-bank scan (the short-list form is <8, 10, 4>)
    if k not in d or "hbm_bytes_per_dispatch" not in d[k]:
        return None, None
    return d[k]["hbm_bytes_per_dispatch"]["total"], os.path.relpath(PMC_SUMMARY, ROOT)


def fingerprint(pose, pq, pl):
    """capi.result_fingerprint: every bit of a query's result except the stage timings, as 8 bytes."""
    from sfmlocalization_amd import capi
    return capi.result_fingerprint(pose, pq, pl)


def cpu_baseline(m, queries, seconds):
    """The C oracle's restatement of the whole per-query path on this host's cores (OpenMP), on a bounded
    sample of the full-scan workload: exact 2-NN + ratio on a sample of views (scaled to the full bank: that stage is
    linear in rows) plus the query's own place's views, where every later stage (F-matrix AC-RANSAC, 2D-3D set,
    P3P AC-RANSAC) does all its work."""
    import numpy as np
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
    import numpy as np
    from oracle import oracle_c, pipeline as opipe
    threads = max(1, min(16, os.cpu_count() or 1, oracle_c.max_threads()))
    t0 = time.perf_counter()
    done, ok, t_bow = 0, 0, 0.0
    while done == 0 or time.perf_counter() - t0 < seconds:
        q = queries[done % len(queries)]
        t1 = time.perf_counter()
        d = ((bow - qbow[done % len(queries)][None, :]) ** 2).sum(1)
        sel = np.sort(np.argsort(d, kind="stable")[:knn]).astype(np.uint32)
        t_bow += time.perf_counter() - t1
        r = opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=sel, ransac_round=25, threads=threads)
        ok += int(bool(r["ok"]))
        done += 1
    dt = time.perf_counter() - t0
    # the two further rows SURVEY 8(d) asks for: the same port on ONE core, and the reference's I/O pattern -- it re-reads
    # the .bow file of EVERY candidate view (BoFUtils.cpp:33-42) and the .desc file of every shortlisted view
    # (MatchUtils.cpp:328-332) for every query
    q = queries[0]
    d0 = ((bow - qbow[0][None, :]) ** 2).sum(1)
    sel0 = np.sort(np.argsort(d0, kind="stable")[:knn]).astype(np.uint32)
    t1 = time.perf_counter()
    opipe.localize(m, q.desc, q.kpt_xy, (q.width, q.height), view_sel=sel0, ransac_round=25, threads=1)
    t_one = time.perf_counter() - t1 + t_bow / done
    import tempfile
    from sfmlocalization_amd import fileio
    n_io = max(8, min(m.n_views, 200))
    with tempfile.TemporaryDirectory() as td:
        for v in range(n_io):
            fileio.write_mat_bin(os.path.join(td, f"v{v}.bow"), bow[v].astype(np.float64).reshape(-1, 1))
        for v in sel0[:n_io]:
            fileio.write_desc(os.path.join(td, f"v{int(v)}.desc"), m.desc[int(m.view_off[v]):int(m.view_off[v + 1])])
        t1 = time.perf_counter()
        for v in range(n_io):
            fileio.read_mat_bin(os.path.join(td, f"v{v}.bow"))
        t_bow_io = (time.perf_counter() - t1) / n_io * m.n_views
        t1 = time.perf_counter()
        for v in sel0[:n_io]:
            fileio.read_desc(os.path.join(td, f"v{int(v)}.desc"))
        t_desc_io = (time.perf_counter() - t1) / min(n_io, len(sel0)) * len(sel0)
    return {"value": done / dt, "unit": "queries/s", "cores": threads, "kind": "port",
            "single_core_value": 1.0 / t_one,
            "with_per_query_bow_and_desc_reread_value": 1.0 / (dt / done + t_bow_io + t_desc_io),
            "reread_seconds_per_query": {"bow_files_of_every_view": t_bow_io, "desc_files_of_the_shortlist": t_desc_io},
            "sample": f"{done} whole queries ({queries[0].desc.shape[0]} feats): exact BoW shortlist k={knn} of {m.n_views} "
                      f"views in NumPy ({t_bow / done * 1e3:.1f} ms/query) + the C oracle's path on the shortlisted views "
                      f"(exact 2-NN + ratio, F-matrix AC-RANSAC, 2D-3D set, P3P AC-RANSAC; localised {ok}/{done}); "
                      "OpenMP, -O3 -march=x86-64-v3"}


def synth_bow(m, queries, seed=33):
    """One BoW prototype per place + noise per view: the shortlist finds the query's place (TrainBoW's vectors are
    500-dimensional, BoFUtils.cpp:43-45)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    place_bow = rng.uniform(0, 1, (len(m.place_center), 500)).astype(np.float32)
    bow = (place_bow[m.view_place] + rng.normal(0, 0.05, (m.n_views, 500))).astype(np.float32)
    qbow = [(place_bow[q.place] + rng.normal(0, 0.05, 500)).astype(np.float32) for q in queries]
    return bow, qbow


def valu_floor(rows, nq, k1_ms, lane_ops, st_roof):
    """K1's own issue ceiling for one full scan: its lane-ops counted per instruction class (exact head / screened pair /
    finished pair, counted by the kernel) over the two classes' measured issue rates -> dict (floor_ms, frac, ...)."""
    n_head = rows * min(K1_HEAD_ROWS, nq)
    n_scr = rows * max(0, nq - K1_HEAD_ROWS)
    n_fin = st_roof.hamming_pairs_finished / max(1, st_roof.launches[0])
    vop2 = (K1_CLASS_OPS["head"][0] * n_head + K1_CLASS_OPS["screened"][0] * n_scr + K1_CLASS_OPS["finished_extra"][0] * n_fin)
    vop3 = (K1_CLASS_OPS["head"][1] * n_head + K1_CLASS_OPS["screened"][1] * n_scr + K1_CLASS_OPS["finished_extra"][1] * n_fin)
    floor_ms = (vop2 / (VOP2_TOPS * 1e12) + vop3 / (VOP3_TOPS * 1e12)) * 1e3
    pairs = rows * nq
    return {"floor_ms": floor_ms, "frac": floor_ms / k1_ms, "vop2": vop2, "vop3": vop3,
            "achieved": lane_ops / (k1_ms * 1e-3) / 1e12, "peak": (vop2 + vop3) / (floor_ms * 1e-3) / 1e12,
            "ops_per_pair_issued": lane_ops / max(1, pairs), "pairs_per_s": pairs / (k1_ms * 1e-3),
            "pairs_finished_frac": st_roof.hamming_pairs_finished / max(1, st_roof.hamming_pairs),
            "rows_flagged_per_scan": st_roof.hamming_rows_flagged / max(1, st_roof.launches[0])}


def roofline_phase(S, dev_map, dq, rows, nq, n_reps=5):
    """K1 on the FULL bank of this map, one launch in flight, bracketed by HIP events on the stream it runs on
    (sfmloc_stats_read, params.profile = 2): the isolated kernel time that `rocprofv3 --kernel-trace --stats` of this
    command reproduces.  Returns (ms per scan (screen + rows kernels), issued lane-ops per scan, emitted matches,
    stats)."""
    dev_map.set_profile(2)
    dev_map.match_putative(dq)      # warm
    dev_map.sync()
    dev_map.stats_reset()
    for _ in range(n_reps):
        dev_map.match_putative(dq)
        dev_map.sync()              # one in flight
    st = dev_map.stats()
    n_match = int(dev_map.putative_read()[0].sum())
    ms = st.total_ms[0] / max(1, st.launches[0])
    return ms, st.hamming_lane_ops / max(1, st.launches[0]), n_match, st


def nq_sweep_phase(S, dev_map, rows, n_reps=10):
    """The same kernel family where it IS HBM-bound: N_q = 1, 2, 4, 8 query rows per pass over the
    full bank (larger than the 256 MiB Infinity Cache), measured live."""
    import numpy as np
    import synthdata as synth
    rng = np.random.Generator(np.random.PCG64(5))
    out = []
    for nq in (1, 2, 4, 8):
        q = dev_map.query(synth.random_descriptors(rng, nq))
        for _ in range(2):
            dev_map.match_putative(q)
        dev_map.sync()
        dev_map.stats_reset()
        for _ in range(n_reps):
            dev_map.match_putative(q)
            dev_map.sync()
        st = dev_map.stats()
        ms = st.total_ms[0] / max(1, st.launches[0])
        out.append({"nq": nq, "kernel_ms": ms, "bank_GBps": (rows * 64 + nq * 64) / (ms * 1e-3) / 1e9})
        q.close()
    return out


def real_statistics_phase(S, synth, device):
    """The screening Hamming kernel on descriptors that look like M-LDB (K9-extracted from textured images, 222 +- 42
    bits between unrelated ones) instead of uniform random bits (243 +- 11), beside the exact kernel on the same bank:
    the gain the bound really buys on image data.  Outside the timed region; ~400 k rows x 2 000 query rows."""
    import numpy as np
    q, bank, stats = synth.mldb_like_bank(S, device=device)
    view_off = np.arange(0, len(bank) + 1, len(bank) // 200, dtype=np.uint32)
    view_off[-1] = len(bank)
    out = dict(stats)
    for exact in (0, 1):
        p = S.default_params(device=device, profile=2, exact_rows=exact)
        with S.Map(np.arange(len(view_off) - 1, dtype=np.uint32), view_off, bank, params=p) as dm:
            dq = dm.query(q)
            dm.match_putative(dq)
            dm.sync()
            dm.stats_reset()
            for _ in range(10):
                dm.match_putative(dq)
                dm.sync()
            st = dm.stats()
            out["exact_kernel" if exact else "screening_kernel"] = {
                "kernel_ms": st.total_ms[0] / st.launches[0], "lane_ops_per_pair": st.hamming_lane_ops / st.hamming_pairs,
                "pairs_finished_frac": st.hamming_pairs_finished / st.hamming_pairs,
                "matches": int(dm.putative_read()[0].sum())}
            dq.close()
    out["same_matches"] = out["exact_kernel"]["matches"] == out["screening_kernel"]["matches"]
    out["screening_gain"] = out["exact_kernel"]["kernel_ms"] / out["screening_kernel"]["kernel_ms"]
    return out


def oracle_image_chain(world, frame, bow_model, knn):
    """The whole image-in chain on the host, by the oracle: AKAZE + M-LDB of the frame, the BoW vector of the frame
    (dense gray -> dense-grid descriptors -> PCA / eigenvalue -> BoF), exact L2 shortlist, then the path on the
    shortlisted views.  -> dict with kp, desc, bow, shortlist and pipeline.localize's result."""
    import numpy as np
    from oracle import oracle_c, pipeline as opipe
    from sfmlocalization_amd import engine
    m = world.m
    kp, desc = oracle_c.akaze_detect_and_compute(frame)[:2]
    desc64 = np.zeros((len(desc), 64), np.uint8)
    desc64[:, :desc.shape[1]] = desc
    pca, bowm = bow_model
    gray = oracle_c.dense_gray(np.stack([frame, frame, frame], 2), 300)
    grid = engine.dense_grid_keypoints(300)
    dd, _ = oracle_c.akaze_compute(gray, grid)
    bow = oracle_c.bof(dd[:, :61].astype(np.float32), grid[:, :2].copy(), bowm["Centers"], 300, 2, 2,
                       pca["MeanPCA"], pca["EigenVectorsPCA"], pca["EigenValuesPCA"], int(pca["DimPCA"]))
    sel = np.sort(oracle_c.bow_select(world.bow, bow.astype(np.float32), knn)).astype(np.uint32)
    r = opipe.localize(m, desc64, kp[:, :2].copy(), (m.width, m.height), view_sel=sel, ransac_round=25,
                       threads=max(1, min(16, os.cpu_count() or 1)))
    return {"kp": kp, "desc": desc64, "bow": bow, "sel": sel, "res": r}


def image_in_phase(a, S, local_rank, log, hd=False):
    """frame -> AKAZE + M-LDB (K9) -> the frame's BoW vector (A5a dense gray + dense-grid descriptors, A5b PCA, A5c BoF)
    -> shortlist (A5d) -> Hamming 2-NN + ratio -> F-matrix AC-RANSAC -> 2D-3D -> P3P -> pose, measured live on a map whose
    bank was EXTRACTED: --image-views views rendered from a textured plane and put through the product's own extraction
    (every keypoint a landmark), padded with views of random descriptors up to --views, the .bow vector of every real
    view computed by the same chain.  What is localised is what was extracted from the frame
    (localization.cpp:323,346-368; DenseLocalFeatureWrapper.cpp:83-183; PcaWrapper.cpp:67-89;
    BoFSpatialPyramids.cpp:108-302).  Frames are gray VGA images in host memory (PCIe inclusive)."""
    import tempfile
    import threading
    import numpy as np
    import imageworld as iw
    from sfmlocalization_amd import capi, engine, fileio
    t_build = time.perf_counter()
    knn = a.bow_knn if a.bow_knn > 0 else 100
    if hd:
        # BASELINE configs[4]'s query size.  Same field of view (8 m x 6 m from 10 m: f = 2400 px), the plane's texture at
        # 240 texels per metre so that a 1080p pixel sees what a VGA pixel sees of the 100 texel/m plane, its density set for
        # 5-6 k AKAZE keypoints per frame (BASELINE: "1080p queries", 4-6 k features); 3 x 3 places of 16 m
        W, H, focal, px_per_m, tiles, tile_px = 1920, 1080, 2400.0, 240.0, 3, 3840
        atlas_kw = {"blobs_per_tile": 4500, "rects_per_tile": 700}
        n_views, n_real = a.image_views_1080p, a.image_views_1080p
        batch, steps, n_oracle = a.batch_1080p, a.image_steps_1080p, min(4, a.image_oracle_frames)
    else:
        W, H, focal, px_per_m, tiles, tile_px = 640, 480, 800.0, 100.0, a.image_tiles, 1600
        atlas_kw = None
        n_views = a.views
        n_real = min(a.image_views, a.views) if a.image_views > 0 else a.views
        batch, steps, n_oracle = a.batch, a.image_steps, a.image_oracle_frames
    rng = np.random.Generator(np.random.PCG64(77))
    # a BoW model of the reference's shapes, trained on dense features of a few rendered frames (training is offline)
    import torch
    tdev = torch.device("cuda", local_rank)
    atlas0 = iw.make_atlas(901, 1, tile_px, tdev, **(atlas_kw or {}))
    Rs, Cs = iw.cameras(rng, 12, (0.0, 0.0), 16.0)
    train_imgs = iw.render(atlas0, px_per_m, Rs, Cs, focal, W, H)
    del atlas0
    grid = engine.dense_grid_keypoints(300)
    ak300 = S.Akaze(300, 300, 4, 4, 0.001, device=local_rank)
    feats = []
    for g in train_imgs:
        gray = capi.dense_gray(np.stack([g, g, g], 2), 300, device=local_rank)
        feats.append(ak300.compute(gray, grid)[0][:, :61].astype(np.float32))
    ak300.close()
    pca, bowm = iw.train_bow_model(np.concatenate(feats)[::3], rng)
    tmp = tempfile.TemporaryDirectory()
    bow_file, pca_file = os.path.join(tmp.name, "BOWfile.yml"), os.path.join(tmp.name, "PCAfile.yml")
    fileio.write_cv_yaml(pca_file, pca)
    fileio.write_cv_yaml(bow_file, bowm)
    dense0 = engine.DenseBow(bow_file, pca_file, device=local_rank)
    world = iw.build(S, 31, n_real, a.queries, tiles=tiles, tile_px=tile_px, px_per_m=px_per_m, focal=focal, width=W,
                     height=H, device=local_rank, n_pad_views=n_views - n_real, pad_desc_per_view=a.desc_per_view,
                     dense_bow=dense0, atlas_kw=atlas_kw, progress=log)
    m = world.m
    params = S.default_params(device=local_rank, profile=0, ransac_round=25)
    dev_map = S.Map(m.view_id, m.view_off, m.desc, params=params, view_wh=m.view_wh, kpt_xy=m.kpt_xy,
                    row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X,
                    intrinsic=m.intrinsic, bow=world.bow)
    t_build = time.perf_counter() - t_build
    log(f"image-in world ({W}x{H}): {n_real} extracted views ({world.extra['rows_real']} rows) + {n_views - n_real} padding "
        f"views, {m.n_rows} rows, built in {t_build:.1f} s")
    frames = [np.ascontiguousarray(f) for f in world.frames]
    bgrs = [np.ascontiguousarray(np.stack([f, f, f], 2)) for f in frames]
    nf = len(frames)
    nw, G = a.image_workers, max(1, min(a.image_batch, capi.GANG_MAX))
    groups, keep_alive = [], []
    for k in range(nw):
        lead = dev_map.context()
        cs = [lead] + [dev_map.context(share=lead) for _ in range(G - 1)]
        es = [S.Akaze(W, H, device=local_rank) for _ in range(G)]
        for e in es:
            e.share_stream(lead)
        ibs = [S.ImgBow.from_files(bow_file, pca_file, W, H, 1, device=local_rank) for _ in range(G)]  # gray: one channel
        if not a.image_bow_own_stream:
            for ib in ibs:
                ib.share_stream(lead)
        elif a.image_bow_worker_stream:     # the worker's BoW chains in turn on ONE stream beside the extraction's
            bow_ctx = dev_map.context(merge_only=True)
            for ib in ibs:
                ib.share_stream(bow_ctx)
            keep_alive.append(bow_ctx)
        groups.append((cs, es, ibs))
    stage_t = {"extract(K9, incl. the count's synchronisation)": 0.0, "bow_vector(A5a-c, queued)": 0.0, "query_view": 0.0,
               "shortlist+path(A5d..A12, incl. waiting for the BoW chain)": 0.0}
    lock = threading.Lock()
    lat, fps, n_ok, n_feat, err_c = [], [], [0], [0, 0], []

    def localise_frames(k, idx, record=True, staged=False):
        cs, es, ibs = groups[k]
        n = len(idx)
        t0 = time.perf_counter()
        if staged:      # everything through the host: features downloaded, query uploaded (round 2's route)
            if n == 1:
                fe = [es[0].detect_and_compute(frames[idx[0] % nf])]
            else:
                fe = S.Akaze.detect_and_compute_batch(es[:n], [frames[i % nf] for i in idx])
            t1 = time.perf_counter()
            qs = [dev_map.query(d, kp[:, :2], W, H) for kp, d in fe]
            t2 = time.perf_counter()
            for dq, i, ib, c in zip(qs, idx, ibs, cs):
                ib.compute(frames[i % nf], dq)      # queued on the worker's stream, lands in the query's BoW slot
                if a.image_bow_own_stream:
                    ib.order_before(c)
        else:           # resident: features and BoW vector stay on the device, the query is a view over them
            if a.image_bow_own_stream:   # queued first, on its own stream: runs beside the extraction below
                if n > 1:                # one gang session for the turn's frames: one launch per kernel of the chain
                    S.ImgBow.compute_batch(ibs[:n], [frames[i % nf] for i in idx])
                else:
                    ibs[0].compute(frames[idx[0] % nf], None, want_vector=False)
            ns = S.Akaze.detect_resident_batch(es[:n], [frames[i % nf] for i in idx])
            fe = [(None, range(c)) for c in ns]
            t1 = time.perf_counter()
            for i, ib, c in zip(idx, ibs, cs):
                if a.image_bow_own_stream:
                    ib.order_before(c)
                else:
                    ib.compute(frames[i % nf], None, want_vector=False)
            t2 = time.perf_counter()
            qs = [e.query_view(dev_map, c, ib.vector_dev()) for e, c, ib in zip(es, ns, ibs)]
        t3 = time.perf_counter()
        with capi.gang(cs[:n]):
            for c, dq in zip(cs, qs):
                c.begin_bow(dq, None, knn)
        ends = [c.end() for c in cs[:n]]
        t4 = time.perf_counter()
        for dq in qs:
            dq.close()
        if record:
            with lock:
                lat.extend([t4 - t0] * n)
                n_ok[0] += sum(int(e[0].ok) for e in ends)
                fps.extend((i % nf, fingerprint(*e)) for i, e in zip(idx, ends))
                n_feat[0] += sum(len(d) for _, d in fe)     # (resident route: d is range(count))
                n_feat[1] += n
                for key, dt in zip(stage_t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                    stage_t[key] += dt
                for i, e in zip(idx, ends):
                    if e[0].ok:
                        err_c.append(float(np.abs(np.array(e[0].center) - world.frame_C[i % nf]).max()))
        return fe, ends

    def run(first, count):
        def worker(k):
            for i0 in range(first + k * G, first + count, nw * G):
                localise_frames(k, list(range(i0, min(i0 + G, first + count))))
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(nw)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    # K1 on a bank of REAL statistics: the full scan of this map -- every row a K9-extracted M-LDB descriptor -- by one
    # frame's own features, one launch in flight (roofline.real_bank; the headline bank is uniform random bits, the best
    # case of the screening bound)
    real_bank = None
    if not hd and not a.no_roofline_phase:
        kp0, d0 = groups[0][1][0].detect_and_compute(frames[0])
        dq0 = dev_map.query(d0, kp0[:, :2], W, H)
        k1_ms, lane_ops, n_match, st_roof = roofline_phase(S, dev_map, dq0, int(m.n_rows), len(d0))
        dev_map.set_profile(0)
        dq0.close()
        vf = valu_floor(int(m.n_rows), len(d0), k1_ms, lane_ops, st_roof)
        real_bank = {"bank": f"{m.n_rows} rows: the image-in map, {n_real} of {m.n_views} views K9-extracted from rendered images",
                     "bank_rows": int(m.n_rows), "nq": int(len(d0)), "kernel_ms": k1_ms, "emitted_matches": n_match,
                     "ops_per_pair_issued": vf["ops_per_pair_issued"], "pairs_finished_frac": vf["pairs_finished_frac"],
                     "pairs_per_s": vf["pairs_per_s"],
                     "valu": {"achieved": vf["achieved"], "peak": vf["peak"], "unit": "T lane-ops/s", "frac": vf["frac"],
                              "floor_ms": vf["floor_ms"]},
                     "hbm_GBps": (64 * int(m.n_rows) + 64 * len(d0) + 12 * n_match) / (k1_ms * 1e-3) / 1e9}
    run(0, batch)                           # warm-up
    dev_map.sync()
    with lock:
        lat.clear(); fps.clear(); err_c.clear()
        n_ok[0] = 0; n_feat[0] = n_feat[1] = 0
        for key in stage_t:
            stage_t[key] = 0.0
    n_timed = steps * batch
    t0 = time.perf_counter()
    run(0, n_timed)
    dev_map.sync()
    dt = time.perf_counter() - t0
    fps_timed, lat_load, n_ok_timed = list(fps), list(lat), n_ok[0]
    stage_load = {k: v / max(1, n_timed) * 1e3 for k, v in stage_t.items()}
    feat_mean = n_feat[0] / max(1, n_feat[1])
    err_timed = list(err_c)
    # one frame at a time: latency, the stage split of a frame alone, and the results every timed frame is compared with
    with lock:
        lat.clear(); fps.clear()
        for key in stage_t:
            stage_t[key] = 0.0
    ref_fp = {}
    for i in range(nf):
        _, ends = localise_frames(0, [i])
        ref_fp[i] = fingerprint(*ends[0])
    lat_single = list(lat)
    stage_single = {k: v / nf * 1e3 for k, v in stage_t.items()}
    # the BoW chain's own split (three synchronous calls and the host glue between them)
    dense = dense0
    ib0 = groups[0][2][0]
    t1 = time.perf_counter()
    for i in range(min(nf, 32)):
        ib0.compute(frames[i], None, want_vector=True)
    bow_resident_ms = (time.perf_counter() - t1) / min(nf, 32) * 1e3
    bow_split = {"dense_gray(A5a resize+gray+minmax)": 0.0, "dense_descriptors(A5a AKAZE compute, 10000 kpts)": 0.0,
                 "pca+bof(A5b,A5c)": 0.0}
    for i in range(min(nf, 32)):
        t1 = time.perf_counter()
        gray = capi.dense_gray(bgrs[i], dense.size, device=local_rank)
        t2 = time.perf_counter()
        dd, _ = dense.akaze.compute(gray, dense.grid)
        ff = dd[:, :61].astype(np.float32)
        t3 = time.perf_counter()
        dense.bof.compute(ff, dense.grid[:, :2].copy())
        t4 = time.perf_counter()
        for key, d in zip(bow_split, (t2 - t1, t3 - t2, t4 - t3)):
            bow_split[key] += d / min(nf, 32) * 1e3
    n_same = sum(1 for i, f in fps_timed if ref_fp.get(i) == f)
    # the path's own stages for a frame alone (HIP-event brackets, the reference's `times` buckets) and what the
    # frames look like to the path: views with >= 16 putative matches, views that pass the F-matrix filter
    dev_map.set_profile(1)
    path_split = {"selectBow": 0.0, "putMatch": 0.0, "geoMatch": 0.0, "PnP": 0.0, "others": 0.0}
    n_put_v = n_geo_v = n_23 = n_inl = 0
    n_ps = min(nf, 32)
    for i in range(n_ps):
        _, ends = localise_frames(0, [i], record=False)
        ss = ends[0][0].stage_seconds
        for key, v in zip(path_split, (ss[1], ss[3], ss[4], ss[5], ss[6])):
            path_split[key] += v / n_ps * 1e3
        n_put_v += ends[0][0].n_putative_views
        n_geo_v += ends[0][0].n_geometric_views
        n_23 += ends[0][0].n_matches_2d3d
        n_inl += ends[0][0].n_inliers
    dev_map.set_profile(0)
    path_shape = {"views_with_16_or_more_putative_matches": n_put_v / n_ps, "views_passing_the_F_matrix_filter": n_geo_v / n_ps,
                  "correspondences_2d3d": n_23 / n_ps, "inliers": n_inl / n_ps}
    # a sample of the frames against the oracle, end to end
    checked, agree = 0, 0
    oracle_note = []
    for i in range(0, nf, max(1, nf // max(1, n_oracle)))[:n_oracle]:
        fe, ends = localise_frames(0, [i], record=False, staged=True)
        if capi.result_fingerprint(*ends[0]) != ref_fp[i]:
            _, again = localise_frames(0, [i], record=False)
            pa, pb = ends[0][0], again[0][0]
            raise SystemExit(f"image-in: frame {i}: the staged route and the resident route give different results: staged "
                             f"(ok {pa.ok}, 2d3d {pa.n_matches_2d3d}, inliers {pa.n_inliers}, iterations {pa.iterations}, views "
                             f"{pa.n_putative_views}/{pa.n_geometric_views}, nfa {pa.nfa}) resident now (ok {pb.ok}, 2d3d "
                             f"{pb.n_matches_2d3d}, inliers {pb.n_inliers}, iterations {pb.iterations}, views "
                             f"{pb.n_putative_views}/{pb.n_geometric_views}, nfa {pb.nfa}); resident now == resident before: "
                             f"{capi.result_fingerprint(*again[0]) == ref_fp[i]}")
        o = oracle_image_chain(world, frames[i], (pca, bowm), knn)
        kp, d = fe[0]
        pose, pq, pl = ends[0]
        same = (len(kp) == len(o["kp"]) and np.array_equal(kp.view(np.uint32), o["kp"].astype(np.float32).view(np.uint32))
                and np.array_equal(d, o["desc"]))
        same_bow = np.array_equal(ib0.compute(frames[i], None, want_vector=True).view(np.uint64), o["bow"].view(np.uint64))
        same_sel = np.array_equal(np.sort(dev_map.bow_select(o["bow"].astype(np.float32), knn)), o["sel"])
        r = o["res"]
        same_pose = bool(pose.ok) == bool(r["ok"]) and (not r["ok"] or (
            np.array_equal(pq, r["pair_qfeat"]) and np.array_equal(pl, r["pair_landmark"])
            and np.array_equal(np.array(pose.P).view(np.uint64), np.ascontiguousarray(r["P"]).ravel().view(np.uint64))))
        checked += 1
        agree += int(same and same_bow and same_sel and same_pose)
        oracle_note.append({"frame": i, "features": int(len(kp)), "features_equal": bool(same), "bow_equal": bool(same_bow),
                            "shortlist_equal": bool(same_sel), "pose_and_inliers_equal": bool(same_pose),
                            "localised": bool(r["ok"])})
    out = {
        "metric": "query images localized/sec, image in", "value": n_timed / dt, "unit": "images/s", "frame": f"{W}x{H}",
        "frames_timed": n_timed, "frames_localised": f"{n_ok_timed}/{n_timed}",
        "identical_to_single_flight": f"{n_same}/{len(fps_timed)}",
        "oracle_end_to_end": {"frames_checked": checked, "frames_identical": agree, "detail": oracle_note},
        "workload": f"{W}x{H} gray frames in host memory -> AKAZE + M-LDB (K9, outputs resident: only the keypoint count is read "
                    f"back) -> BoW vector of the frame (sfmloc_imgbow: dense gray, "
                    f"10 000 dense-grid descriptors, PCA-32 / eigenvalue, BoF 5 x 100) -> shortlist of {knn} of {m.n_views} "
                    f"views -> whole path -> pose; map: {n_real} views rendered from a textured plane and EXTRACTED by K9 "
                    f"({world.extra['rows_real']} descriptors, one landmark each) + {m.n_views - n_real} padding views of random "
                    f"descriptors = {m.n_rows} rows; {nw} workers x {G} frames per turn",
        "features_per_frame": feat_mean,
        "latency_ms": {"p50": float(np.percentile(lat_single, 50) * 1e3), "p95": float(np.percentile(lat_single, 95) * 1e3),
                       "mode": "one frame in flight", "p50_at_throughput": float(np.percentile(lat_load, 50) * 1e3)},
        "stage_ms_one_frame_alone": stage_single, "stage_ms_per_frame_under_load": stage_load,
        "bow_vector_ms": {"resident_chain_alone(sfmloc_imgbow, incl. upload + D2H of the vector)": bow_resident_ms,
                          "three_staged_calls_alone(dense_gray, akaze_compute, bof_compute)": bow_split},
        "path_stage_ms_one_frame_alone": path_split, "path_shape_per_frame": path_shape,
        "centre_error_m": {"median": float(np.median(err_timed)) if err_timed else None,
                           "max": float(np.max(err_timed)) if err_timed else None},
        "world_build_s": round(t_build, 1),
    }
    if real_bank is not None:
        out["k1_full_scan_of_this_map"] = real_bank
    for cs, es, ibs in groups:
        for ib in ibs:
            ib.close()
        for e in es:
            e.close()
        for c in reversed(cs):
            c.close()
    for c in keep_alive:
        c.close()
    dense0.close()
    dev_map.close()
    tmp.cleanup()
    return out


def plan(a, world, replicas):
    """The run's shape from its arguments: (shortlist, forced, sharded_mode, gang, nctx, hardware queues wanted); sets
    a.threads.  replicas: every rank is a single-GPU run on the whole map."""
    shortlist = a.bow_knn > 0
    # measured (profiles/r02_inflight_sweep.txt): with the shortlist 12 contexts driven by 4 host threads; full scans fill
    # the chip with 4
    forced = os.environ.get("SFMLOC_BENCH_FORCE_SHARDED") == "1"
    sharded_mode = (world > 1 or forced) and not replicas
    # (the sharded path keeps two slots of contexts; one query per launch: 8 per slot -- with 12 the 24 contexts and the
    # collective's stream no longer get a hardware queue each and the rate drops by 40 %,
    # profiles/r02_sharded_inflight_sweep.txt)
    # (images in: a worker thread is ONE stream -- its extractors and contexts queue on the first context's; 4 workers
    # are the measured optimum, profiles/r02_image_in_sweep.txt)
    # (N ranks: a rank scans 1/N of the map per query but issues every query's launches, and a device serves few
    # hardware queues well -- 16 queries per launch on 2 streams per slot instead of 8 streams with a query each:
    # tools/rank_emulation.py, profiles/r02_rank_emulation.jsonl)
    # (round 4: 32 queries per launch and 96 contexts per slot.  A rank's scan of 16 queries' shortlisted views is 784
    # workgroups for 1 024 places and the next session's scan waits behind this one's chain of latency-bound launches: more
    # queries per launch, three sessions per slot.  K3's argument list was 280 bytes and a gang launch of it held 14 members
    # -- a session of 32 went out as 14 + 14 + 4, three launches of ~250 us one after the other --: the list's constant part
    # now lives on the device (FFilterStatic) and a launch holds 32.  One emulated rank of 2 / 4 / 8: 6.4 / 14.0 / 21.3 k
    # queries/s (16 per launch on 32 contexts: 6.1 / 10.1 / 14.5 k), profiles/r04_rank_emulation.jsonl)
    gang = a.gang if a.gang > 0 else (32 if (world > 1 and not replicas) else int(os.environ.get("SFMLOC_GANG", "1")))
    gang = gang if sharded_mode else 1
    nctx = a.in_flight if a.in_flight > 0 else (4 if a.from_images else
                                                (((3 * gang if gang > 1 else 8) if sharded_mode else 20) if shortlist else 4))
    if a.threads == 0 and shortlist and nctx >= 8 and not a.from_images and not sharded_mode:
        a.threads = 4
    # one HW queue per stream that carries work (sfmlocalization_amd/_lib.py): a context each without gangs, a gang's
    # first context with them (two slots + the two stage-2 groups)
    # (+2: the map's own stream and torch's; measured: 8 contexts on 8 queues lose 20 % to two streams sharing one)
    n_streams = (2 * -(-nctx // gang) + 2) if gang > 1 else (2 * nctx if sharded_mode else nctx)
    hwq = min(24, max(8, n_streams + 2))
    # (a sharded rank's gang sessions: 16 queues whatever the count of streams says -- one emulated rank of 2 / 4 / 8 with
    # 8, 12, 16, 24 queues: 5.5 / 6.0 / 6.0 / 6.0, 11.3 / 12.7 / 12.9 / 11.3 and 18.4 / 18.4 / 18.3 / 16.1 k queries/s)
    if sharded_mode and gang > 1:
        hwq = 16
    return shortlist, forced, sharded_mode, gang, nctx, hwq



def measure(a, rank, world, local_rank, dist, torch, replicas):
    """One whole measurement in the mode `replicas` says; -> the JSON object (rank 0) or None."""
    import numpy as np
    shortlist, forced, sharded_mode, gang, nctx, _ = plan(a, world, replicas)
    log = lambda msg: print("[bench] " + msg, file=sys.stderr, flush=True)  # noqa: E731
    import sfmlocalization_amd as S
    import synthdata as synth

    if a.image_in_only:
        out = {"image_in": image_in_phase(a, S, local_rank, log)}
        if not a.no_image_in_1080p:
            out["image_in_1080p"] = image_in_phase(a, S, local_rank, log, hd=True)
        return out
    # every rank builds the same seeded map and keeps its shard of views (contiguous view ranges)
    t_gen = time.perf_counter()
    m = synth.make_map(2, n_views=a.views, desc_per_view=a.desc_per_view)
    queries = [synth.make_query(m, 1000 + i, n_feat=a.nq) for i in range(a.queries)]
    bow = qbow = None
    if shortlist:
        bow, qbow = synth_bow(m, queries)
    t_gen = time.perf_counter() - t_gen
    # (replicas: the whole map on every rank, and of every batch the queries i with i mod world == rank)
    v0 = 0 if replicas else (a.views * rank) // world
    v1 = a.views if replicas else (a.views * (rank + 1)) // world
    stride, phase = (world, rank) if replicas else (1, 0)
    r0, r1 = int(m.view_off[v0]), int(m.view_off[v1])
    params = S.default_params(device=local_rank, profile=0, ransac_round=25)
    # diagnosis only (what each latency-bound stage costs the throughput); a line produced with these set is not the metric
    diag = {k: int(os.environ[k]) for k in ("SFMLOC_DIAG_RANSAC_ROUND", "SFMLOC_DIAG_P3P_ITER", "SFMLOC_DIAG_STOP_AFTER")
            if k in os.environ}
    if "SFMLOC_DIAG_RANSAC_ROUND" in diag:
        params.ransac_round = diag["SFMLOC_DIAG_RANSAC_ROUND"]
    if "SFMLOC_DIAG_P3P_ITER" in diag:
        params.p3p_max_iteration = diag["SFMLOC_DIAG_P3P_ITER"]
    dev_map = S.Map(m.view_id[v0:v1], m.view_off[v0:v1 + 1] - m.view_off[v0], m.desc[r0:r1], params=params,
                    view_wh=m.view_wh[v0:v1], kpt_xy=m.kpt_xy[r0:r1], row_landmark=m.row_landmark[r0:r1],
                    landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
                    bow=None if bow is None else bow[v0:v1])
    dqs = [dev_map.query(q.desc, q.kpt_xy, q.width, q.height) for q in queries]
    if shortlist:
        for dq, qb in zip(dqs, qbow):
            dq.set_bow(qb)          # the query's BoW vector is an input: resident before the timed region
    lat = []
    n_ok = [0]
    fps = []            # (query index, fingerprint) of every query localised since the last clear
    sharded = None
    xchg = {}
    if sharded_mode:
        from sfmlocalization_amd import dist as D
        comp = D.HipShardCompute(dev_map, n_contexts=nctx, device=torch.device("cuda", local_rank), gang=gang)
        sharded = D.ShardedLocalizer(comp, rank=rank, world=world, always_gather=forced and dist is not None,
                                     n_views_global=a.views)
    ctxs = [dev_map.context() for _ in range(nctx)] if sharded is None else []
    # queries per context STREAM (--queries-per-stream G > 1): every stream carries G contexts that take a query each through
    # one gang session per turn -- G x nctx queries in flight on nctx hardware queues
    G = max(1, int(getattr(a, "queries_per_stream", 1) or 1)) if sharded is None and not a.from_images else 1
    members = [[c] + [dev_map.context(share=c) for _ in range(G - 1)] for c in ctxs]
    t_begin = [0.0] * nctx
    busy = [False] * nctx

    q_of = [0] * nctx

    def finish(k):
        pose, pq, pl = ctxs[k].end()
        lat.append(time.perf_counter() - t_begin[k])
        n_ok[0] += int(pose.ok)
        fps.append((q_of[k], fingerprint(pose, pq, pl)))
        busy[k] = False

    def begin(k, i):
        dq = dqs[i % len(dqs)]
        if shortlist:   # configs[2]: shortlist + path in one asynchronous call, the shortlist stays on the device
            ctxs[k].begin_bow(dq, None, a.bow_knn)
        else:
            ctxs[k].begin(dq)

    def run_sharded(first, count):
        """`count` queries starting at query `first`, in batches of a.batch through the two-slot pipeline: stage 1 of
        batch b+1 is queued before batch b's all-gather and P3P stage."""
        idx = list(range(first, first + count))
        batches = [idx[k:k + a.batch] for k in range(0, len(idx), a.batch)]
        t_mark = [time.perf_counter(), time.perf_counter()]   # batch b was enqueued when batch b-2 was yielded
        stream = sharded.localize_stream([[dqs[i % len(dqs)] for i in b] for b in batches], gather_results=False,
                                         bow_knn=a.bow_knn if shortlist else 0)
        for b, res in zip(batches, stream):
            now = time.perf_counter()
            lat.extend([now - t_mark[0]] * len(b))          # a query's latency in batch mode = its batch's wall time
            t_mark = [t_mark[1], now]
            n_ok[0] += sum(int(r["ok"]) for r in res.values())
            fps.extend((b[j] % len(dqs), r["fingerprint"]) for j, r in res.items())

    def run_threads(first, count):
        import threading
        nthr = min(a.threads, nctx)
        lock = threading.Lock()

        def worker_gang(t):
            # G queries per turn of a stream: one gang session (one launch per kernel for the G members)
            from sfmlocalization_amd import capi
            mine = list(range(t, nctx, nthr))
            tb = {k: 0.0 for k in mine}
            cur = {k: [] for k in mine}
            lat_l, ok_l, fp_l = [], 0, []

            def fin(k):
                ok = 0
                for c, i in zip(members[k], cur[k]):
                    pose, pq, pl = c.end()
                    lat_l.append(time.perf_counter() - tb[k])
                    fp_l.append((i % len(dqs), fingerprint(pose, pq, pl)))
                    ok += int(pose.ok)
                cur[k] = []
                return ok

            todo = list(range(first + phase + t * stride, first + count, nthr * stride))
            for n, j0 in enumerate(range(0, len(todo), G)):
                k = mine[n % len(mine)]
                if cur[k]:
                    ok_l += fin(k)
                idx = todo[j0:j0 + G]
                tb[k] = time.perf_counter()
                with capi.gang(members[k][:len(idx)]):
                    for c, i in zip(members[k], idx):
                        dq = dqs[i % len(dqs)]
                        if shortlist:
                            c.begin_bow(dq, None, a.bow_knn)
                        else:
                            c.begin(dq)
                cur[k] = idx
            for k in mine:
                if cur[k]:
                    ok_l += fin(k)
            with lock:
                lat.extend(lat_l)
                n_ok[0] += ok_l
                fps.extend(fp_l)

        def worker(t):
            mine = list(range(t, nctx, nthr))          # this thread's contexts
            tb = {k: 0.0 for k in mine}
            bz = {k: False for k in mine}
            qi = {k: 0 for k in mine}
            lat_l, ok_l, fp_l = [], 0, []

            def fin(k):
                pose, pq, pl = ctxs[k].end()
                lat_l.append(time.perf_counter() - tb[k])
                fp_l.append((qi[k], fingerprint(pose, pq, pl)))
                return int(pose.ok)

            for n, i in enumerate(range(first + phase + t * stride, first + count, nthr * stride)):
                k = mine[n % len(mine)]
                if bz[k]:
                    ok_l += fin(k)
                tb[k] = time.perf_counter()
                begin(k, i)
                qi[k] = i % len(dqs)
                bz[k] = True
            for k in mine:
                if bz[k]:
                    ok_l += fin(k)
            with lock:
                lat.extend(lat_l)
                n_ok[0] += ok_l
                fps.extend(fp_l)

        ts = [threading.Thread(target=worker_gang if G > 1 else worker, args=(t,)) for t in range(nthr)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    def run(first, count):
        if sharded is not None:
            run_sharded(first, count)
            return
        if a.threads > 1:
            run_threads(first, count)
            return
        for n, i in enumerate(range(first + phase, first + count, stride)):   # up to `nctx` queries overlap on the GPU
            k = n % nctx
            if busy[k]:
                finish(k)
            t_begin[k] = time.perf_counter()
            begin(k, i)
            q_of[k] = i % len(dqs)
            busy[k] = True

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
        from sfmlocalization_amd import capi
        imgs = [synth.texture_image(900 + k, 480, 640) for k in range(4)]
        G = max(1, min(a.image_batch, capi.GANG_MAX))
        # a worker = ONE stream: its G extractors and G contexts all queue on the first context's (the frames of a turn
        # are extracted together and localised in one gang session)
        groups = []
        for k in range(nctx):
            cs = [ctxs[k]] + [dev_map.context(share=ctxs[k]) for _ in range(G - 1)]
            es = [S.Akaze(640, 480, device=local_rank) for _ in range(G)]
            if os.environ.get("SFMLOC_BENCH_SHARE_STREAM", "1") != "0":
                for e in es:
                    e.share_stream(ctxs[k])
            groups.append((cs, es))
        extractors = [e for _, es in groups for e in es]
        n_feat = [0]
        lock = threading.Lock()

        def worker(k, first, count):
            cs, es = groups[k]
            mine = list(range(first + k * G, first + count, nctx * G))
            for i0 in mine:
                idx = [i for i in range(i0, min(i0 + G, first + count))]
                t1 = time.perf_counter()
                if len(idx) == 1:
                    feats = [es[0].detect_and_compute(imgs[idx[0] % len(imgs)])]
                else:
                    feats = S.Akaze.detect_and_compute_batch(es[:len(idx)], [imgs[i % len(imgs)] for i in idx])
                with capi.gang(cs[:len(idx)]):
                    for c, i in zip(cs, idx):
                        dq = dqs[i % len(dqs)]
                        if shortlist:
                            c.begin_bow(dq, None, a.bow_knn)
                        else:
                            c.begin(dq)
                ends = [c.end() for c in cs[:len(idx)]]
                oks = [int(e[0].ok) for e in ends]
                with lock:
                    lat.extend([time.perf_counter() - t1] * len(idx))
                    n_ok[0] += sum(oks)
                    fps.extend((i % len(dqs), fingerprint(*e)) for i, e in zip(idx, ends))
                    n_feat[0] = len(feats[0][0])

        def run_images(first, count):
            ts = [threading.Thread(target=worker, args=(k, first, count)) for k in range(nctx)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()

        run = run_images  # noqa: F811
        img_mode = {"extractors": extractors, "n_feat": n_feat, "frames_per_turn": G,
                    "member_contexts": [c for cs, _ in groups for c in cs[1:]]}

    n_timed = a.steps * a.batch
    run(0, a.warmup * a.batch)
    fence()
    dev_map.stats_reset()
    if sharded is not None:
        sharded.reset_counters()
    lat.clear()
    fps.clear()
    n_ok[0] = 0
    fence()
    t0 = time.perf_counter()
    run(0, n_timed)
    fence()
    dt = time.perf_counter() - t0
    fps_timed = list(fps)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lat_throughput = list(lat)
    n_ok_timed = n_ok[0]
    if world > 1:
        ok_t = torch.tensor([n_ok_timed], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok_t)
        n_ok_timed = int(ok_t.item())
    if sharded is not None:
        xchg = sharded.counters()

    # ---- outside the timed region -------------------------------------------------------------------------------
    # (a) stage split and the in-path K1 launches: the same queries again with every stage bracketed by HIP events
    dev_map.set_profile(1)
    dev_map.stats_reset()
    n_prof = min(n_timed, 4 * a.batch)
    lat.clear()
    run(0, n_prof)
    fence()
    st = dev_map.stats()
    # (b) per-query latency proper: one query in flight (with several in flight a query's wall time is mostly queueing)
    #     and, from the same pass, what every timed query is compared with: each distinct query's result when it is
    #     alone on the GPU.  A query localised under load must give the same bits (status, P, K, R, t, inlier pairs).
    lat_single = []
    ref_fp = {}
    if sharded is None:
        dev_map.set_profile(0)
        dev_map.sync()
        for i in range(max(min(n_timed, 64), len(dqs))):
            t1 = time.perf_counter()
            begin(0, i)
            pose, pq, pl = ctxs[0].end()
            lat_single.append(time.perf_counter() - t1)
            ref_fp.setdefault(i % len(dqs), fingerprint(pose, pq, pl))
        # the reference's `times` buckets of one query alone, from HIP events (profile = 1)
        dev_map.set_profile(1)
        begin(0, 0)
        pose = ctxs[0].end()[0]
        stage_seconds = [float(x) for x in pose.stage_seconds]
        dev_map.set_profile(0)
    else:
        for i in range(len(dqs)):     # a batch of one query through the same sharded path, every rank gets the result
            res = sharded.localize_batch([dqs[i]], gather_results=True, bow_knn=a.bow_knn if shortlist else 0)
            ref_fp[i] = res[0]["fingerprint"]
    n_same = sum(1 for i, f in fps_timed if ref_fp.get(i) == f)
    n_cmp = len(fps_timed)
    if world > 1:
        c_t = torch.tensor([n_same, n_cmp], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(c_t)
        n_same, n_cmp = int(c_t[0].item()), int(c_t[1].item())
    # (c) the roofline kernel: full-bank scan of this rank's bank, one launch in flight
    roof = sweep = None
    rows_rank = r1 - r0
    if not a.no_roofline_phase:
        k1_ms, lane_ops, n_match, st_roof = roofline_phase(S, dev_map, dqs[0], rows_rank, a.nq)
        alg_bytes = 64 * rows_rank + 64 * a.nq + 12 * n_match
        pairs = rows_rank * a.nq
        roof = (k1_ms, lane_ops, n_match, alg_bytes, pairs, st_roof)
        sweep = nq_sweep_phase(S, dev_map, rows_rank)
    dev_map.set_profile(0)
    real_stats = None
    if rank == 0 and world == 1 and not a.no_roofline_phase and not a.no_real_stats:
        real_stats = real_statistics_phase(S, synth, local_rank)
    if rank == 0:
        sel_rows = None
        if shortlist and sharded is None:
            sel0 = dev_map.bow_select(qbow[0], a.bow_knn)
            sel_rows = int(sum(int(m.view_off[v + 1] - m.view_off[v]) for v in sel0))
        out = {
            "metric": "query images localized/sec",
            "value": n_timed / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32 popcount (Hamming); f64 geometry",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{2 if world == 1 else 3}]: " if (shortlist and a.views == 10000) else "")
                                   + f"{a.views}-image / {m.n_rows}-descriptor synthetic map, {a.nq} feats/query, "
                                   + (f"BoW shortlist of {a.bow_knn} views (exact L2 over the {a.views} x 500 .bow matrix) then "
                                      if shortlist else "")
                                   + "the whole per-query path: brute-force Hamming 2-NN + Lowe ratio -> >=16 filter -> "
                                   "F-matrix AC-RANSAC (25 rounds) -> 2D-3D set -> P3P AC-RANSAC (4096) -> pose; "
                                   f"one step = a batch of {a.batch} queries, {nctx} in flight per GPU",
                       "views": a.views, "rows": int(m.n_rows), "nq": a.nq, "bow_knn": a.bow_knn, "batch": a.batch,
                       "queries_per_step": a.batch, "queries_timed": n_timed, "in_flight_per_gpu": nctx,
                       "host_threads": max(1, a.threads) if sharded is None else 1,
                       **({"queries_per_launch": gang} if sharded is not None else {}),
                       "parallelism": ((f"replicas x{world}: every rank holds the whole bank, query i of a batch on rank i mod "
                                        f"{world}, no collective") if (replicas and world > 1) else
                                       (f"bank + .bow sharded by view x{world}; per {a.batch}-query batch one all-gather of "
                                        "per-shard k-best BoW keys and ONE all-gather of candidate parts (RCCL), P3P of "
                                        "query i on rank i mod N") if world > 1 else "1 GPU, whole bank"),
                       "queries_localised": f"{n_ok_timed}/{n_timed}",
                       "map_generation_s": round(t_gen, 1), **({"DIAGNOSTIC_OVERRIDES_NOT_THE_METRIC": diag} if diag else {})},
            "identical_to_single_flight": f"{n_same}/{n_cmp}",
            "latency_ms": {"p50": float(np.percentile(lat_single or lat_throughput, 50) * 1e3),
                           "p95": float(np.percentile(lat_single or lat_throughput, 95) * 1e3),
                           "mode": "one query in flight" if lat_single else f"{a.batch}-query batches",
                           "p50_at_throughput": float(np.percentile(lat_throughput, 50) * 1e3),
                           "p95_at_throughput": float(np.percentile(lat_throughput, 95) * 1e3)},
            "stage_ms": {"note": f"HIP-event brackets per stage over {n_prof} queries re-run after the timed region with "
                                 f"{nctx} in flight (a bracket contains the other queries' work)",
                         "selectBow(K8)": st.total_ms[5] / max(1, n_prof),
                         "putMatch(K1+K2)": (st.total_ms[0] + st.total_ms[1]) / max(1, n_prof),
                         "geoMatch(K3)": st.total_ms[2] / max(1, n_prof), "matchSet(K4)": st.total_ms[3] / max(1, n_prof),
                         "PnP(K5)": st.total_ms[4] / max(1, n_prof)},
        }
        if lat_single:
            out["latency_ms"]["stage_seconds_last_query"] = dict(zip(
                ["selectBeacon", "selectBow", "extFeat", "putMatch", "geoMatch", "PnP", "others"], stage_seconds))
        if xchg:
            out["exchange"] = xchg
        if roof is not None:
            k1_ms, lane_ops, n_match, alg_bytes, pairs, st_roof = roof
            achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
            valu = lane_ops / (k1_ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(world == 1 and (a.views, a.desc_per_view, a.nq) == (10000, 2000, 2000))
            vf = valu_floor(rows_rank, a.nq, k1_ms, lane_ops, st_roof)
            floor_ms, vop2, vop3 = vf["floor_ms"], vf["vop2"], vf["vop3"]
            sq_insts = None
            if traffic is not None:
                with open(PMC_SUMMARY) as fh:
                    sq_insts = json.load(fh).get("k_hamming_screen<8, 10, 1>", {}).get("SQ_INSTS_VALU", {}).get("mean_per_dispatch")
            # The bound that applies first: the kernel is VALU-bound 200 : 1, so the
            # top-level keys are the VALU figures; the HBM figures the metric names sit under `hbm`.
            out["roofline"] = {
                "bound": "valu", "achieved": valu, "peak": vf["peak"], "unit": "T lane-ops/s", "frac": vf["frac"],
                "traffic": traffic,
                "traffic_unit": "HBM bytes/launch (PMC FETCH_SIZE*2+WRITE_SIZE of separate rocprofv3 --pmc passes; read from "
                                "the committed summary, PMC cannot run inside this process)",
                "traffic_source": traffic_src,
                "kernel": "k_hamming_screen (+k_hamming_rows): full-bank scan of this map, one launch in flight, "
                          "after the timed region",
                "kernel_ms": k1_ms, "launches": int(st_roof.launches[0]), "algorithmic_bytes": alg_bytes,
                "bank_rows": rows_rank, "nq": a.nq, "emitted_matches": n_match,
                "note": "SURVEY F7: at N_q=2000 the kernel is VALU-bound (intensity N_q/2 lane-ops per bank byte against "
                        "a machine balance of ~6).  achieved = lane-ops the kernel counted itself / kernel time; peak = the "
                        "harmonic combination of the measured issue rates of its two instruction classes for ITS class "
                        "counts (a floor no schedule of these instructions can beat); frac = floor time / measured time",
                "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes,
                        "note": "the metric as BASELINE.json words it (Hamming-match achieved HBM GB/s): algorithmic bytes "
                                "(64 D + 64 N_q + 12 matches) / kernel time against 8 TB/s; <1 % by arithmetic at N_q = 2000"},
                "valu": {"achieved": valu, "peak": vf["peak"], "unit": "T lane-ops/s",
                         "frac": vf["frac"],
                         "peak_source": "harmonic combination of the measured issue rates of the kernel's two instruction "
                                        f"classes (VOP2-rate {VOP2_TOPS} T, VOP3-only {VOP3_TOPS} T lane-ops/s at 8 waves per "
                                        "SIMD, profiles/r02_valu_rates.jsonl) for ITS class counts: floor = vop2 / R2 + "
                                        "vop3 / R3; frac = floor time / measured time",
                         "floor_ms": floor_ms, "vop2_rate_lane_ops_per_launch": vop2, "vop3_only_lane_ops_per_launch": vop3,
                         "counted_lane_ops_per_launch": lane_ops,
                         "SQ_INSTS_VALU_x64_per_launch(PMC pass, incl. addressing and loop instructions)":
                             None if sq_insts is None else sq_insts * 64,
                         "frac_of_1to1_mix_ceiling_41.3T(round 2's figure)": valu / VALU_PEAK_TOPS,
                         "frac_of_nominal_single_issue_rate_78.6T(256 CU x 4 SIMD x 32 lanes x 2.4 GHz)": valu / 78.6,
                         "ops_per_pair_exact": OPS_PER_PAIR, "ops_per_pair_issued": vf["ops_per_pair_issued"],
                         "pairs_per_s": vf["pairs_per_s"],
                         "pairs_finished_frac": vf["pairs_finished_frac"],
                         "rows_flagged_per_scan": vf["rows_flagged_per_scan"]}}
            if sweep:
                best = max(sweep, key=lambda r: r["bank_GBps"])
                out["roofline"]["hbm"]["hbm_bound_regime"] = {
                    "measured": "live, this run: N_q query rows per full-bank pass, one launch in flight",
                    "sweep": sweep, "nq": best["nq"], "bank_rows": rows_rank, "achieved": best["bank_GBps"],
                    "unit": "GB/s of bank bytes", "peak": HBM_PEAK_GBS, "frac": best["bank_GBps"] / HBM_PEAK_GBS,
                    "frac_of_measured_copy": best["bank_GBps"] / HBM_COPY_GBS,
                    "note": "plus 12.5 % partial-result writes; 6.29 TB/s is the guide's measured float4 copy"}
            if real_stats is not None:
                # the synthetic bank is uniform random bits -- the best case of the screening bound; the same two
                # kernels on M-LDB-like descriptors
                out["roofline"]["real_statistics"] = real_stats
                out["roofline"]["real_statistics"]["note"] = (
                    "uniform random bits (the headline bank) are the best case of the screening bound; this is the same "
                    "kernel pair on K9-extracted M-LDB descriptors")
            # the K1 launches inside the path (short block list of the shortlist): event brackets with nctx in flight
            if st.launches[0]:
                out["roofline"]["in_path_scan"] = {
                    "rows_per_query": sel_rows, "event_bracket_ms": st.total_ms[0] / st.launches[0],
                    "note": "side note only: brackets of overlapping launches contain queue wait"}
        if img_mode is not None:
            out["config"]["workload"] += ("; image-in mode: each query first runs AKAZE + M-LDB extraction of a 640x480 "
                                          f"synthetic image on the GPU ({img_mode['n_feat'][0]} keypoints), "
                                          f"{img_mode['frames_per_turn']} frames per worker turn")
        if not a.no_cpu_baseline and world == 1:   # the CPU leg is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = (cpu_baseline_shortlist(m, queries, bow, qbow, a.bow_knn, a.cpu_seconds)
                                   if shortlist else cpu_baseline(m, queries, a.cpu_seconds))
    if img_mode is not None:
        for e in img_mode["extractors"]:
            e.close()
        for c in img_mode["member_contexts"]:
            c.close()
    for ms in members:
        for c in reversed(ms[1:]):
            c.close()
    for c in ctxs:
        c.close()
    if sharded is not None:
        sharded.compute.close()
    for dq in dqs:
        dq.close()
    dev_map.close()
    if rank == 0:
        # the image in front of the path, on a map of its own (the headline's map and contexts are gone by now)
        if world == 1 and not a.no_image_in and not a.from_images and shortlist:
            del m, queries, bow, qbow
            out["image_in"] = image_in_phase(a, S, local_rank, log)
            rb = out["image_in"].pop("k1_full_scan_of_this_map", None)
            if rb is not None and "roofline" in out:
                out["roofline"]["real_bank"] = rb
            if not a.no_image_in_1080p:
                out["image_in_1080p"] = image_in_phase(a, S, local_rank, log, hd=True)
        return out
    return None


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))   # before torch / HIP are touched in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (it launches "
              f"its own ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus", file=sys.stderr)
        sys.exit(2)
    _, forced, _, _, _, hwq = plan(argparse.Namespace(**vars(a)), world, a.replicas)
    if world > 1:
        hwq = 24     # (both legs of a multi-GPU run)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(hwq))

    import numpy as np
    sys.path.insert(0, ROOT)
    import torch
    dist = None
    if world > 1 or (forced and "RANK" in os.environ):
        import torch.distributed as dist_
        dist = dist_
        # SFMLOC_BENCH_BACKEND=gloo: rehearse N ranks on fewer GPUs (ranks share devices, the exchange goes through the
        # host); the measured configuration is always nccl = RCCL, one rank per GPU
        backend = os.environ.get("SFMLOC_BENCH_BACKEND", "nccl")
        # a collective that cannot complete (a wedged link, a rank that died) must end the run with a line, not sit for
        # the default ten minutes: 120 s per collective, and the watchdog below for anything that never returns at all
        from datetime import timedelta
        coll_timeout = timedelta(seconds=float(os.environ.get("SFMLOC_BENCH_COLLECTIVE_TIMEOUT_S", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=coll_timeout)
        else:
            dist.init_process_group(backend, timeout=coll_timeout)
            local_rank = local_rank % max(1, torch.cuda.device_count())
        assert dist.get_world_size() == a.gpus, "process group size differs from --gpus"
    torch.cuda.set_device(local_rank)

    watchdog = None
    if dist is not None:
        watchdog = Watchdog(float(os.environ.get("SFMLOC_BENCH_WATCHDOG_S", "900")), rank, a)
    try:
        out = run_all(a, rank, world, local_rank, dist, torch)
    except BaseException as e:  # noqa: BLE001 -- incl. DistBackendError / a collective's timeout
        if dist is None or isinstance(e, (SystemExit, KeyboardInterrupt)):
            raise
        fail(a, rank, f"{type(e).__name__}: {e}", 3)
    if watchdog is not None:
        watchdog.stop()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def fail(a, rank, why, code):
    """A multi-rank run that cannot finish: one JSON line from rank 0 (so the driver records WHY instead of a silent
    timeout), the traceback on stderr, and the process ends at once -- no destructor of a wedged process group is run,
    and nothing is re-exec'ed (a process that has touched the GPU must not be replaced)."""
    import traceback
    traceback.print_exc()
    print(f"[bench] rank {rank}: {why}", file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps({"metric": "query images localized/sec", "value": None, "unit": "queries/s", "n_gpus": a.gpus,
                          "steps": a.steps, "warmup": a.warmup, "error": why[:2000]}), flush=True)
    sys.stdout.flush()
    os._exit(code)


class Watchdog:
    """Ends a multi-rank run that makes no progress at all (a hang INSIDE a call, which no exception reports): every
    phase of measure() completes well inside the limit on a healthy node."""

    def __init__(self, limit_s, rank, a):
        import threading
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, args=(limit_s, rank, a), daemon=True)
        self._t.start()

    def _run(self, limit_s, rank, a):
        if not self._stop.wait(limit_s):
            fail(a, rank, f"watchdog: the run did not finish within {limit_s:.0f} s (SFMLOC_BENCH_WATCHDOG_S)", 4)

    def stop(self):
        self._stop.set()


def run_all(a, rank, world, local_rank, dist, torch):
    out = measure(argparse.Namespace(**vars(a)), rank, world, local_rank, dist, torch, a.replicas)
    if world > 1 and not a.replicas and not a.no_replica_leg and not a.image_in_only:
        # the comparison SCALE runs need: N independent replicas of the whole map, same queries, no exchange
        b = argparse.Namespace(**vars(a))
        b.no_cpu_baseline = b.no_roofline_phase = b.no_image_in = True
        b.threads = b.in_flight = 0
        rep = measure(b, rank, world, local_rank, dist, torch, True)
        if rank == 0:
            out["replicas"] = {k: rep[k] for k in ("value", "unit", "ms_per_step", "identical_to_single_flight", "latency_ms")}
            out["replicas"]["parallelism"] = rep["config"]["parallelism"]
            out["replicas"]["queries_localised"] = rep["config"]["queries_localised"]
    return out


if __name__ == "__main__":
    main()
