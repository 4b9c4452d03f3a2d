"""bench.py's host-side pieces on the CPU: the module imports, the CPU-baseline legs (the only place outside tests/ and
smoke() that may use the oracle) run on a small map and report the contract's fields, and the committed evidence
bench.py reads (PMC traffic, the HBM-bound sweep) parses."""
import os

import numpy as np

import bench
import synthdata as synth


def test_cpu_baseline_legs_report_the_contract_fields():
    m = synth.make_map(2, n_views=60, desc_per_view=300, views_per_place=10, landmarks_per_place=200, obs_per_view=90)
    queries = [synth.make_query(m, 1000 + i, n_feat=300) for i in range(2)]
    cb = bench.cpu_baseline(m, queries, 1.0)
    assert cb["unit"] == "queries/s" and cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["single_core_value"] > 0 and cb["with_per_query_desc_reread_value"] <= cb["value"]
    assert "views" in cb["sample"]
    rng = np.random.Generator(np.random.PCG64(33))
    proto = rng.uniform(0, 1, (len(m.place_center), 500)).astype(np.float32)
    bow = (proto[m.view_place] + rng.normal(0, 0.05, (60, 500))).astype(np.float32)
    qbow = [(proto[q.place] + rng.normal(0, 0.05, 500)).astype(np.float32) for q in queries]
    cs = bench.cpu_baseline_shortlist(m, queries, bow, qbow, 10, 1.0)
    assert cs["unit"] == "queries/s" and cs["value"] > 0 and "shortlist k=10" in cs["sample"]
    assert cs["single_core_value"] > 0 and 0 < cs["with_per_query_bow_and_desc_reread_value"] <= cs["value"]


def test_committed_evidence_parses():
    """The one field bench.py cannot measure in-process (PMC traffic of the roofline kernel) comes from a committed
    summary of separate rocprofv3 --pmc passes; when the file is present it must parse and be plausible for the
    20 M-row bank (1.28 GB read once)."""
    import os
    traffic, src = bench.pmc_traffic(True)
    if os.path.exists(bench.PMC_SUMMARY):
        assert traffic is not None and 1.2e9 < traffic < 2.0e9 and src.startswith("profiles/")
    else:
        assert traffic is None and src is None
    assert bench.pmc_traffic(False) == (None, None)


def test_gpus_flag_is_honoured(monkeypatch):
    """--gpus N without a launcher starts N ranks (a child torch.distributed.run, never an exec from this process);
    under a launcher a rank refuses a world size that differs from --gpus."""
    import subprocess
    import sys
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append(cmd) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.spawn_ranks(bench.parse(["--gpus", "4", "--steps", "3"])) == 0
    cmd = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert "127.0.0.1" in cmd
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), "bench.py"), "--gpus", "4"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_plan_of_a_run():
    """plan(): the default single-GPU shortlist run keeps 20 queries in flight on 4 host threads and asks for a hardware
    queue per context + 2 (the 24-queue limit is why 22 contexts are slower); N ranks run gangs of 32 on 96 contexts per
    slot; --replicas makes every rank a single-GPU run; the image-in leg's defaults are 10 workers, a worker's BoW
    chains on one stream."""
    a = bench.parse([])
    shortlist, forced, sharded, gang, nctx, hwq = bench.plan(a, 1, False)
    assert shortlist and not sharded and gang == 1 and nctx == 20 and a.threads == 4 and hwq == 22
    assert a.image_workers == 10 and a.image_bow_worker_stream and a.image_bow_own_stream
    a = bench.parse(["--gpus", "8"])
    _, _, sharded, gang, nctx, hwq = bench.plan(a, 8, False)
    assert sharded and gang == 32 and nctx == 96 and hwq == 16
    a = bench.parse(["--gpus", "8", "--replicas"])
    _, _, sharded, gang, nctx, _ = bench.plan(a, 8, True)
    assert not sharded and gang == 1 and nctx == 20
    a = bench.parse(["--bow-knn", "0", "--views", "1000"])
    shortlist, _, _, _, nctx, _ = bench.plan(a, 1, False)
    assert not shortlist and nctx == 4
