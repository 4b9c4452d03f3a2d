"""bench.py's host-side pieces on the CPU: the module imports, the CPU-baseline legs (the only place outside tests/ and
smoke() that may use the oracle) run on a small map and report the contract's fields, and the committed evidence
bench.py reads (PMC traffic, the HBM-bound sweep) parses."""
import numpy as np

import bench
from sfmlocalization_amd import synth


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


def test_committed_evidence_parses():
    traffic, src = bench.pmc_traffic(True)
    assert traffic is not None and 1.0e8 < traffic < 2.0e8 and src.startswith("profiles/")
    hr = bench.hbm_regime()
    assert hr is not None and hr["nq"] <= 8 and 0.3 < hr["frac"] < 1.0 and hr["source"].startswith("profiles/")
