"""GPU: BASELINE config 1 -- a 50-image toy OpenMVG map on disk, queries as .desc/.feat files, localised through
sfmloc_open, the LocalizeEngine mirror and the OpenMVGLocalization_AKAZE-compatible command line; results
equal the oracle's on the same inputs and land on the planted pose."""
import json
import os

import numpy as np
import pytest

import sfmlocalization_amd as S
from sfmlocalization_amd import capi, engine, fileio
import synthdata as synth
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy(tmp_path_factory):
    root = tmp_path_factory.mktemp("cfg1")
    m = synth.make_map(1, n_views=50, desc_per_view=400, views_per_place=10, landmarks_per_place=300,
                       obs_per_view=130, view_id_stride=2)
    names = synth.write_map_to_disk(m, str(root / "sfm"), str(root / "matches"), unposed_views=(4,))
    qdir = root / "queries"
    qdir.mkdir()
    queries = []
    for k in range(4):
        q = synth.make_query(m, 50 + k, n_feat=500, n_copies=180, outlier_frac=0.25 if k < 3 else 0.0,
                             place=k % 5)
        if k == 3:
            q = synth.make_query(m, 99, n_feat=300, n_copies=0)          # will not localise
        base = f"q{k:03d}"
        fileio.write_desc(qdir / (base + ".desc"), q.desc)
        kp = np.concatenate([q.kpt_xy, np.zeros((len(q.kpt_xy), 2), np.float32)], 1)
        # the reference keeps full-precision keypoints in memory (locFeat); a .feat file holds 6 digits, so the
        # file-based query IS the rounded one
        fileio.write_feat(qdir / (base + ".feat"), kp)
        (qdir / (base + ".jpg")).write_bytes(b"")                          # image itself is not decoded
        queries.append((base, q))
    return m, root, names, queries


def posed_submap(m, skip_ids):
    """the synthetic map restricted to posed views, as the loader sees it"""
    keep = [k for k, v in enumerate(m.view_id) if int(v) not in skip_ids]
    rows = np.concatenate([np.arange(m.view_off[k], m.view_off[k + 1]) for k in keep]).astype(np.int64)
    off = np.zeros(len(keep) + 1, np.uint32)
    off[1:] = np.cumsum([m.view_off[k + 1] - m.view_off[k] for k in keep])

    class Sub:
        pass
    s = Sub()
    s.view_id, s.view_off, s.view_wh = m.view_id[keep], off, m.view_wh[keep]
    s.desc, s.row_landmark = m.desc[rows], m.row_landmark[rows]
    s.landmark_id, s.landmark_X, s.intrinsic = m.landmark_id, m.landmark_X, m.intrinsic
    return s, keep, rows


def test_open_equals_in_memory_map(toy, oracle_c):
    m, root, names, queries = toy
    sub, keep, rows = posed_submap(m, (4,))
    # map keypoints as the files hold them
    sub.kpt_xy = np.concatenate([fileio.read_feat(root / "matches" / (names[k] + ".feat"))[:, :2] for k in keep])
    p = S.default_params(ransac_round=25)
    with capi.Map.open(str(root / "sfm"), str(root / "matches"), p) as dm:
        assert dm.n_views == 49 and dm.n_rows == len(rows)
        np.testing.assert_array_equal(dm.view_id, sub.view_id)
        np.testing.assert_array_equal(dm.view_off, sub.view_off)
        np.testing.assert_allclose(dm.view_center, m.view_C[keep])
        for base, q in queries:
            desc = fileio.read_desc(root / "queries" / (base + ".desc"))
            kp = fileio.read_feat(root / "queries" / (base + ".feat"))[:, :2]
            dq = dm.query(desc, kp, 640, 480)
            pose, pq, pl = dm.localize(dq)
            dq.close()
            exp = opipe.localize(sub, desc, kp, (640, 480))
            assert bool(pose.ok) == exp["ok"]
            if exp["ok"]:
                np.testing.assert_array_equal(pq, exp["pair_qfeat"])
                np.testing.assert_array_equal(pl, exp["pair_landmark"])
                assert np.abs(np.array(pose.R).reshape(3, 3) - exp["R"]).max() <= 1e-4
                assert np.abs(np.array(pose.center) - exp["center"]).max() <= 1e-4
                assert np.abs(np.array(pose.center) - q.C_true).max() < 0.3


def test_open_packed_equals_open(toy, tmp_path):
    """sfmloc_pack + sfmloc_open_packed: the same map as sfmloc_open on the files (view table, camera centres, every
    localisation bit for bit), opened without touching them again."""
    import time
    m, root, names, queries = toy
    packed = str(tmp_path / "toy.sfmlocmap")
    capi.pack(str(root / "sfm"), str(root / "matches"), packed)
    p = S.default_params(ransac_round=25)
    t0 = time.perf_counter()
    a = capi.Map.open(str(root / "sfm"), str(root / "matches"), p)
    t1 = time.perf_counter()
    b = capi.Map.open_packed(packed, p)
    t2 = time.perf_counter()
    try:
        assert (a.n_views, a.n_rows) == (b.n_views, b.n_rows)
        np.testing.assert_array_equal(a.view_id, b.view_id)
        np.testing.assert_array_equal(a.view_off, b.view_off)
        np.testing.assert_array_equal(a.view_center, b.view_center)
        for base, q in queries:
            desc = fileio.read_desc(root / "queries" / (base + ".desc"))
            kp = fileio.read_feat(root / "queries" / (base + ".feat"))[:, :2]
            ra = a.localize(a.query(desc, kp, 640, 480))
            rb = b.localize(b.query(desc, kp, 640, 480))
            assert ra[0].ok == rb[0].ok and ra[0].n_inliers == rb[0].n_inliers
            np.testing.assert_array_equal(ra[1], rb[1])
            np.testing.assert_array_equal(np.array(ra[0].P).view(np.uint64), np.array(rb[0].P).view(np.uint64))
        assert (t2 - t1) < (t1 - t0)          # no JSON, no text .feat files
    finally:
        a.close()
        b.close()
    # both command-line programs take the packed file in place of <sfmDataDir>: same poses as from the directory
    import subprocess
    out_d, out_p, out_c = tmp_path / "o_dir", tmp_path / "o_packed", tmp_path / "o_packed_cc"
    assert engine.main([str(root / "queries"), str(root / "sfm"), str(root / "matches"), str(out_d), "-r=25"]) == 0
    assert engine.main([str(root / "queries"), packed, str(root / "matches"), str(out_p), "-r=25"]) == 0
    r = subprocess.run([CLI_BIN, str(root / "queries"), packed, str(root / "matches"), str(out_c), "-r=25"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for base, q in queries:
        jd, jp = json.load(open(out_d / (base + ".json"))), json.load(open(out_p / (base + ".json")))
        assert jp["sfm_data"] == packed and jd["sfm_data"].endswith("sfm_data.json")
        for key in ("K", "R", "t", "pair"):
            assert jd.get(key) == jp.get(key), (base, key)
        assert (out_c / (base + ".json")).read_bytes() == (out_p / (base + ".json")).read_bytes(), base


def test_command_line_writes_the_reference_json(toy):
    m, root, names, queries = toy
    out = root / "loc_out"
    rc = engine.main([str(root / "queries"), str(root / "sfm"), str(root / "matches"), str(out), "-f=0.6", "-r=25"])
    assert rc == 0
    files = sorted(os.listdir(out))
    assert files == [b + ".json" for b, _ in queries]
    n_ok = 0
    for base, q in queries:
        d = json.load(open(out / (base + ".json")))
        assert d["filename"].endswith(base + ".jpg") and d["sfm_data"].endswith("sfm_data.json")
        assert d["matches_dir"] == str(root / "matches")
        if "t" in d:                                   # mergeSfM.py:61 success test
            n_ok += 1
            assert np.abs(np.array(d["t"]) - q.C_true).max() < 0.3
            K = np.array(d["K"])
            assert abs(K[0, 0] - 800) < 1e-3 and abs(K[0, 2] - 320) < 1e-3
            assert len(d["pair"]) > 10
            lm_ids = set(int(x) for x in m.landmark_id)
            assert all(int(b) in lm_ids and 0 <= int(a) < 500 for a, b in d["pair"])
        else:
            assert list(d) == ["filename", "sfm_data", "matches_dir"]
    assert n_ok == 3 and "t" not in json.load(open(out / "q003.json"))
    # dead-reckoning restriction (-x -y -z -d): far away centre -> no view -> failure JSON
    out2 = root / "loc_out2"
    rc = engine.main([str(root / "queries" / "q000.jpg"), str(root / "sfm"), str(root / "matches"), str(out2),
                      "-r=25", "-x=1000", "-y=1000", "-z=1000", "-d=4"])
    assert rc == 0 and "t" not in json.load(open(out2 / "q000.json"))
    # bad sfm dir -> error exit like the reference (EXIT_FAILURE)
    assert engine.main([str(root / "queries"), str(root), str(root / "matches"), str(out)]) == 1


CLI_BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfmlocalization_amd", "bin",
                       "OpenMVGLocalization_AKAZE")


def _run_cli(args):
    import subprocess
    env = dict(os.environ)
    r = subprocess.run([CLI_BIN] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_cpp_command_line_equals_python_mirror(toy, capsys):
    """The C++ host program over the C ABI (sfmlocalization_amd/csrc/localize_cli.cpp, the reference's own host
    language for this tool) writes byte-identical result files to the Python mirror's, prints the reference's
    messages and exits like it."""
    m, root, names, queries = toy
    assert os.path.exists(CLI_BIN), "build it: make -C sfmlocalization_amd/csrc"
    out_py, out_cc = root / "cmp_py", root / "cmp_cc"
    args = [root / "queries", root / "sfm", root / "matches"]
    capsys.readouterr()
    assert engine.main([str(a) for a in args] + [str(out_py), "-f=0.6", "-r=25"]) == 0
    py_out = capsys.readouterr().out
    rc, so, se = _run_cli(args + [out_cc, "-f=0.6", "-r=25"])
    assert rc == 0, se
    assert so == py_out                                 # the same console messages, line for line
    assert "number of putative matches : " in so and "number of geometric matches : " in so and "cpt = " in so
    assert so.startswith("Start localizing input image.") and so.count("complete") == 3
    assert "Fail to estimate camera matrix" in so or "Not enough putative matches" in so
    for base, _ in queries:
        assert (out_cc / (base + ".json")).read_bytes() == (out_py / (base + ".json")).read_bytes(), base
    # dead-reckoning restriction and the error exit
    rc, so, se = _run_cli([root / "queries" / "q000.jpg", root / "sfm", root / "matches", root / "cmp_cc2", "-r=25",
                           "-x=1000", "-y=1000", "-z=1000", "-d=4"])
    assert rc == 0 and "t" not in json.load(open(root / "cmp_cc2" / "q000.json")) and "Not enough putative matches" in so
    rc, so, se = _run_cli([root / "queries", root, root / "matches", root / "cmp_cc3"])
    assert rc == 1 and "cannot be read" in se
    rc, so, se = _run_cli([root / "queries" / "q000.jpg", root / "sfm", root / "matches", root / "cmp_cc4", "-r=25",
                           "-x=%g" % m.view_C[0, 0], "-y=%g" % m.view_C[0, 1], "-z=%g" % m.view_C[0, 2], "-d=40"])
    out_py4 = root / "cmp_py4"
    engine.main([str(root / "queries" / "q000.jpg"), str(root / "sfm"), str(root / "matches"), str(out_py4), "-r=25",
                 "-x=%g" % m.view_C[0, 0], "-y=%g" % m.view_C[0, 1], "-z=%g" % m.view_C[0, 2], "-d=40"])
    assert (root / "cmp_cc4" / "q000.json").read_bytes() == (out_py4 / "q000.json").read_bytes()
    # -w: <matchDir>/matches.fQ.txt = the geometric matches of the (last) query, keyed (view id, last view id + 1)
    # (localization.cpp:371,452-455): the same bytes from both programs, and the oracle's inlier lists
    fq = root / "matches" / "matches.fQ.txt"
    one = [str(root / "queries" / "q000.jpg"), str(root / "sfm"), str(root / "matches")]
    assert engine.main(one + [str(root / "w_py"), "-r=25", "-w"]) == 0
    py_bytes = fq.read_bytes()
    os.remove(fq)
    rc, so, se = _run_cli(one + [root / "w_cc", "-r=25", "-w"])
    assert rc == 0, se
    assert fq.read_bytes() == py_bytes and len(py_bytes) > 100
    got = fileio.read_matches_txt(fq)
    sub, keep, rows = posed_submap(m, (4,))
    sub.kpt_xy = np.concatenate([fileio.read_feat(root / "matches" / (names[k] + ".feat"))[:, :2] for k in keep])
    desc = fileio.read_desc(root / "queries" / "q000.desc")
    kp = fileio.read_feat(root / "queries" / "q000.feat")[:, :2]
    exp = opipe.localize(sub, desc, kp, (640, 480))
    ind_q = int(max(m.view_id)) + 1
    assert sorted(got) == [(int(sub.view_id[v]), ind_q) for v in np.nonzero(exp["geo_count"])[0]]
    for v in np.nonzero(exp["geo_count"])[0]:
        o0 = int(sub.view_off[v])
        pp = exp["geo_idx"][o0:o0 + int(exp["geo_count"][v])].astype(np.int64)
        gi, gj = got[(int(sub.view_id[v]), ind_q)]
        np.testing.assert_array_equal(gi, exp["put_i"][o0 + pp])
        np.testing.assert_array_equal(gj, exp["put_j"][o0 + pp])
    os.remove(fq)


def test_engine_mirror_return_convention(toy):
    m, root, names, queries = toy
    A = np.array([[0, -2.0, 0, 10], [2.0, 0, 0, -5], [0, 0, 2.0, 1]])      # similarity: scale 2, rot 90 deg about z
    fileio.write_cv_yaml(root / "Amat.yml", {"A": A})
    e0 = engine.LocalizeEngine(str(root / "sfm"), str(root / "matches"), None, 0.6, 25, 4.0, False)
    e1 = engine.LocalizeEngine(str(root / "sfm"), str(root / "matches"), str(root / "Amat.yml"), 0.6, 25, 4.0, False)
    base, q = queries[0]
    desc = fileio.read_desc(root / "queries" / (base + ".desc"))
    kp = fileio.read_feat(root / "queries" / (base + ".feat"))[:, :2]
    r0, ex0 = e0.localize(desc, kp, 640, 480, return_time=True)
    r1, _ = e1.localize(desc, kp, 640, 480)
    assert len(r0) == 12 and len(r1) == 12 and len(ex0["times"]) == 6        # LocalizeEngine.cc:593-602,651-657
    # the reference's stage buckets: putMatch, geoMatch and PnP come from this query's HIP events, are non-zero and sum
    # to no more than the call's wall time; no iBeacon / BoW / extraction stage ran in this call
    t = ex0["times"]
    assert t[0] == 0.0 and t[1] == 0.0 and t[2] == 0.0 and min(t[3:]) > 0.0 and sum(t) <= ex0["time_total"]
    ss = list(ex0["pose"].stage_seconds)
    assert abs(sum(ss) - ex0["time_total"]) < 0.05 and ss[6] >= 0.0
    c0, R0 = np.array(r0[:3]), np.array(r0[3:]).reshape(3, 3)
    c1, R1 = np.array(r1[:3]), np.array(r1[3:]).reshape(3, 3)
    np.testing.assert_allclose(c1, A[:, :3] @ c0 + A[:, 3], atol=1e-9)
    np.testing.assert_allclose(R1, R0 @ (A[:, :3] / 2.0).T, atol=1e-12)
    np.testing.assert_allclose(R1 @ R1.T, np.eye(3), atol=1e-9)
    # restricting to views near the true position keeps the answer; restricting far away gives the empty vector
    near, _ = e0.localize(desc, kp, 640, 480, center=list(q.C_true), radius=40.0 ** 2)
    assert len(near) == 12 and np.abs(np.array(near[:3]) - q.C_true).max() < 0.3
    far, _ = e0.localize(desc, kp, 640, 480, center=[1e3, 1e3, 1e3], radius=1.0)
    assert far == []
    base3, _ = queries[3]
    none, _ = e0.localize(fileio.read_desc(root / "queries" / (base3 + ".desc")),
                          fileio.read_feat(root / "queries" / (base3 + ".feat"))[:, :2], 640, 480)
    assert none == []                                                        # LocalizeEngine.cc:453,481,579
    # the header-only C++ class (include/sfmloc_engine.hpp) driven by tests/cpp/engine_smoke.cpp: same numbers
    import subprocess
    smoke = os.path.join(os.path.dirname(CLI_BIN), "engine_smoke")
    qd, qf = str(root / "queries" / (base + ".desc")), str(root / "queries" / (base + ".feat"))

    def run_cpp(amat, extra=()):
        r = subprocess.run([smoke, str(root / "sfm"), str(root / "matches"), amat, qd, qf, "640", "480"] + list(extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().split("\n")
        vals = [] if lines[0] == "FAIL" else [float(x) for x in lines[0].split()]
        run_cpp.times = [float(x) for x in lines[4].split()] if len(lines) > 4 else []
        return vals, int(lines[1]), [int(x) for x in lines[2].split()], int(lines[3])

    v0, n23, inl, ntimes = run_cpp("-")
    assert v0 == r0 and ntimes == 6 and n23 == ex0["pose"].n_matches_2d3d and len(inl) == ex0["pose"].n_inliers
    ct = run_cpp.times                        # six buckets + the call's wall time, from the C++ class
    assert len(ct) == 7 and ct[0] == 0.0 and min(ct[3:6]) > 0.0 and sum(ct[:6]) <= ct[6]
    v1, _, _, _ = run_cpp(str(root / "Amat.yml"))
    np.testing.assert_allclose(v1, r1, rtol=0, atol=1e-12)
    vfar, _, _, _ = run_cpp("-", ["1000", "1000", "1000", "1"])
    assert vfar == []
    vnear, _, _, _ = run_cpp("-", ["%r" % q.C_true[0], "%r" % q.C_true[1], "%r" % q.C_true[2], "1600"])
    assert vnear == near
    # sfmloc::ShardRank (the same header): a rank's batch entry points from C++, world size 1 -- two copies of the query
    # through a gang session of stage 1 and the merge contexts of stage 2 = the unsharded sfmloc_localize, bit for bit
    shard_smoke = os.path.join(os.path.dirname(CLI_BIN), "shard_rank_smoke")
    r = subprocess.run([shard_smoke, str(root / "sfm"), str(root / "matches"), qd, qf, "640", "480"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.split() == ["OK", str(len(inl))]
    e0.close()
    e1.close()


def test_bow_shortlist_in_engine(tmp_path_factory):
    """-k / bowKnnNum: views are shortlisted by their .bow vectors before matching (localization.cpp:346-368)."""
    root = tmp_path_factory.mktemp("bowmap")
    m = synth.make_map(3, n_views=60, desc_per_view=300, views_per_place=10, landmarks_per_place=250, obs_per_view=110)
    rng = np.random.Generator(np.random.PCG64(8))
    proto = np.sqrt(rng.random((6, 500)))                      # one BoW prototype per place
    bow = np.clip(proto[m.view_place] + rng.normal(0, 0.02, (60, 500)), 0, None)
    synth.write_map_to_disk(m, str(root / "sfm"), str(root / "matches"), with_bow=bow)
    q = synth.make_query(m, 77, n_feat=450, n_copies=160, outlier_frac=0.2, place=2)
    kp = synth.round6(q.kpt_xy)
    e_all = engine.LocalizeEngine(str(root / "sfm"), str(root / "matches"), None, 0.6, 25, 4.0, False, 0, 0)
    e_bow = engine.LocalizeEngine(str(root / "sfm"), str(root / "matches"), None, 0.6, 25, 4.0, False, 0, 10)
    r_all, x_all = e_all.localize(q.desc, kp, 640, 480)
    qbow = proto[2] + rng.normal(0, 0.02, 500)
    sel = e_bow.map.bow_select(qbow.astype(np.float32), 10)
    assert set(sel) == set(np.nonzero(m.view_place == 2)[0])   # the 10 views of the query's place
    r_bow, x_bow = e_bow.localize(q.desc, kp, 640, 480, bow=qbow)
    assert len(r_all) == 12 and len(r_bow) == 12
    # every surviving view was in the shortlist anyway -> identical answer, an order of magnitude less matching
    assert r_all == r_bow and x_all["pairs"] == x_bow["pairs"]
    assert x_bow["pose"].n_putative_views == x_all["pose"].n_putative_views
    r_none, _ = e_bow.localize(q.desc, kp, 640, 480, bow=proto[5])        # wrong place shortlisted -> fails
    assert r_none == []
    e_all.close()
    e_bow.close()


def test_open_radial_k3_map(tmp_path, oracle_c):
    """sfm_data.json with a pinhole_radial_k3 intrinsic (cereal nests the pinhole base as value0): the loader picks
    up disto_k3 and the 2D points of the resection are undistorted, exactly as the in-memory map does."""
    m = synth.make_map(3, n_views=20, desc_per_view=300, views_per_place=10, landmarks_per_place=250, obs_per_view=120)
    m.intrinsic = tuple(m.intrinsic[:3]) + (-0.1, 0.02, 0.0)
    names = synth.write_map_to_disk(m, str(tmp_path / "sfm"), str(tmp_path / "matches"))
    info = capi.scan(str(tmp_path / "sfm"), str(tmp_path / "matches"))
    assert (info["k1"], info["k2"], info["k3"]) == (-0.1, 0.02, 0.0)
    m.kpt_xy = np.concatenate([fileio.read_feat(tmp_path / "matches" / (n + ".feat"))[:, :2] for n in names])
    q = synth.make_query(m, 77, n_feat=500, n_copies=200, outlier_frac=0.2)
    p = S.default_params(ransac_round=25)
    with capi.Map.open(str(tmp_path / "sfm"), str(tmp_path / "matches"), p) as dm:
        dq = dm.query(q.desc, q.kpt_xy, 640, 480)
        pose, pq, pl = dm.localize(dq)
        dq.close()
    exp = opipe.localize(m, q.desc, q.kpt_xy, (640, 480))
    assert bool(pose.ok) == exp["ok"] and exp["ok"]
    np.testing.assert_array_equal(pq, exp["pair_qfeat"])
    np.testing.assert_array_equal(np.array(pose.P), exp["P"].ravel())
