"""GPU: the ExtFeatAndMatch mirror (sfmlocalization_amd/extfeat.py) on a rendered image sequence: images in,
<base>.feat/.desc + matches.putative.txt + matches.f.txt out, each stage equal to the oracle's on the same files."""
import os

import numpy as np
import pytest

from sfmlocalization_amd import extfeat, fileio

import synthdata as synth
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu
F, W, H, PPM = 800.0, 640, 480, 100.0


def render_sequence(tmp_path, n=6):
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(8))
    tex = synth.texture_image(17, 1600, 1600, n_blobs=3000, n_rects=900)
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    names = []
    for k in range(n):
        R, C = synth.plane_camera(rng, (6.0 + 0.8 * k, 8.0 + 0.2 * k), 10.0, tilt=0.08)
        name = f"frame{k:04d}.png"
        Image.fromarray(synth.render_plane_view(tex, PPM, R, C, F, W, H)).save(img_dir / name)
        names.append(name)
    match_dir = tmp_path / "matches"
    match_dir.mkdir()
    sd = fileio.make_sfm_data(list(range(n)), names, W, H, F, W / 2, H / 2, root_path=str(img_dir))
    fileio.write_sfm_data(match_dir / "sfm_data.json", sd)
    return match_dir, names


def test_track_mode_end_to_end(tmp_path, oracle_c):
    match_dir, names = render_sequence(tmp_path)
    rc = extfeat.main([str(match_dir), "-mf=3", "-mm=20", "-r=200"])
    assert rc == 0
    bases = [os.path.splitext(n)[0] for n in names]
    opt = fileio.read_image_describer(match_dir / "image_describer.txt")
    assert int(opt["desc_ch"]) == 3 and int(opt["nOct"]) == 4
    descs = [fileio.read_desc(match_dir / (b + ".desc")) for b in bases]
    kps = [fileio.read_feat(match_dir / (b + ".feat"))[:, :2] for b in bases]
    assert all(len(d) > 100 and len(d) == len(k) for d, k in zip(descs, kps))
    put = fileio.read_matches_txt(match_dir / "matches.putative.txt")
    exp = opipe.track_akaze(descs, 3, 0.6)
    assert list(put) == list(exp)
    for k in exp:
        np.testing.assert_array_equal(put[k][0], exp[k][0])
        np.testing.assert_array_equal(put[k][1], exp[k][1])
    assert sum(len(v[0]) for v in put.values()) > 300
    kept = {k: v for k, v in put.items() if len(v[0]) >= 20}
    geo = fileio.read_matches_txt(match_dir / "matches.f.txt")
    egeo = opipe.geometric_match(kps, [(W, H)] * len(names), list(range(len(names))), kept, ransac_round=200)
    assert list(geo) == list(egeo) and len(geo) >= 4
    for k in egeo:
        np.testing.assert_array_equal(geo[k][0], egeo[k][0])
        np.testing.assert_array_equal(geo[k][1], egeo[k][1])
    # a second run keeps every file (the reference skips what exists)
    stamp = {p: os.path.getmtime(match_dir / p) for p in os.listdir(match_dir)}
    assert extfeat.main([str(match_dir), "-mf=3", "-mm=20", "-r=200"]) == 0
    assert all(os.path.getmtime(match_dir / p) == t for p, t in stamp.items() if p != "image_describer.txt")


def test_guided_matching_end_to_end(tmp_path, oracle_c):
    """-gm (the map builder's default, reconstructGraph.py:155-163): matches.f.txt holds, for every pair that passes the
    F-matrix AC-RANSAC, OpenMVG's guided matches under the estimated F over ALL features of both images -- equal to the
    oracle's, different from the unguided file, and identical from the C++ tool."""
    import shutil
    import subprocess
    match_dir, names = render_sequence(tmp_path)
    assert extfeat.main([str(match_dir), "-mf=3", "-mm=20", "-r=200", "-gm"]) == 0
    bases = [os.path.splitext(n)[0] for n in names]
    descs = [fileio.read_desc(match_dir / (b + ".desc")) for b in bases]
    kps = [fileio.read_feat(match_dir / (b + ".feat"))[:, :2] for b in bases]
    put = fileio.read_matches_txt(match_dir / "matches.putative.txt")
    kept = {k: v for k, v in put.items() if len(v[0]) >= 20}
    geo = fileio.read_matches_txt(match_dir / "matches.f.txt")
    egeo = opipe.geometric_match(kps, [(W, H)] * len(names), list(range(len(names))), kept, ransac_round=200,
                                 guided=True, descs=descs)
    plain = opipe.geometric_match(kps, [(W, H)] * len(names), list(range(len(names))), kept, ransac_round=200)
    assert list(geo) == list(egeo) == list(plain) and len(geo) >= 4
    n_diff = 0
    for k in egeo:
        np.testing.assert_array_equal(geo[k][0], egeo[k][0])
        np.testing.assert_array_equal(geo[k][1], egeo[k][1])
        assert (np.diff(geo[k][0].astype(np.int64)) > 0).all()           # one match per feature of I, ascending
        n_diff += int(len(geo[k][0]) != len(plain[k][0]) or not np.array_equal(np.sort(plain[k][0]), geo[k][0]))
    assert n_diff >= 1                                                    # guided matching changes the map's matches
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfmlocalization_amd", "bin",
                       "ExtFeatAndMatch")
    match_cc = tmp_path / "matches_cc"
    match_cc.mkdir()
    shutil.copy(match_dir / "sfm_data.json", match_cc / "sfm_data.json")
    r = subprocess.run([exe, str(match_cc), "-mf=3", "-mm=20", "-r=200", "-gm"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert (match_cc / "matches.f.txt").read_bytes() == (match_dir / "matches.f.txt").read_bytes()


def test_pair_modes(tmp_path, oracle_c):
    match_dir, names = render_sequence(tmp_path, n=4)
    assert extfeat.main([str(match_dir), "-sm"]) == 1            # extraction only; the reference returns 1 here
    assert not os.path.exists(match_dir / "matches.putative.txt")
    assert extfeat.main([str(match_dir), "-v=2", "-mm=20", "-r=100"]) == 0
    put = fileio.read_matches_txt(match_dir / "matches.putative.txt")
    bases = [os.path.splitext(n)[0] for n in names]
    descs = [fileio.read_desc(match_dir / (b + ".desc")) for b in bases]
    exp = opipe.match_akaze(descs, extfeat.generate_video_match_pairs(list(range(4)), 2), 0.6)
    assert list(put) == list(exp)
    for k in exp:
        np.testing.assert_array_equal(put[k][0], exp[k][0])
    assert extfeat.main([str(match_dir), "-v=2", "-mf=3"]) == 1  # mutually exclusive options


def test_cpp_tool_writes_the_same_files(tmp_path):
    """sfmlocalization_amd/bin/ExtFeatAndMatch (C++ over the C ABI) against the Python mirror: every output file
    byte for byte, in track mode and in video-pair mode, and the same exit codes."""
    import shutil
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfmlocalization_amd", "bin",
                       "ExtFeatAndMatch")
    (tmp_path / "a").mkdir()
    match_py, names = render_sequence(tmp_path / "a", n=5)
    match_cc = tmp_path / "matches_cc"
    match_cc.mkdir()
    shutil.copy(match_py / "sfm_data.json", match_cc / "sfm_data.json")
    for mode in (["-mf=3", "-mm=20", "-r=150"], ["-v=2", "-mm=20", "-r=150"]):
        for d in (match_py, match_cc):
            for f in ("matches.putative.txt", "matches.f.txt"):
                if os.path.exists(d / f):
                    os.remove(d / f)
        assert extfeat.main([str(match_py)] + mode) == 0
        r = subprocess.run([exe, str(match_cc)] + mode, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "Start Putative Matching..." in r.stdout and "number of geometric matches" in r.stdout
        files = sorted(f for f in os.listdir(match_py))
        assert files == sorted(os.listdir(match_cc))
        for f in files:
            if f != "sfm_data.json":
                assert (match_py / f).read_bytes() == (match_cc / f).read_bytes(), (mode, f)
        assert os.path.getsize(match_cc / "matches.f.txt") > 100
    r = subprocess.run([exe, str(match_cc), "-v=2", "-mf=3"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1
    r = subprocess.run([exe, str(match_cc), "-sm"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Skip matching" in r.stdout
    r = subprocess.run([exe, str(tmp_path / "nowhere")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Cannot load" in r.stderr
