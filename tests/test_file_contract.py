"""CPU: the on-disk contract -- Python codecs round-trip byte for byte, and the native
loader (sfmloc_scan, the host half of sfmloc_open) reads the same thing from the same files."""
import json
import os
import struct

import numpy as np
import pytest

from sfmlocalization_amd import capi, fileio

import synthdata as synth


def test_desc_layout_and_roundtrip(tmp_path):
    rng = np.random.Generator(np.random.PCG64(1))
    d61 = rng.integers(0, 256, (7, 61), dtype=np.uint8)
    p = tmp_path / "a.desc"
    fileio.write_desc(p, d61)
    raw = p.read_bytes()
    assert len(raw) == 8 + 7 * 64 and struct.unpack("<Q", raw[:8])[0] == 7     # u64 count, 64-byte rows
    back = fileio.read_desc(p)
    assert back.shape == (7, 64) and (back[:, :61] == d61).all() and (back[:, 61:] == 0).all()  # FileUtils.cpp:82-88
    fileio.write_desc(p, np.zeros((0, 64), np.uint8))
    assert fileio.read_desc(p).shape == (0, 64)
    p.write_bytes(struct.pack("<Q", 3) + b"\0" * 100)
    with pytest.raises(IOError):
        fileio.read_desc(p)


def test_feat_is_six_significant_digits(tmp_path):
    k = np.array([[123.456789, 0.000123456789, 4.8, 359.9999], [1e-7, 639.5, 7.25, 0.0]], np.float32)
    p = tmp_path / "a.feat"
    fileio.write_feat(p, k)
    lines = p.read_text().splitlines()
    assert lines[0].split() == ["123.457", "0.000123457", "4.8", "360"]      # `ostream << float`, AKAZEOpenCV.cpp:80-81
    back = fileio.read_feat(p)
    np.testing.assert_allclose(back, synth.round6(k), rtol=0, atol=0)


def test_bow_mat_bin(tmp_path):
    v = np.linspace(0, 1, 500).reshape(500, 1)
    p = tmp_path / "a.bow"
    fileio.write_mat_bin(p, v)
    raw = p.read_bytes()
    assert struct.unpack("<iii", raw[:12]) == (500, 1, 6) and len(raw) == 12 + 4000    # CV_64F = 6, TrainBoW.cpp:268
    np.testing.assert_array_equal(fileio.read_mat_bin(p), v)
    fileio.write_mat_bin(p, None)
    assert p.read_bytes() == struct.pack("<i", 0)                                   # empty mat: FileUtils.cpp:47-51
    assert fileio.read_mat_bin(p).size == 0


def test_cv_yaml_and_image_describer(tmp_path):
    p = tmp_path / "image_describer.txt"
    fileio.write_image_describer(p)
    assert fileio.read_image_describer(p) == {"desc_ch": 3, "thres": 0.001, "nOct": 4, "nOctLay": 4}  # AKAZEOption.h:31-34
    assert fileio.read_image_describer(tmp_path / "missing.txt")["nOct"] == 4
    centers = np.arange(12, dtype=np.float32).reshape(3, 4)
    fileio.write_cv_yaml(tmp_path / "BOWfile.yml", {"K": 3, "ResizedImageSize": 300, "NormBofFeatureType": "L1",
                                                   "UseSpatialPyramid": 1, "PyramidLevel": 2, "Centers": centers})
    y = fileio.read_cv_yaml(tmp_path / "BOWfile.yml")
    assert y["K"] == 3 and y["NormBofFeatureType"] == "L1" and y["PyramidLevel"] == 2
    np.testing.assert_array_equal(y["Centers"], centers)


def test_result_json_format():
    K = np.array([[800, 0, 320], [0, 800, 240], [0, 0, 1.0]])
    R = np.array([[0.123456789, -0.5, 1e-7], [0, 1, 0], [-1.5, 2.25, 3]])
    s = fileio.format_result_json("/q/img.jpg", "/sfm/sfm_data.json", "/m", K=K, R=R, center=[1.0, -2.5, 3.14159265],
                                  pairs=[(5, 100), (7, 101)])
    d = json.loads(s)
    assert list(d) == ["filename", "sfm_data", "matches_dir", "K", "R", "t", "pair"]       # localization.cpp:125-143
    assert d["K"] == K.tolist() and d["pair"] == [[5, 100], [7, 101]]
    assert d["R"][0] == [0.123457, -0.5, 1e-07] and d["t"] == [1, -2.5, 3.14159]         # 6 significant digits
    assert '"K": [[800,  0,320],\n[  0,800,240],\n[  0,  0,  1]],' in s                    # Eigen aligned columns
    f = json.loads(fileio.format_result_json("/q/img.jpg", "/sfm/sfm_data.json", "/m"))
    assert list(f) == ["filename", "sfm_data", "matches_dir"] and "t" not in f             # failure: localization.cpp:84-109


@pytest.fixture(scope="module")
def toy_map(tmp_path_factory):
    root = tmp_path_factory.mktemp("toy")
    m = synth.make_map(1, n_views=50, desc_per_view=300, views_per_place=10, landmarks_per_place=200,
                       obs_per_view=90, ragged=True, view_id_stride=3)
    bow = np.random.Generator(np.random.PCG64(3)).random((50, 500))
    names = synth.write_map_to_disk(m, str(root / "sfm"), str(root / "matches"), unposed_views=(6, 9), with_bow=bow)
    return m, str(root / "sfm"), str(root / "matches"), names


def test_native_loader_reads_the_contract(toy_map):
    m, sfm_dir, match_dir, names = toy_map
    info = capi.scan(sfm_dir, match_dir)
    posed = [k for k, v in enumerate(m.view_id) if int(v) not in (6, 9)]
    rows = np.concatenate([np.arange(m.view_off[k], m.view_off[k + 1]) for k in posed]).astype(np.int64)
    assert info["n_views_total"] == 50 and info["n_views_posed"] == 48                   # localization.cpp:337-341
    assert info["n_rows"] == len(rows)
    assert info["n_landmarks"] == len(m.landmark_id)
    assert info["n_observations"] == int((m.row_landmark[rows] >= 0).sum())
    assert (info["focal"], info["ppx"], info["ppy"]) == m.intrinsic and info["bow_dim"] == 500
    h = 1469598103934665603
    for b in m.desc[rows].tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert info["desc_fnv1a"] == h
    k6 = np.concatenate([fileio.read_feat(os.path.join(match_dir, names[k] + ".feat"))[:, :2] for k in posed])
    assert abs(info["kpt_sum"] - float(k6.astype(np.float64).sum())) < 1e-6
    assert info["row_landmark_sum"] == int(m.row_landmark[rows].astype(np.int64).sum())


def test_packed_map_file_round_trip(toy_map, tmp_path):
    """sfmloc_pack -> one binary; sfmloc_scan_packed reports exactly what sfmloc_scan reports from the files it was
    packed from (descriptor hash, keypoint and landmark sums included); damaged files are refused."""
    m, sfm_dir, match_dir, names = toy_map
    packed = str(tmp_path / "toy.sfmlocmap")
    capi.pack(sfm_dir, match_dir, packed)
    assert capi.scan_packed(packed) == capi.scan(sfm_dir, match_dir)
    raw = open(packed, "rb").read()
    assert raw[:8] == b"SFMLOCM1" and len(raw) > 64 * info_rows(m)
    for bad in (raw[:len(raw) // 2], b"SFMLOCM2" + raw[8:], raw[:8]):
        (tmp_path / "bad.bin").write_bytes(bad)
        with pytest.raises(capi.SfmlocError) as ei:
            capi.scan_packed(str(tmp_path / "bad.bin"))
        assert ei.value.code == capi.EIO
    with pytest.raises(capi.SfmlocError):
        capi.scan_packed(str(tmp_path / "missing.bin"))
    with pytest.raises(capi.SfmlocError):
        capi.pack(str(tmp_path), match_dir, packed)            # no sfm_data.json there


def info_rows(m):
    return int(m.view_off[-1]) // 2


def test_native_loader_refuses_other_camera_models(toy_map, tmp_path):
    """A map whose intrinsic is another OpenMVG model (radial_k1, brown_t2 ...) is refused instead of being localised
    with the wrong undistortion."""
    m, sfm_dir, match_dir, names = toy_map
    text = open(os.path.join(sfm_dir, "sfm_data.json")).read()
    assert '"principal_point"' in text
    bad = text.replace('"principal_point"', '"disto_k1": [0.1], "principal_point"', 1)
    (tmp_path / "sfm_data.json").write_text(bad)
    with pytest.raises(capi.SfmlocError) as ei:
        capi.scan(str(tmp_path), match_dir)
    assert ei.value.code == capi.EIO and "disto_k1" in str(ei.value) and "pinhole_radial_k3" in str(ei.value)


def test_native_loader_errors(toy_map, tmp_path):
    m, sfm_dir, match_dir, names = toy_map
    with pytest.raises(capi.SfmlocError) as ei:
        capi.scan(str(tmp_path), match_dir)
    assert ei.value.code == capi.EIO and "cannot be read" in str(ei.value)              # localization.cpp:241
    (tmp_path / "sfm_data.json").write_text('{"views": [ {"key": 0, ')
    with pytest.raises(capi.SfmlocError) as ei:
        capi.scan(str(tmp_path), match_dir)
    assert "JSON error" in str(ei.value)
    with pytest.raises(capi.SfmlocError) as ei:
        capi.scan(sfm_dir, str(tmp_path))                                              # descriptor files missing
    assert "cannot open" in str(ei.value)


def test_cli_argument_syntax():
    from sfmlocalization_amd import engine
    pos, o = engine.parse_cv_args(["q.jpg", "sfm", "m", "out", "-f=0.7", "-r=25", "-k=20", "-d=-1.0", "-gm", "-x=-3.5"],
                                  engine.KEYS)
    assert pos == ["q.jpg", "sfm", "m", "out"]
    assert o["fDistRatio"] == 0.7 and o["ransacRound"] == 25 and o["knnbow"] == 20 and o["guidedMatch"] is True
    assert o["cenRadius"] == -1.0 and o["cenLocX"] == -3.5 and o["geomLimit"] == 4.0 and o["locEvryNFrame"] == 1
    _, d = engine.parse_cv_args(["a", "b", "c", "d"], engine.KEYS)               # defaults: localization.cpp:70-82
    assert (d["fDistRatio"], d["ransacRound"], d["knnbow"], d["geomLimit"]) == (0.6, 200, 0, 4.0)


def test_matches_txt_roundtrip(tmp_path):
    from sfmlocalization_amd import fileio
    m = {(0, 1): (np.array([0, 5, 9], np.uint32), np.array([3, 2, 7], np.uint32)),
         (0, 2): (np.zeros(0, np.uint32), np.zeros(0, np.uint32)),
         (3, 10): (np.array([1], np.uint32), np.array([1], np.uint32))}
    p = tmp_path / "matches.putative.txt"
    fileio.write_matches_txt(p, m)
    assert p.read_text() == "0 1\n3\n0 3\n5 2\n9 7\n0 2\n0\n3 10\n1\n1 1\n"   # FileUtils.cpp:123-148 layout
    back = fileio.read_matches_txt(p)
    assert list(back) == sorted(m)
    for k in m:
        assert back[k][0].tolist() == m[k][0].tolist() and back[k][1].tolist() == m[k][1].tolist()


def test_pair_generators_literal():
    from sfmlocalization_amd import extfeat
    ids = [0, 2, 4, 6]
    assert extfeat.generate_all_pairs(ids) == [(0, 2), (0, 4), (0, 6), (2, 4), (2, 6), (4, 6)]
    assert extfeat.generate_video_match_pairs(ids, 2) == [(0, 2), (0, 4), (2, 4), (2, 6), (4, 6)]
    # removeDupPairs (SfMDataUtils.cpp:168-187): a later duplicate (either orientation) goes, earlier entries end up ordered
    assert extfeat.remove_dup_pairs([(3, 1), (1, 3), (2, 5), (5, 2), (1, 3)]) == [(1, 3), (2, 5)]
    assert extfeat.remove_dup_pairs([(0, 1)]) == [(0, 1)]


GOLD_FILES = os.path.join(os.path.dirname(__file__), "golden", "files")


def test_golden_file_samples(tmp_path):
    """Hand-built byte-level samples of the on-disk contract (tests/golden/files, written by make_golden.py WITHOUT
    fileio): the codecs read them to the known values and write the same bytes back."""
    d = fileio.read_desc(os.path.join(GOLD_FILES, "img000.desc"))
    assert d.shape == (3, 64) and d[1, 0] == 64 and d[2, 60] == (188 & 0x3F) and (d[:, 61:] == 0).all()
    fileio.write_desc(tmp_path / "a.desc", d)
    assert (tmp_path / "a.desc").read_bytes() == open(os.path.join(GOLD_FILES, "img000.desc"), "rb").read()
    k = fileio.read_feat(os.path.join(GOLD_FILES, "img000.feat"))
    np.testing.assert_array_equal(k, np.array([[12.5, 7.25, 4.8, 90], [123.457, 0.000123457, 9.6, 359.5],
                                               [639, 479, 19.2, 0]], np.float32))
    fileio.write_feat(tmp_path / "a.feat", k)
    assert (tmp_path / "a.feat").read_text() == open(os.path.join(GOLD_FILES, "img000.feat")).read()
    b = fileio.read_mat_bin(os.path.join(GOLD_FILES, "img000.bow"))
    assert b.shape == (4, 1) and b.ravel().tolist() == [0.0, 0.25, 0.5, 1.0]
    fileio.write_mat_bin(tmp_path / "a.bow", b)
    assert (tmp_path / "a.bow").read_bytes() == open(os.path.join(GOLD_FILES, "img000.bow"), "rb").read()
    o = fileio.read_image_describer(os.path.join(GOLD_FILES, "image_describer.txt"))
    assert o["desc_ch"] == 3 and o["nOct"] == 4 and o["nOctLay"] == 4 and abs(o["thres"] - 0.001) < 1e-9
    sd = fileio.read_sfm_data(os.path.join(GOLD_FILES, "sfm_data.json"))
    assert [v["key"] for v in sd["views"]] == [0, 1] and sd["structure"][0]["key"] == 5


def test_native_loader_reads_golden_files():
    """sfmloc_scan (host-only half of sfmloc_open) on the hand-built samples: one posed view of two, three rows, the
    landmark observed by feature 1."""
    info = capi.scan(GOLD_FILES, GOLD_FILES)
    assert info["n_views_total"] == 2 and info["n_views_posed"] == 1 and info["n_rows"] == 3
    assert info["n_landmarks"] == 1 and info["n_observations"] == 1 and info["bow_dim"] == 4
    assert info["row_landmark_sum"] == -1 + 0 + -1          # rows 0 and 2 unobserved, feature 1 -> landmark slot 0
    assert (info["focal"], info["ppx"], info["ppy"]) == (800.0, 320.0, 240.0)


def test_loader_refuses_hostile_files_without_crashing(tmp_path):
    """Sizes read from files are checked against the file before anything is allocated by them, the JSON reader has a
    nesting limit, and no C++ exception crosses the C ABI: damaged inputs give SFMLOC_EIO, never an abort."""
    import struct
    m = synth.make_map(3, n_views=4, desc_per_view=30, views_per_place=4, landmarks_per_place=40, obs_per_view=15)
    sfm, mat = tmp_path / "sfm", tmp_path / "matches"
    sfm.mkdir()
    mat.mkdir()
    names = synth.write_map_to_disk(m, str(sfm), str(mat))
    assert capi.scan(str(sfm), str(mat))["n_rows"] == m.n_rows
    # a .desc whose count field claims far more rows than the file holds (2^60 would be ~10^19 bytes)
    victim = mat / (names[1] + ".desc")
    good = victim.read_bytes()
    for claimed in (1 << 60, 31, 1 << 33):
        victim.write_bytes(struct.pack("<Q", claimed) + good[8:])
        with pytest.raises(capi.SfmlocError) as e:
            capi.scan(str(sfm), str(mat))
        assert e.value.code == -4 and "truncated" in str(e.value)                 # SFMLOC_EIO
    victim.write_bytes(good)
    # a packed file with an absurd vector length
    packed = tmp_path / "map.bin"
    capi.pack(str(sfm), str(mat), str(packed))
    raw = bytearray(packed.read_bytes())
    assert capi.scan_packed(str(packed))["n_rows"] == m.n_rows
    off = 8 + 48 + 20                                    # magic, six doubles, five u32: the first vector's length
    raw[off:off + 8] = struct.pack("<Q", (1 << 36) - 1)
    packed.write_bytes(bytes(raw))
    with pytest.raises(capi.SfmlocError) as e:
        capi.scan_packed(str(packed))
    assert e.value.code == -4
    # a deeply nested sfm_data.json (the reader recurses per level)
    (sfm / "sfm_data.json").write_text("[" * 100000 + "]" * 100000)
    with pytest.raises(capi.SfmlocError) as e:
        capi.scan(str(sfm), str(mat))
    assert e.value.code == -4
