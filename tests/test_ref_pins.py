"""CPU: what the reference itself can pin of this path -- its defaults and its file contract.

tests/golden/ref_params.json and tests/golden/ref_consumers/ were minted in the build container by
tests/golden/make_ref_fixtures.py FROM THE REFERENCE'S OWN MODULES: the parameter classes imported as they are
(hulo_param/LocalizeParam.py:31-35, ReconstructParam.py:51-75,109-123, hulo_bow/*BOWParam.py) and the consumers of the
localiser's output (hulo_file/FileUtils.py:37-41 loadjson, :117-147 loadBinMat; hulo_sfm/mergeSfM.py:50-66 readMatch)
run on files the product wrote.  Pinned here: defaults, argument sets, the .bow / <img>.json file contract.  NOT pinned
(nothing in the reference can): any arithmetic -- OpenCV / OpenMVG are absent, the oracle stays "parity unpinned"."""
import json
import os

import numpy as np

from sfmlocalization_amd import capi, engine, extfeat, fileio, hulo
import consumer_scene as scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RC = os.path.join(GOLD, "ref_consumers")


def ref_params():
    with open(os.path.join(GOLD, "ref_params.json")) as fh:
        return json.load(fh)


def test_defaults_are_the_reference_parameter_classes():
    ref = ref_params()
    for cls in (hulo.LocalizeParam, hulo.ReconstructParam, hulo.LocalizeBOWParam, hulo.ReconstructBOWParam):
        mine = {k: v for k, v in vars(cls).items() if not k.startswith("_")}
        assert mine, cls
        for k, v in mine.items():
            assert k in ref[cls.__name__], (cls.__name__, k)
            assert ref[cls.__name__][k] == v and type(ref[cls.__name__][k]) is type(v), (cls.__name__, k)
    assert hulo.LOCALIZE_PROJECT == ref["ReconstructParam"]["LOCALIZE_PROJECT"]
    assert hulo.EXTRACT_FEATURE_MATCH_PROJECT == ref["ReconstructParam"]["EXTRACT_FEATURE_MATCH_PROJECT"]
    assert os.path.basename(hulo.LOCALIZE_PROJECT_PATH) == os.path.basename(ref["ReconstructParam"]["LOCALIZE_PROJECT_PATH"])
    # the library's own default ratio is the orchestration's (the tool's -f default, localization.cpp:66)
    assert np.float32(capi.default_params().dist_ratio) == np.float32(ref["LocalizeParam"]["locFeatDistRatio"])


def test_argument_sets_built_from_the_reference_values_parse_to_them():
    """The strings the reference concatenates (str() of its parameters, sfmMergeGraph.py:243-252,
    sfmMergeGraphBOW.py:170-185, reconstructGraph.py:156-163) go through the tools' cv::CommandLineParser mirror."""
    ref = ref_params()
    R, LB, RB = ref["ReconstructParam"], ref["LocalizeBOWParam"], ref["ReconstructBOWParam"]
    a = hulo.localize_args("in", "sfm", "m", "out", param=hulo.ReconstructParam, bow_param=hulo.ReconstructBOWParam,
                           skip_frame=hulo.ReconstructParam.locSkipFrame)
    assert a == ["in", "sfm", "m", "out", "-f=" + str(R["locFeatDistRatio"]), "-r=" + str(R["locRansacRound"]),
                 "-i=" + str(R["locSkipFrame"]), "-k=" + str(RB["locKNNnum"]), "-a=m/BOWfile.yml", "-p=m/PCAfile.yml"]
    pos, o = engine.parse_cv_args(a, engine.KEYS)
    assert pos == ["in", "sfm", "m", "out"]
    assert (o["fDistRatio"], o["ransacRound"], o["locEvryNFrame"], o["knnbow"], o["guidedMatch"]) == \
        (R["locFeatDistRatio"], R["locRansacRound"], R["locSkipFrame"], RB["locKNNnum"], R["bGuidedMatchingLocalize"])
    assert (o["bowModelFile"], o["pcaModelFile"]) == ("m/BOWfile.yml", "m/PCAfile.yml")
    b = hulo.localize_args("in", "sfm", "m", "out", bow_param=hulo.LocalizeBOWParam)
    assert engine.parse_cv_args(b, engine.KEYS)[1]["knnbow"] == LB["locKNNnum"]
    c = hulo.localize_args("in", "sfm", "m", "out", guided=True)
    assert c[-1] == "-gm" and engine.parse_cv_args(c, engine.KEYS)[1]["guidedMatch"] is True
    e = hulo.extfeat_args("m")
    assert e == ["m", "-mf=" + str(R["maxTrackletMatchDistance"]), "-mm=" + str(R["minMatchToRetain"]),
                 "-f=" + str(R["extFeatDistRatio"]), "-r=" + str(R["extFeatRansacRound"]), "-gm"]
    pos, o = engine.parse_cv_args(e, extfeat.KEYS)
    assert (o["maxFrameDist"], o["minMatch"], o["fdistratio"], o["ransacround"], o["guidedMatch"]) == \
        (R["maxTrackletMatchDistance"], R["minMatchToRetain"], R["extFeatDistRatio"], R["extFeatRansacRound"],
         R["bGuidedMatching"])


def test_the_committed_files_are_what_the_product_writes(tmp_path):
    scene.write_fileio_results(str(tmp_path / "loc"))
    names = scene.write_bow_files(str(tmp_path / "bow"))
    for f in sorted(os.listdir(os.path.join(RC, "loc_fileio"))):
        assert (tmp_path / "loc" / f).read_bytes() == open(os.path.join(RC, "loc_fileio", f), "rb").read(), f
    assert names == sorted(os.listdir(os.path.join(RC, "bow")))
    for f in names:
        assert (tmp_path / "bow" / f).read_bytes() == open(os.path.join(RC, "bow", f), "rb").read(), f


def test_result_files_as_the_references_consumers_read_them():
    """mergeSfM.readMatch / FileUtils.loadjson on the files written by fileio.py and by bin/OpenMVGLocalization_AKAZE:
    what THEY returned (expected.json) is what the mirrors in hulo.py return, and what the writers were given."""
    with open(os.path.join(RC, "expected.json")) as fh:
        exp = json.load(fh)
    for sub in ("loc_fileio", "loc_cli"):
        d = os.path.join(RC, sub)
        names, pairs = hulo.read_match(d)
        assert names == exp[sub]["readMatch"]["imgname"]
        assert pairs == exp[sub]["readMatch"]["matchlist"]
        files = sorted(f for f in os.listdir(d) if f.endswith(".json"))
        assert files == sorted(exp[sub]["loadjson"])
        for f in files:
            got = hulo.load_json(os.path.join(d, f))
            assert got == exp[sub]["loadjson"][f]
            assert list(got)[:3] == ["filename", "sfm_data", "matches_dir"]        # localization.cpp:84-109
            if "t" in got:                                                         # :111-153
                assert list(got) == ["filename", "sfm_data", "matches_dir", "K", "R", "t", "pair"]
                assert np.array(got["K"]).shape == (3, 3) and np.array(got["R"]).shape == (3, 3) and len(got["t"]) == 3
                assert all(len(p) == 2 and all(isinstance(x, int) for x in p) for p in got["pair"])
            else:
                assert list(got) == ["filename", "sfm_data", "matches_dir"]
    # the fileio files were written from known values: the reference's reader returns them at 6 significant digits
    assert exp["loc_fileio"]["readMatch"]["imgname"] == ["frame0001.jpg", "frame0003.jpg"]
    assert "t" not in exp["loc_fileio"]["loadjson"]["frame0002.json"]
    assert exp["loc_cli"]["readMatch"]["imgname"] == ["q000.jpg", "q001.jpg", "q002.jpg"]     # q003 cannot be localised
    n_files, n_loc = len(exp["loc_cli"]["loadjson"]), len(exp["loc_cli"]["readMatch"]["imgname"])
    assert (n_files, n_loc) == (4, 3)


def test_center_txt_as_the_merge_step_writes_it(tmp_path):
    import shutil
    d = tmp_path / "loc"
    shutil.copytree(os.path.join(RC, "loc_cli"), d)
    assert hulo.write_center_txt(str(d)) == (4, 3)
    with open(os.path.join(RC, "expected.json")) as fh:
        exp = json.load(fh)["loc_cli"]["loadjson"]
    lines = (d / "center.txt").read_text().splitlines()
    want = [" ".join(str(v) for v in exp[f]["t"]) + " 255 0 0" for f in sorted(exp) if "t" in exp[f]]
    assert lines == want


def test_bow_files_as_the_references_loadBinMat_reads_them():
    exp = np.load(os.path.join(RC, "expected_bow.npz"))
    assert sorted(exp.files) == sorted(os.listdir(os.path.join(RC, "bow")))
    for name in exp.files:
        path = os.path.join(RC, "bow", name)
        ref_mat = exp[name]
        for got in (hulo.load_bin_mat(path), fileio.read_mat_bin(path)):
            assert got.dtype == ref_mat.dtype and got.shape == ref_mat.shape
            np.testing.assert_array_equal(got, ref_mat)
    # TrainBoW's own shape and type (TrainBoW.cpp:268): 500 x 1 CV_64F
    assert exp["view_f64_500x1.bow"].shape == (500, 1) and exp["view_f64_500x1.bow"].dtype == np.float64
