"""GPU: the localiser's result files are, byte for byte, the ones the reference's own consumers were shown.

tests/golden/ref_consumers/loc_cli/*.json were written by bin/OpenMVGLocalization_AKAZE on an MI355X for the scene of
tests/consumer_scene.py and then read, in the build container, by the reference's mergeSfM.readMatch and
FileUtils.loadjson (tests/golden/make_ref_fixtures.py -> expected.json; tests/test_ref_pins.py checks that half).
Here the C++ tool, its Python mirror and the orchestration entry point (hulo.localize_images, the argument set of
sfmMergeGraph.py:243-252) regenerate those files: same bytes => what the reference's consumers returned still holds."""
import json
import os
import subprocess

import pytest

from sfmlocalization_amd import hulo
import consumer_scene as scene

pytestmark = pytest.mark.gpu

RC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_consumers")


def committed(base):
    with open(os.path.join(RC, "loc_cli", base + ".json"), "rb") as fh:
        return fh.read()


def test_tools_reproduce_the_files_the_reference_consumers_read(tmp_path, monkeypatch):
    scene.build(tmp_path)
    monkeypatch.chdir(tmp_path)
    # (1) the drop-in binary, as os.system runs it, with the orchestration's argument set
    a = hulo.localize_args(*scene.REL_ARGS)
    assert a[4:] == ["-f=0.6", "-r=25"]
    r = subprocess.run([hulo.LOCALIZE_PROJECT_PATH] + a, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for base in scene.QUERY_BASES:
        assert (tmp_path / "loc" / (base + ".json")).read_bytes() == committed(base), base
    # (2) in process, through the ctypes C ABI
    assert hulo.localize_images("queries", "sfm", "matches", "loc_py") == 0
    for base in scene.QUERY_BASES:
        assert (tmp_path / "loc_py" / (base + ".json")).read_bytes() == committed(base), base
    # (3) the consumers' view of it, against what the REFERENCE'S readMatch returned for the committed files
    with open(os.path.join(RC, "expected.json")) as fh:
        exp = json.load(fh)["loc_cli"]["readMatch"]
    names, pairs = hulo.read_match("loc")
    assert names == exp["imgname"] and pairs == exp["matchlist"]
    assert hulo.write_center_txt("loc_py") == (4, 3)
