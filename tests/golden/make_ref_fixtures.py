#!/usr/bin/env python3
"""Build-container script (needs /root/reference; never runs on the GPU box): pins what the reference CAN pin of this
path -- its defaults and its file contract -- and writes the result as data fixtures under tests/golden/.

(a) tests/golden/ref_params.json: the reference's parameter classes, imported as they are
    (PyVisionLocalizeCommon/src/hulo_param/LocalizeParam.py, ReconstructParam.py, hulo_bow/LocalizeBOWParam.py,
    ReconstructBOWParam.py: plain Python 3-compatible class bodies), every public attribute.
(b) tests/golden/ref_consumers/: files WRITTEN BY THE PRODUCT -- .bow matrices by fileio.write_mat_bin, result files by
    fileio.write_result_json (loc_fileio/) and by bin/OpenMVGLocalization_AKAZE on the GPU box (loc_cli/, minted by
    tests/golden/make_cli_outputs.py through gpurun) -- and expected.json / expected_bow.npz: what the REFERENCE'S OWN
    consumers return when they read those files: hulo_file/FileUtils.py loadBinMat (:117-147) and loadjson (:37-41),
    hulo_sfm/mergeSfM.py readMatch (:50-66, the `"t" in json` success test).  Those two modules are Python 2 text
    (print statements); they are converted in memory with lib2to3 and executed as modules -- nothing of them is written
    anywhere, the fixtures hold only the product's files and the values returned.

What this pins: the file contract and the defaults.  What it does not: any arithmetic (OpenCV / OpenMVG are absent).

    python tests/golden/make_ref_fixtures.py [--cli-dir gpurun_out/ref_consumers_cli]
"""
import argparse
import importlib.util
import json
import os
import shutil
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/PyVisionLocalizeCommon/src"
OUT = os.path.join(HERE, "ref_consumers")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def import_plain(rel):
    path = os.path.join(REF_SRC, rel)
    spec = importlib.util.spec_from_file_location("ref_" + os.path.basename(rel)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def public_attrs(cls):
    return {k: v for k, v in vars(cls).items() if not k.startswith("_") and isinstance(v, (int, float, str, bool))}


def import_py2(rel, name, package_modules=()):
    """A Python 2 module of the reference, converted in memory (lib2to3) and executed under `name`."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
    with open(os.path.join(REF_SRC, rel)) as fh:
        src = fh.read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    code = str(tool.refactor_string(src if src.endswith("\n") else src + "\n", rel))
    mod = types.ModuleType(name)
    mod.__file__ = os.path.join(REF_SRC, rel)
    for pkg in package_modules:
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    sys.modules[name] = mod
    exec(compile(code, mod.__file__, "exec"), mod.__dict__)
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cli-dir", default=os.path.join(ROOT, "gpurun_out", "ref_consumers_cli"))
    a = ap.parse_args()
    import numpy as np

    # (a) defaults
    params = {}
    for rel, cls in [("hulo_param/LocalizeParam.py", "LocalizeParam"), ("hulo_param/ReconstructParam.py", "ReconstructParam"),
                     ("hulo_bow/LocalizeBOWParam.py", "LocalizeBOWParam"), ("hulo_bow/ReconstructBOWParam.py", "ReconstructBOWParam")]:
        params[cls] = public_attrs(getattr(import_plain(rel), cls))
    with open(os.path.join(HERE, "ref_params.json"), "w") as fh:
        json.dump(params, fh, indent=1, sort_keys=True)
        fh.write("\n")

    # (b) the reference's consumers on the product's files
    futils = import_py2("hulo_file/FileUtils.py", "hulo_file.FileUtils", package_modules=("hulo_file",))
    merge = import_py2("hulo_sfm/mergeSfM.py", "hulo_sfm.mergeSfM", package_modules=("hulo_sfm",))
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    import consumer_scene as scene
    scene.write_fileio_results(os.path.join(OUT, "loc_fileio"))
    if os.path.isdir(a.cli_dir) and os.listdir(a.cli_dir):
        shutil.copytree(a.cli_dir, os.path.join(OUT, "loc_cli"))
    else:
        raise SystemExit(f"{a.cli_dir}: no result files of bin/OpenMVGLocalization_AKAZE; run "
                         "`gpurun -- python tests/golden/make_cli_outputs.py` first")
    expected = {}
    for sub in ("loc_fileio", "loc_cli"):
        d = os.path.join(OUT, sub)
        names, matches = merge.readMatch(d)
        expected[sub] = {"readMatch": {"imgname": names, "matchlist": matches},
                         "loadjson": {f: futils.loadjson(os.path.join(d, f)) for f in sorted(os.listdir(d))}}
    with open(os.path.join(OUT, "expected.json"), "w") as fh:
        json.dump(expected, fh, indent=1, sort_keys=True)
        fh.write("\n")
    bow_names = scene.write_bow_files(os.path.join(OUT, "bow"))
    got = {n: futils.loadBinMat(os.path.join(OUT, "bow", n)) for n in bow_names}
    np.savez(os.path.join(OUT, "expected_bow.npz"), **got)
    print("ref_params.json:", {k: len(v) for k, v in params.items()})
    for sub in expected:
        print(sub, "readMatch ->", expected[sub]["readMatch"]["imgname"])
    print("loadBinMat ->", {k: (v.shape, str(v.dtype)) for k, v in got.items()})


if __name__ == "__main__":
    main()
