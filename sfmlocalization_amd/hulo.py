"""The callers' side of the localiser, as the reference's Python orchestration drives it (SURVEY 8b "Callers").

The reference's Python never computes anything on this path: it builds an argument string from its parameter classes,
shells out to OpenMVGLocalization_AKAZE / ExtFeatAndMatch and reads the files they leave behind.  This module holds
exactly that much -- the path's defaults, the argument sets and the consumers of the result files -- so that a pipeline
can call the MI355X path in process (engine.main over the ctypes C ABI) or as the drop-in binaries, and so that the
tests can pin the file contract against what the reference's own readers return (tests/golden/ref_consumers, minted by
tests/golden/make_ref_fixtures.py from the reference's modules).

Reference (relative to /root/reference/PyVisionLocalizeCommon/src unless a directory is given):
  hulo_param/LocalizeParam.py:31-35        locFeatDistRatio, locRansacRound
  hulo_param/ReconstructParam.py:51-75     maxTrackletMatchDistance, minMatchToRetain, extFeatDistRatio, bGuidedMatching*,
                                           extFeatRansacRound; :109-123 locFeatDistRatio, locRansacRound, locSkipFrame
  hulo_bow/LocalizeBOWParam.py:34, hulo_bow/ReconstructBOWParam.py:32   locKNNnum
  hulo_sfm/sfmMergeGraph.py:239-252, hulo_bow/sfmMergeGraphBOW.py:166-182,
  PyEvaluateAccuracy/src/localizeGlobalCoordinate.py:103-159             the localiser's argument sets
  PyReconstruct/src/reconstructGraph.py:153-163                           ExtFeatAndMatch's argument set
  hulo_sfm/mergeSfM.py:50-66 (readMatch), hulo_sfm/sfmMergeGraph.py:255-269 (center.txt),
  hulo_file/FileUtils.py:37-41 (loadjson), :117-147 (loadBinMat)          the consumers
"""
import json
import os
import struct
import subprocess

import numpy as np

_BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")
LOCALIZE_PROJECT = "OpenMVGLocalization_AKAZE"          # ReconstructParam.py:35
EXTRACT_FEATURE_MATCH_PROJECT = "ExtFeatAndMatch"        # ReconstructParam.py:34
LOCALIZE_PROJECT_PATH = os.path.join(_BIN, LOCALIZE_PROJECT)
EXTRACT_FEATURE_MATCH_PROJECT_PATH = os.path.join(_BIN, EXTRACT_FEATURE_MATCH_PROJECT)


class LocalizeParam:                                     # hulo_param/LocalizeParam.py:23-35
    locFeatDistRatio = 0.6
    locRansacRound = 25


class ReconstructParam:                                  # hulo_param/ReconstructParam.py (the path's members only)
    maxTrackletMatchDistance = 10
    minMatchToRetain = 30
    extFeatDistRatio = 0.7
    bGuidedMatching = True
    bGuidedMatchingLocalize = False
    extFeatRansacRound = 500
    locFeatDistRatio = 0.6
    locRansacRound = 25
    locSkipFrame = 3


class LocalizeBOWParam:                                  # hulo_bow/LocalizeBOWParam.py:34
    locKNNnum = 20


class ReconstructBOWParam:                               # hulo_bow/ReconstructBOWParam.py:32
    locKNNnum = 100


def localize_args(input_dir, sfm_data_dir, matches_dir, out_dir, param=LocalizeParam, bow_param=None,
                  bow_file=None, pca_file=None, skip_frame=None, guided=None):
    """The localiser's argument list as the reference's callers build it: `-f -r` always, `-i` from mergeOneModel
    (sfmMergeGraph.py:250), `-k -a -p` on the BoW variants (sfmMergeGraphBOW.py:179-181,
    localizeGlobalCoordinate.py:113-115), `-gm` when bGuidedMatchingLocalize (sfmMergeGraph.py:240-242).  Numbers are
    formatted with str(), as the reference concatenates them."""
    args = [input_dir, sfm_data_dir, matches_dir, out_dir,
            "-f=" + str(param.locFeatDistRatio), "-r=" + str(param.locRansacRound)]
    if skip_frame is not None:
        args.append("-i=" + str(skip_frame))
    if bow_param is not None:
        bow_file = os.path.join(matches_dir, "BOWfile.yml") if bow_file is None else bow_file
        pca_file = os.path.join(matches_dir, "PCAfile.yml") if pca_file is None else pca_file
        args += ["-k=" + str(bow_param.locKNNnum), "-a=" + bow_file, "-p=" + pca_file]
    if getattr(param, "bGuidedMatchingLocalize", False) if guided is None else guided:
        args.append("-gm")
    return args


def extfeat_args(matches_dir, param=ReconstructParam):
    """ExtFeatAndMatch's argument list (reconstructGraph.py:156-163)."""
    args = [matches_dir, "-mf=" + str(param.maxTrackletMatchDistance), "-mm=" + str(param.minMatchToRetain),
            "-f=" + str(param.extFeatDistRatio), "-r=" + str(param.extFeatRansacRound)]
    if param.bGuidedMatching:
        args.append("-gm")
    return args


def localize_images(input_dir, sfm_data_dir, matches_dir, out_dir, in_process=True, **kw):
    """mergeOneModel's step "localize the images from model2 on model1" (sfmMergeGraph.py:232-252): run the localiser
    over a folder of images, one <base>.json per image in out_dir.  in_process: engine.main over the C ABI in this
    process (the map is loaded once and stays on the GPU for the folder); otherwise the drop-in binary, as os.system
    runs the reference's.  Returns the tool's exit status (the reference ignores it)."""
    args = localize_args(input_dir, sfm_data_dir, matches_dir, out_dir, **kw)
    if in_process:
        from . import engine
        return engine.main(args)
    return subprocess.call([LOCALIZE_PROJECT_PATH] + args)


def load_json(path):
    """FileUtils.loadjson (FileUtils.py:37-41)."""
    with open(path) as fh:
        return json.load(fh)


def read_match(loc_folder):
    """mergeSfM.readMatch (mergeSfM.py:50-66): the result files of a folder in sorted order; a file counts iff it has
    the key "t"; -> (image base names, per-image [[query feature, landmark id], ...])."""
    names, pairs = [], []
    for entry in sorted(os.listdir(loc_folder)):
        if entry[-4:] != "json":          # (the reference's test: also takes "xjson"; kept)
            continue
        result = load_json(os.path.join(loc_folder, entry))
        if "t" in result:
            names.append(os.path.basename(result["filename"]))
            pairs.append(result["pair"])
    return names, pairs


def write_center_txt(loc_folder):
    """sfmMergeGraph.py:255-269: centres of the localised frames as "x y z 255 0 0" lines in <loc_folder>/center.txt,
    formatted with str() as there; -> (frames with a result file, frames localised)."""
    results = [load_json(os.path.join(loc_folder, e)) for e in sorted(os.listdir(loc_folder)) if e[-4:] == "json"]
    centres = [r["t"] for r in results if "t" in r]
    with open(os.path.join(loc_folder, "center.txt"), "w") as fh:
        for c in centres:
            fh.write(" ".join(str(v) for v in c[:3]) + " 255 0 0\n")
    return len(results), len(centres)


_CV_DTYPES = {0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}


def load_bin_mat(path):
    """FileUtils.loadBinMat (FileUtils.py:117-147): `i32 rows, i32 cols, i32 cvType`, then rows x cols values (what
    FileUtils.cpp:43-60 writes, e.g. a view's .bow vector)."""
    with open(path, "rb") as fh:
        row = struct.unpack("<i", fh.read(4))[0]
        if row == 0:
            return np.zeros((0, 0))
        col, mattype = struct.unpack("<ii", fh.read(8))
        if mattype not in _CV_DTYPES:
            raise ValueError("invalid mat type : " + str(mattype))
        return np.fromfile(fh, _CV_DTYPES[mattype]).reshape(row, col)
