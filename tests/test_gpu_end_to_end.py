"""GPU: image in, pose out.  A textured plane rendered from known camera poses; the map is built from the AKAZE
features the GPU extracts from the rendered views (every keypoint is a landmark on the plane), written to disk in
the reference's file contract; a query IMAGE from a new pose is localised through the command line
(extraction -> matching -> F-matrix filter -> P3P) and lands on the true pose."""
import json
import os

import numpy as np
import pytest

import sfmlocalization_amd as S
from sfmlocalization_amd import engine, fileio
import synthdata as synth

pytestmark = pytest.mark.gpu
F, W, H, PPM = 800.0, 640, 480, 100.0


def test_image_to_pose(tmp_path):
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(3))
    tex = synth.texture_image(7, 1600, 1600, n_blobs=3000, n_rects=900)       # 16 m x 16 m plane, 1 cm texels
    centres = [(x, y) for x in (5.5, 8.0, 10.5) for y in (5.5, 8.0, 10.5)]
    ak = S.Akaze(W, H)
    views, view_off, desc_all, kp_all, X_all = [], [0], [], [], []
    for cxy in centres:
        R, C = synth.plane_camera(rng, cxy, 10.0, tilt=0.15)
        img = synth.render_plane_view(tex, PPM, R, C, F, W, H)
        kp, desc = ak.detect_and_compute(img)
        views.append((R, C, img))
        desc_all.append(desc)
        kp_all.append(synth.round6(kp[:, :2]))
        X_all.append(synth.backproject_to_plane(kp[:, :2].astype(np.float64), R, C, F, W, H))
        view_off.append(view_off[-1] + len(desc))
    n_rows = view_off[-1]
    assert n_rows > 9 * 120, n_rows
    m = synth.SynthMap(view_id=np.arange(9, dtype=np.uint32), view_off=np.array(view_off, np.uint32),
                       view_wh=np.tile(np.array([[W, H]], np.uint32), (9, 1)), desc=np.concatenate(desc_all),
                       kpt_xy=np.concatenate(kp_all), row_landmark=np.arange(n_rows, dtype=np.int32),
                       landmark_id=np.arange(n_rows, dtype=np.uint32) + 1000, landmark_X=np.concatenate(X_all),
                       landmark_desc=np.zeros((0, 64), np.uint8), landmark_place=np.zeros(n_rows, np.int64),
                       view_place=np.zeros(9, np.int64), view_R=np.stack([v[0] for v in views]),
                       view_C=np.stack([v[1] for v in views]), place_center=np.zeros((1, 3)),
                       intrinsic=(F, W / 2.0, H / 2.0), width=W, height=H)
    synth.write_map_to_disk(m, str(tmp_path / "sfm"), str(tmp_path / "matches"))
    qdir = tmp_path / "q"
    qdir.mkdir()
    truth = {}
    for k in range(3):
        R, C = synth.plane_camera(rng, (6.5 + 1.2 * k, 9.0 - k), 9.5 + 0.6 * k, tilt=0.12)
        Image.fromarray(synth.render_plane_view(tex, PPM, R, C, F, W, H)).save(qdir / f"query{k}.png")
        truth[f"query{k}"] = (R, C)
    out = tmp_path / "out"
    rc = engine.main([str(qdir), str(tmp_path / "sfm"), str(tmp_path / "matches"), str(out), "-r=25"])
    assert rc == 0
    n_ok = 0
    for name, (R, C) in truth.items():
        d = json.load(open(out / (name + ".json")))
        if "t" not in d:
            continue
        n_ok += 1
        assert np.abs(np.array(d["t"]) - C).max() < 0.15, (name, d["t"], C)      # metres; camera ~10 m above a PLANAR scene
        assert np.abs(np.array(d["R"]) - R).max() < 0.02
        assert len(d["pair"]) >= 11
    assert n_ok >= 2
    ak.close()
    # the same PNG queries through the C++ host program: its own PNG decoder + GPU AKAZE + localisation
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfmlocalization_amd", "bin",
                       "OpenMVGLocalization_AKAZE")
    out2 = tmp_path / "out_cc"
    r = subprocess.run([cli, str(qdir), str(tmp_path / "sfm"), str(tmp_path / "matches"), str(out2), "-r=25"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Extract features from query image") == 3
    for name in truth:
        assert (out2 / (name + ".json")).read_bytes() == (out / (name + ".json")).read_bytes(), name
    # the same views as colour JPEGs (what a phone sends): both programs decode them with sfmloc_image_read -- the Y
    # plane, as imread(IMREAD_GRAYSCALE) gets it from libjpeg -- and still find the poses
    jdir = tmp_path / "qj"
    jdir.mkdir()
    for k, (name, (R, C)) in enumerate(truth.items()):
        g = synth.render_plane_view(tex, PPM, R, C, F, W, H)
        Image.fromarray(np.stack([g, g, g], 2)).save(jdir / f"{name}.jpg", quality=92, progressive=(k == 1))
    outj, outj2 = tmp_path / "out_jpg", tmp_path / "out_jpg_cc"
    assert engine.main([str(jdir), str(tmp_path / "sfm"), str(tmp_path / "matches"), str(outj), "-r=25"]) == 0
    r = subprocess.run([cli, str(jdir), str(tmp_path / "sfm"), str(tmp_path / "matches"), str(outj2), "-r=25"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    n_ok = 0
    for name, (R, C) in truth.items():
        assert (outj2 / (name + ".json")).read_bytes() == (outj / (name + ".json")).read_bytes(), name
        d = json.load(open(outj / (name + ".json")))
        if "t" in d:
            n_ok += 1
            assert np.abs(np.array(d["t"]) - C).max() < 0.2, (name, d["t"], C)
    assert n_ok >= 2
    # the server's entry (localizeImage.cc:463): imread(IMREAD_COLOR) -> LocalizeEngine::localize, where cv::AKAZE
    # converts BGR2GRAY itself; the Python and the header-only C++ class give the same twelve numbers
    eng = engine.LocalizeEngine(str(tmp_path / "sfm"), str(tmp_path / "matches"), None, 0.6, 25, 4.0, False, 0, 0)
    smoke = os.path.join(os.path.dirname(cli), "engine_smoke")
    n_ok = 0
    for name, (R, C) in truth.items():
        res, ex = eng.localize_file(str(jdir / f"{name}.jpg"))
        r = subprocess.run([smoke, str(tmp_path / "sfm"), str(tmp_path / "matches"), "-", "--image", str(jdir / f"{name}.jpg")],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        first = r.stdout.strip().split("\n")[0]
        assert ([] if first == "FAIL" else [float(x) for x in first.split()]) == list(res), name
        if res:
            n_ok += 1
            assert np.abs(np.array(res[0:3]) - C).max() < 0.2
    assert n_ok >= 2
    # the server's whole chain (execLocalizeImage, localizeImage.cc:133-200): the user's K / dist files -> undistort
    # (GPU) -> crop to the valid region -> localize.  A distortion-free user camera: the undistorted image is the
    # input resampled by a fraction of a pixel, and the pose stays where it was.
    fileio.write_cv_yaml(tmp_path / "K.yml", {"K": np.array([[F, 0, W / 2.0], [0, F, H / 2.0], [0, 0, 1.0]])})
    fileio.write_cv_yaml(tmp_path / "dist.yml", {"dist": np.zeros((1, 5))})
    cam = engine.UserCamera(str(tmp_path / "K.yml"), str(tmp_path / "dist.yml"), W, H)
    assert cam.undistorter.roi[2] >= W - 2 and cam.undistorter.roi[3] >= H - 2
    n_ok = 0
    for name, (R, C) in truth.items():
        res, ex = engine.exec_localize_image(eng, cam, S.image_read(str(jdir / f"{name}.jpg"), color=True))
        if res:
            n_ok += 1
            assert np.abs(np.array(res[0:3]) - C).max() < 0.3
    assert n_ok >= 2
    cam.close()
    eng.close()
    # BoW shortlist computed from the query IMAGE (-k -a -p): dense AKAZE -> PCA -> BoF in both programs; the map's
    # .bow files hold what the same chain gives for the map images, so the shortlist is meaningful
    rngb = np.random.Generator(np.random.PCG64(21))
    K, npca = 12, 10
    pca = {"DimPCA": npca, "EigenVectorsPCA": rngb.normal(size=(61, 61)).astype(np.float32),
           "EigenValuesPCA": rngb.uniform(0.5, 4.0, (61, 1)).astype(np.float32),
           "MeanPCA": rngb.uniform(0, 255, (1, 61)).astype(np.float32)}
    bowm = {"ResizedImageSize": 300, "UseSpatialPyramid": 1, "PyramidLevel": 2, "NormBofFeatureType": "L1",
            "Centers": (rngb.normal(size=(K, npca)) * 30).astype(np.float32)}
    fileio.write_cv_yaml(tmp_path / "matches" / "PCAfile.yml", pca)
    fileio.write_cv_yaml(tmp_path / "matches" / "BOWfile.yml", bowm)
    db = engine.DenseBow(str(tmp_path / "matches" / "BOWfile.yml"), str(tmp_path / "matches" / "PCAfile.yml"))
    for k, (R, C, img) in enumerate(views):
        vec = db.compute(np.stack([img, img, img], 2))
        fileio.write_mat_bin(tmp_path / "matches" / (f"img{k:06d}.bow"), vec.reshape(-1, 1))
    db.close()
    common = [str(qdir), str(tmp_path / "sfm"), str(tmp_path / "matches")]
    opts = ["-r=25", "-k=4", "-a=" + str(tmp_path / "matches" / "BOWfile.yml"), "-p=" + str(tmp_path / "matches" / "PCAfile.yml")]
    out3, out4 = tmp_path / "out_bow_py", tmp_path / "out_bow_cc"
    assert engine.main(common + [str(out3)] + opts) == 0
    r = subprocess.run([cli] + common + [str(out4)] + opts, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    n_loc = 0
    for name in truth:
        a, b = (out4 / (name + ".json")).read_bytes(), (out3 / (name + ".json")).read_bytes()
        assert a == b, name
        n_loc += b'"t"' in b
    assert n_loc >= 1
