"""GPU: K9 AKAZE + M-LDB against the CPU restatement -- scale space, keypoints and descriptor bits exact."""
import numpy as np
import pytest

import sfmlocalization_amd as S
import synthdata as synth

pytestmark = pytest.mark.gpu


def bits32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("shape,seed", [((480, 640), 1), ((300, 300), 2), ((270, 481), 3), ((1080, 1920), 4)])
def test_detect_and_compute_matches_oracle(oracle_c, shape, seed):
    g = synth.texture_image(seed, *shape)
    ekp, edesc, eldet, elt = oracle_c.akaze_detect_and_compute(g, want_levels=True)
    ak = S.Akaze(shape[1], shape[0])
    assert [tuple(x) for x in oracle_c.akaze_levels(shape[1], shape[0])] == ak.levels
    kp, desc = ak.detect_and_compute(g)
    ldet, lt = ak.read_levels()
    np.testing.assert_array_equal(bits32(lt), bits32(elt), err_msg="nonlinear scale space Lt")
    np.testing.assert_array_equal(bits32(ldet), bits32(eldet), err_msg="Hessian determinant response")
    assert len(kp) == len(ekp) and len(kp) > 10
    np.testing.assert_array_equal(bits32(kp), bits32(ekp), err_msg="keypoints (x, y, size, angle, response, level)")
    np.testing.assert_array_equal(desc[:, :61], edesc)
    assert (desc[:, 61:] == 0).all()
    ak.close()


def dense_image(seed, shape, block):
    """Random two-level blocks of a few pixels: thousands of extrema per level, many within each other's radius."""
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    h, w = shape
    small = rng.integers(0, 2, ((h + block - 1) // block, (w + block - 1) // block))
    return (30 + 195 * np.kron(small, np.ones((block, block), np.int64))[:h, :w]).astype(np.uint8)


@pytest.mark.parametrize("form", ["", "spill", "global", "spill,global"])
def test_duplicate_suppression_forms_on_dense_frames(oracle_c, form, monkeypatch):
    """k_suppress decides out of neighbour lists (k_suppress_nbr) in LDS; a candidate whose lists do not fit its record is
    decided by band scans, a level above 4 096 candidates out of the global arrays.  Dense frames reach both by themselves
    (asserted), SFMLOC_AKAZE_SUPPRESS forces either for every candidate / level: always the oracle's keypoints, bit for bit."""
    if form:
        monkeypatch.setenv("SFMLOC_AKAZE_SUPPRESS", form)
    else:
        monkeypatch.delenv("SFMLOC_AKAZE_SUPPRESS", raising=False)
    seen = {"spilled_ap": 0, "spilled_n": 0, "global_levels": 0, "candidates": 0}
    for shape, seed, block in (((240, 320), 11, 3), ((300, 400), 12, 5), ((480, 640), 13, 4)):
        g = dense_image(seed, shape, block)
        ekp, edesc = oracle_c.akaze_detect_and_compute(g, cap=60000)[:2]
        ak = S.Akaze(shape[1], shape[0])
        kp, desc = ak.detect_and_compute(g)
        st = ak.suppress_stats()
        ak.close()
        assert st["candidates"] < 65536
        assert len(kp) == len(ekp) and len(kp) > 200
        np.testing.assert_array_equal(bits32(kp), bits32(ekp), err_msg=f"keypoints, form {form!r}, blocks of {block}: {st}")
        np.testing.assert_array_equal(desc[:, :61], edesc)
        for k in seen:
            seen[k] = max(seen[k], st[k])
        if "spill" in form:
            assert st["spilled_ap"] == st["candidates"] and st["spilled_n"] == st["candidates"]
        if "global" in form:
            assert st["global_levels"] >= 3
        print(form, shape, block, st)
    if not form:  # the largest frame has a level above the LDS arrays without being told to
        assert seen["global_levels"] > 0, seen


@pytest.mark.parametrize("shape", [(480, 640), (270, 481)])
def test_batch_of_images_equals_one_at_a_time(shape):
    """sfmloc_akaze_detect_and_compute_batch (one launch per kernel for all the frames, the stages after the extrema taken
    together): keypoints, descriptors and the scale space of every extractor as from separate calls -- with extractors on
    streams of their own and with all of them on one context's stream."""
    imgs = [synth.texture_image(40 + k, *shape) for k in range(7)]
    one = S.Akaze(shape[1], shape[0])
    ref = [one.detect_and_compute(g) for g in imgs]
    ref_levels = one.read_levels()                                   # (of the last image)
    exs = [S.Akaze(shape[1], shape[0]) for _ in range(7)]
    for n in (1, 2, 7):
        got = S.Akaze.detect_and_compute_batch(exs[:n], imgs[:n])
        for (kp, d), (rkp, rd) in zip(got, ref):
            np.testing.assert_array_equal(bits32(kp), bits32(rkp))
            np.testing.assert_array_equal(d, rd)
    ldet, lt = exs[6].read_levels()
    np.testing.assert_array_equal(bits32(lt), bits32(ref_levels[1]))
    np.testing.assert_array_equal(bits32(ldet), bits32(ref_levels[0]))
    assert sum(len(kp) for kp, _ in ref) > 50
    # a worker's extractors and context on ONE stream (bench.py --from-images)
    m = synth.make_map(1, n_views=6, desc_per_view=50, views_per_place=3, landmarks_per_place=40, obs_per_view=20)
    with S.Map(m.view_id, m.view_off, m.desc) as dm:
        c = dm.context()
        for e in exs[:3]:
            e.share_stream(c)
        got = S.Akaze.detect_and_compute_batch(exs[:3], imgs[2:5])
        for (kp, d), (rkp, rd) in zip(got, ref[2:5]):
            np.testing.assert_array_equal(bits32(kp), bits32(rkp))
            np.testing.assert_array_equal(d, rd)
        for e in exs[:3]:
            e.share_stream(None)
        c.close()
    # an extractor that has only worked in sessions so far (it has no stream of its own yet: one is created at the first
    # call that needs it, ordered after the sessions' work) now works alone, then leads a session, then follows again
    kp, d = exs[5].detect_and_compute(imgs[1])
    np.testing.assert_array_equal(bits32(kp), bits32(ref[1][0]))
    np.testing.assert_array_equal(d, ref[1][1])
    got = S.Akaze.detect_and_compute_batch([exs[5], exs[4], exs[6]], imgs[3:6])
    for (kp, d), (rkp, rd) in zip(got, ref[3:6]):
        np.testing.assert_array_equal(bits32(kp), bits32(rkp))
        np.testing.assert_array_equal(d, rd)
    got = S.Akaze.detect_and_compute_batch([exs[4], exs[5]], imgs[0:2])
    for (kp, d), (rkp, rd) in zip(got, ref[0:2]):
        np.testing.assert_array_equal(bits32(kp), bits32(rkp))
        np.testing.assert_array_equal(d, rd)
    other = S.Akaze(shape[1] + 16, shape[0])
    with pytest.raises(S.SfmlocError):                                # one size per batch
        S.Akaze.detect_and_compute_batch([exs[0], other], [imgs[0], synth.texture_image(9, shape[0], shape[1] + 16)])
    with pytest.raises(S.SfmlocError):                                # an extractor takes one image of a batch
        S.Akaze.detect_and_compute_batch([exs[0], exs[0]], imgs[:2])
    for e in exs + [one, other]:
        e.close()


def test_dense_compute_matches_oracle(oracle_c):
    g = synth.texture_image(4, 300, 300)
    xs = np.arange(0, 300, 6, dtype=np.float32)
    kin, size = [], 4.0
    for s in range(4):
        kin += [(x, y, size, s) for y in xs for x in xs]
        size *= 1.5
    kin = np.array(kin, np.float32)
    ak = S.Akaze(300, 300)
    desc, ang = ak.compute(g, kin)
    edesc, eang = oracle_c.akaze_compute(g, kin)
    np.testing.assert_array_equal(bits32(ang), bits32(eang))
    np.testing.assert_array_equal(desc[:, :61], edesc)
    ak.close()


def test_extracted_query_localises(oracle_c):
    """Descriptors extracted by the GPU from an image feed the matcher: an image and its shifted copy match."""
    g = synth.texture_image(5)
    ak = S.Akaze(640, 480)
    kp, desc = ak.detect_and_compute(g)
    kp2, desc2 = ak.detect_and_compute(np.roll(np.roll(g, 6, axis=0), 10, axis=1))
    view_off = np.array([0, len(desc)], np.uint32)
    with S.Map([0], view_off, desc) as m:
        q = m.query(desc2)
        m.match_putative(q)
        cnt, mi, mj, md = m.putative_read()
        n = int(cnt[0])
        assert n > 0.7 * len(kp)
        shift = kp2[mj[:n], :2] - kp[mi[:n], :2]
        assert np.abs(np.median(shift, axis=0) - [10, 6]).max() < 0.05
    ak.close()


def test_batch_of_1080p_frames_equals_one_at_a_time():
    """Two 1920 x 1080 frames per call: the large-image forms of the tail (the segment scan as two launches over many
    compute units, the all-level kernels with four pixels / segments per thread / wave) in their gang forms -- keypoints and
    descriptors as from separate calls, resident outputs as the downloaded ones."""
    shape = (1080, 1920)
    imgs = [synth.texture_image(70 + k, *shape, n_blobs=2500, n_rects=1200) for k in range(2)]
    one = S.Akaze(shape[1], shape[0])
    ref = [one.detect_and_compute(g) for g in imgs]
    exs = [S.Akaze(shape[1], shape[0]) for _ in range(2)]
    got = S.Akaze.detect_and_compute_batch(exs, imgs)
    for (kp, d), (rkp, rd) in zip(got, ref):
        assert len(rkp) > 500
        np.testing.assert_array_equal(bits32(kp), bits32(rkp))
        np.testing.assert_array_equal(d, rd)
    ns = S.Akaze.detect_resident_batch(exs, imgs[::-1])
    assert list(ns) == [len(ref[1][0]), len(ref[0][0])]
    for e in exs + [one]:
        e.close()
