"""AKAZE + M-LDB extraction time per image: the texture of round 1/2 (few hundred keypoints) and the image-world frames
(~2 000 keypoints at VGA, imageworld.py), one image per call and eight per call."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S
import synthdata as synth
import imageworld as iw
import torch

os.environ["SFMLOC_AKAZE_TIMING"] = "0"
dev = torch.device("cuda", 0)
atlas = iw.make_atlas(5, 2, 1600, dev)
rng = np.random.Generator(np.random.PCG64(3))
for (h, w, f) in ((480, 640, 800.0), (1080, 1920, 2000.0)):
    Rs, Cs = iw.cameras(rng, 8, (8.0, 8.0), 16.0)
    rich = iw.render(atlas, 100.0, Rs, Cs, f, w, h)
    plain = [synth.texture_image(1 + k, h, w, n_blobs=int(400 * w * h / 307200), n_rects=int(200 * w * h / 307200)) for k in range(8)]
    for name, imgs in (("plain texture", plain), ("image-world frame", list(rich))):
        aks = [S.Akaze(w, h) for _ in range(8)]
        for _ in range(3):
            kp, d = aks[0].detect_and_compute(imgs[0])
        # (medians of per-call times: a mean over a handful of calls took the host's first touches of the wrappers' output
        # buffers -- 46 MB per batch of eight -- for the extraction's time)
        ts = []
        for k in range(20):
            t = time.perf_counter()
            kp, d = aks[0].detect_and_compute(imgs[k % 8])
            ts.append(time.perf_counter() - t)
        dt1 = float(np.median(ts))
        for _ in range(3):
            S.Akaze.detect_and_compute_batch(aks, imgs)
        ts = []
        for k in range(9):
            t = time.perf_counter()
            out = S.Akaze.detect_and_compute_batch(aks, imgs)
            ts.append(time.perf_counter() - t)
        dt8 = float(np.median(ts)) / 8
        print(json.dumps({"image": f"{w}x{h}", "content": name, "keypoints": len(kp), "ms_per_image_one_per_call": round(dt1 * 1e3, 3),
                          "ms_per_image_eight_per_call": round(dt8 * 1e3, 3)}), flush=True)
        for a in aks:
            a.close()
