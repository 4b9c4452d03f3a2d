"""10 extractions of one plain VGA texture (or `rich`: an image-world frame) one per call, for rocprofv3 --kernel-trace:
tools/akaze_trace_report.py prints the last extraction's launches with start offsets, durations and gaps."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S
import synthdata as synth
rich = len(sys.argv) > 1 and sys.argv[1] == "rich"
w, h = (1920, 1080) if "1080p" in sys.argv[1:] else (640, 480)
if rich:
    import imageworld as iw
    import torch
    atlas = iw.make_atlas(5, 2, 1600, torch.device("cuda", 0))
    rng = np.random.Generator(np.random.PCG64(3))
    Rs, Cs = iw.cameras(rng, 1, (8.0, 8.0), 16.0)
    g = iw.render(atlas, 100.0, Rs, Cs, 800.0 * w / 640, w, h)[0]
else:
    g = synth.texture_image(1, h, w, n_blobs=int(400 * w * h / 307200), n_rects=int(200 * w * h / 307200))
a = S.Akaze(w, h)
import time
for k in range(10):
    t = time.perf_counter()
    kp, d = a.detect_and_compute(g)
    dt = time.perf_counter() - t
print(len(kp), "keypoints", round(dt * 1e3, 3), "ms (last call, under the profiler)")
