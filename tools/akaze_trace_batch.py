"""Six extractions of a batch of eight image-world VGA frames (or `1080p`) in one call, for rocprofv3 --kernel-trace:
tools/akaze_trace_report.py prints the last batch's launches with start offsets, durations and gaps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S
import imageworld as iw
import torch
w, h = (1920, 1080) if "1080p" in sys.argv[1:] else (640, 480)
atlas = iw.make_atlas(5, 2, 1600, torch.device("cuda", 0))
rng = np.random.Generator(np.random.PCG64(3))
Rs, Cs = iw.cameras(rng, 8, (8.0, 8.0), 16.0)
imgs = list(iw.render(atlas, 100.0, Rs, Cs, 800.0 * w / 640, w, h))
aks = [S.Akaze(w, h) for _ in range(8)]
for k in range(6):
    t = time.perf_counter()
    out = S.Akaze.detect_and_compute_batch(aks, imgs)
    dt = time.perf_counter() - t
print([len(kp) for kp, _ in out], "keypoints", round(dt * 1e3, 3), "ms for the batch (last call, under the profiler)")
