"""20 extractions of one image-world VGA frame (for rocprofv3 --kernel-trace --stats: per-kernel time of K9)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S
import imageworld as iw
import torch
w, h = (1920, 1080) if len(sys.argv) > 1 and sys.argv[1] == "1080p" else (640, 480)
atlas = iw.make_atlas(5, 2, 1600, torch.device("cuda", 0))
rng = np.random.Generator(np.random.PCG64(3))
Rs, Cs = iw.cameras(rng, 1, (8.0, 8.0), 16.0)
g = iw.render(atlas, 100.0, Rs, Cs, 800.0 * w / 640, w, h)[0]
a = S.Akaze(w, h)
for _ in range(20):
    kp, d = a.detect_and_compute(g)
print(len(kp), "keypoints")
