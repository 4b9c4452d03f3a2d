"""k_suppress from the inside (SFMLOC_AKAZE_TIMING): per level candidates, rounds and in-kernel clocks, on image-world
frames and plain textures at VGA and 1080p; the lines come on stderr from the library."""
import os, sys
os.environ["SFMLOC_AKAZE_TIMING"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S
import synthdata as synth
import imageworld as iw
import torch

dev = torch.device("cuda", 0)
atlas = iw.make_atlas(5, 2, 1600, dev)
rng = np.random.Generator(np.random.PCG64(3))
for (h, w, f) in ((480, 640, 800.0), (1080, 1920, 2000.0)):
    Rs, Cs = iw.cameras(rng, 2, (8.0, 8.0), 16.0)
    rich = iw.render(atlas, 100.0, Rs, Cs, f, w, h)
    plain = [synth.texture_image(1, h, w, n_blobs=int(400 * w * h / 307200), n_rects=int(200 * w * h / 307200))]
    ak = S.Akaze(w, h)
    for name, img in (("plain", plain[0]), ("rich", rich[0])):
        for k in range(3):
            sys.stderr.write(f"== {w}x{h} {name} call {k}\n")
            sys.stderr.flush()
            ak.detect_and_compute(img)
    ak.close()
