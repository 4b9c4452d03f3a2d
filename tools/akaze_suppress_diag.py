"""k_suppress from the inside (SFMLOC_AKAZE_TIMING): per level candidates, rounds and in-kernel clocks, on image-world
frames and plain textures at VGA and 1080p (`bench1080p`: a frame of the bench's 1080p world -- denser levels); the lines
come on stderr from the library, the counts of sfmloc_akaze_suppress_stats on stdout."""
import os, sys
os.environ["SFMLOC_AKAZE_TIMING"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S
import synthdata as synth
import imageworld as iw
import torch

dev = torch.device("cuda", 0)
rng = np.random.Generator(np.random.PCG64(3))
cases = []
if "bench1080p" in sys.argv[1:]:
    atlas = iw.make_atlas(31, 3, 3840, dev, blobs_per_tile=4500, rects_per_tile=700)
    Rs, Cs = iw.cameras(rng, 2, (24.0, 24.0), 16.0)
    cases.append((1080, 1920, [("bench-1080p-world", f) for f in iw.render(atlas, 240.0, Rs, Cs, 2400.0, 1920, 1080)]))
else:
    atlas = iw.make_atlas(5, 2, 1600, dev)
    for (h, w, f) in ((480, 640, 800.0), (1080, 1920, 2000.0)):
        Rs, Cs = iw.cameras(rng, 2, (8.0, 8.0), 16.0)
        rich = iw.render(atlas, 100.0, Rs, Cs, f, w, h)
        plain = synth.texture_image(1, h, w, n_blobs=int(400 * w * h / 307200), n_rects=int(200 * w * h / 307200))
        cases.append((h, w, [("plain", plain), ("rich", rich[0])]))
for h, w, imgs in cases:
    ak = S.Akaze(w, h)
    for name, img in imgs:
        for k in range(3):
            sys.stderr.write(f"== {w}x{h} {name} call {k}\n")
            sys.stderr.flush()
            kp, _ = ak.detect_and_compute(img)
        print(f"{w}x{h} {name}: {len(kp)} keypoints", ak.suppress_stats(), flush=True)
    ak.close()
