import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfmlocalization_amd as S
import synthdata as synth
for (h, w) in ((480, 640), (1080, 1920)):
    g = synth.texture_image(1, h, w, n_blobs=int(400 * w * h / 307200), n_rects=int(200 * w * h / 307200))
    ak = S.Akaze(w, h)
    for _ in range(3):
        kp, d = ak.detect_and_compute(g)
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        kp, d = ak.detect_and_compute(g)
    dt = (time.perf_counter() - t) / n
    print(json.dumps({"image": f"{w}x{h}", "keypoints": len(kp), "ms_per_image": round(dt * 1e3, 3)}), flush=True)
    ak.close()
