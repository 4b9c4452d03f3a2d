#!/usr/bin/env python3
"""What the path does on EXTRACTED descriptors (an image world of imageworld.py, small): per frame the path's shape
(views, correspondences, inliers, AC-RANSAC iterations) and its stage times; with the instrumented build
(tools/run_stamps.sh) the per-round time stamps of K5 as tools/k35_stamps.py prints them.
    python tools/image_lab.py [n_real_views] [tiles]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imageworld as iw  # noqa: E402
import sfmlocalization_amd as S  # noqa: E402
from sfmlocalization_amd import _lib  # noqa: E402


def main():
    n_real = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    W, H = 640, 480
    world = iw.build(S, 31, n_real, 8, tiles=tiles, progress=lambda s: print("[lab]", s, file=sys.stderr))
    m = world.m
    L = _lib.load()
    stamps = hasattr(L, "sfmloc_debug_stamps_read")
    if stamps:
        L.sfmloc_debug_stamps_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong]
    for prof in (1, 0):
        params = S.default_params(profile=prof, ransac_round=25)
        dm = S.Map(m.view_id, m.view_off, m.desc, params=params, view_wh=m.view_wh, kpt_xy=m.kpt_xy,
                   row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic)
        ak = S.Akaze(W, H)
        c = dm.context()
        for i, f in enumerate(world.frames):
            kp, d = ak.detect_and_compute(f)
            dq = dm.query(d, kp[:, :2], W, H)
            c.begin(dq)
            c.end()                                   # warm
            if stamps:
                L.sfmloc_debug_stamps_clear()
            t0 = time.perf_counter()
            c.begin(dq)
            pose, pq, pl = c.end()
            dt = (time.perf_counter() - t0) * 1e3
            ss = [x * 1e3 for x in pose.stage_seconds]
            print(f"profile={prof} frame {i}: {len(kp)} feats, views>=16 {pose.n_putative_views}, geo views {pose.n_geometric_views}, "
                  f"2d3d {pose.n_matches_2d3d}, inliers {pose.n_inliers}, iterations {pose.iterations}, ok {pose.ok}, wall {dt:.2f} ms"
                  + (f" | putMatch {ss[3]:.2f} geoMatch {ss[4]:.2f} PnP {ss[5]:.2f} others {ss[6]:.2f}" if prof else ""))
            if stamps and prof == 0:
                p3p = np.zeros(16 * 256 * 8, np.uint64)
                L.sfmloc_debug_stamps_read(0, p3p.ctypes.data, p3p.size)
                p3p = p3p.reshape(16, 256, 8).astype(np.int64)
                t_first = None
                for r in range(16):
                    live = p3p[r, :, 0] > 0
                    if not live.any():
                        break
                    blk = p3p[r][live]
                    start = blk[:, 0].min()
                    t_first = start if t_first is None else t_first
                    end = blk[:, 5].max()
                    med = np.median(blk, axis=0)
                    seg = [med[1] - med[0], med[2] - med[1], med[3] - med[2], med[6] - med[3], med[7] - med[6], med[4] - med[7], med[5] - med[4]]
                    print(f"    K5 round {r:2d}: {int(live.sum()):3d} blocks, t0 {(start - t_first) / 100:7.1f} us, span {(end - start) / 100:6.1f} us"
                          f" | median block [sample prep model resid sort nfa write] " + " ".join(f"{x / 100:5.1f}" for x in seg))
            dq.close()
        c.close()
        ak.close()
        dm.close()


if __name__ == "__main__":
    main()
