#!/usr/bin/env python3
"""Where the time of K3 (F-matrix AC-RANSAC) and K5 (P3P AC-RANSAC) goes, from in-kernel time stamps.
Needs the instrumented build:  make -C sfmlocalization_amd/csrc EXTRA=-DSFMLOC_STAMPS OBJDIR=../build_stamps \
                                    OUT=../lib/libsfmloc_hip_stamps.so ../lib/libsfmloc_hip_stamps.so
and SFMLOC_LIB_PATH pointing at it (tools/run_stamps.sh does both)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synth_bow)
import sfmlocalization_amd as S  # noqa: E402
import synthdata as synth  # noqa: E402
from sfmlocalization_amd import _lib  # noqa: E402


def main():
    views = int(os.environ.get("VIEWS", "10000"))
    L = _lib.load()
    L.sfmloc_debug_stamps_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_ulonglong]
    m = synth.make_map(2, n_views=views, desc_per_view=2000)
    queries = [synth.make_query(m, 1000 + i, n_feat=2000) for i in range(8)]
    bow, qbow = bench.synth_bow(m, queries)
    params = S.default_params(device=0, profile=0, ransac_round=25)
    dm = S.Map(m.view_id, m.view_off, m.desc, params=params, view_wh=m.view_wh, kpt_xy=m.kpt_xy,
               row_landmark=m.row_landmark, landmark_id=m.landmark_id, landmark_X=m.landmark_X, intrinsic=m.intrinsic,
               bow=bow)
    dqs = [dm.query(q.desc, q.kpt_xy, q.width, q.height) for q in queries]
    for dq, qb in zip(dqs, qbow):
        dq.set_bow(qb)
    c = dm.context()
    for dq in dqs:  # warm
        c.begin_bow(dq, None, 100)
        c.end()
    for qi, dq in enumerate(dqs[:4]):
        L.sfmloc_debug_stamps_clear()
        c.begin_bow(dq, None, 100)
        pose, _, _ = c.end()
        p3p = np.zeros(16 * 256 * 8, np.uint64)
        sel = np.zeros(16 * 8, np.uint64)
        f = np.zeros(256 * 64, np.uint64)
        L.sfmloc_debug_stamps_read(0, p3p.ctypes.data, p3p.size)
        L.sfmloc_debug_stamps_read(1, sel.ctypes.data, sel.size)
        L.sfmloc_debug_stamps_read(2, f.ctypes.data, f.size)
        p3p = p3p.reshape(16, 256, 8).astype(np.int64)
        sel = sel.reshape(16, 8).astype(np.int64)
        print(f"== query {qi}: ok={pose.ok} inliers={pose.n_inliers} iterations={pose.iterations}")
        # ---- K5: per round, the block that finished last (it sets the round's duration) and the median block
        us = lambda t: t / 100.0
        t_first = None
        for r in range(16):
            live = p3p[r, :, 0] > 0
            if not live.any():
                break
            blk = p3p[r][live]
            start = blk[:, 0].min()
            if t_first is None:
                t_first = start
            end = blk[:, 5].max()
            last = blk[np.argmax(blk[:, 5])]
            # points: 0 entry, 1 sampled+loaded, 2 prepared (quartic), 3 models, 6 residuals, 7 sorted, 4 nfa+reduce, 5 end
            def seg(b):
                o = [b[1] - b[0], b[2] - b[1], b[3] - b[2], b[6] - b[3], b[7] - b[6], b[4] - b[7], b[5] - b[4]]
                return " ".join(f"{us(x):5.1f}" for x in o)
            med = np.median(blk, axis=0)
            prep = (blk[:, 2] - blk[:, 1]) / 100.0
            tot = (blk[:, 5] - blk[:, 0]) / 100.0
            dist = (f" | prepare over the blocks: p10 {np.percentile(prep, 10):.1f} median {np.median(prep):.1f} "
                    f"p90 {np.percentile(prep, 90):.1f} max {prep.max():.1f}; whole block: median {np.median(tot):.1f} "
                    f"max {tot.max():.1f}")
            s0, s3 = sel[r, 0], sel[r, 3] if sel[r, 3] > 0 else sel[r, 2]
            print(f"  K5 round {r:2d}: {live.sum():3d} blocks, t0 {us(start - t_first):7.1f} us, span {us(end - start):5.1f} us "
                  f"| slowest block [sample prep model resid sort nfa write] {seg(last)} | entry spread "
                  f"{us(blk[:, 0].max() - start):4.1f} | select: starts {us(s0 - end):4.1f} after, lasts {us(s3 - s0):4.1f}"
                  + dist)
        # ---- K3: per view timeline
        tags = (f >> np.uint64(48)).astype(np.int64).reshape(256, 64)
        tim = (f & np.uint64(0xFFFFFFFFFFFF)).astype(np.int64).reshape(256, 64)
        starts = [tim[b, 0] for b in range(256) if tags[b, 0] == 1]
        if not starts:
            continue
        t0 = min(starts)
        rows = []
        for b in range(256):
            if tags[b, 0] != 1:
                continue
            n = int((tags[b] > 0).sum())
            if n <= 2:
                continue
            ev = [(int(tags[b, k]), us(tim[b, k] - t0)) for k in range(n)]
            rows.append((ev[-1][1], b, ev))
        rows.sort(reverse=True)
        print(f"  K3: {len(rows)} active views; the 4 slowest (tag@us: 1 entry 2 prelude 3 solves 4 eval-round 5 batch-closed "
              f"6 seq-solved 7 seq-done 8 loop-end 9 exit):")
        for end, b, ev in rows[:4]:
            print(f"    wg {b:3d} end {end:6.1f}: " + " ".join(f"{t}@{x:.0f}" for t, x in ev))
        ends = np.array([r[0] for r in rows])
        print(f"    end times: median {np.median(ends):.0f} us, p90 {np.percentile(ends, 90):.0f}, max {ends.max():.0f}")


if __name__ == "__main__":
    main()
