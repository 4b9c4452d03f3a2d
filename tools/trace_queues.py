#!/usr/bin/env python3
"""Per hardware queue of a rocprofv3 kernel trace (CSV): busy time, the gaps by the kernel that follows them, and the
timeline of the busiest queue -- where a stream of dependent launches spends its time.
usage: python tools/trace_queues.py <kernel_trace.csv> [window_ms_from_end=120] [skip_ms_at_end=20] [timeline_rows=60]"""
import csv
import re
import sys
from collections import defaultdict


def name(n):
    g = re.search(r"k_gang<.*?::([A-Za-z0-9]+Body)", n)
    if g:
        return "gang:" + g.group(1)
    m = re.search(r"(k_[a-z0-9_]+)", n)
    return m.group(1) if m else n[:30]


def main():
    path = sys.argv[1]
    w = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    n_rows = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r["Kernel_Name"]), r["Queue_Id"],
                 int(r["Grid_Size_Z"])) for r in csv.DictReader(open(path)))
    t_end = ev[-1][1]
    win = [e for e in ev if t_end - w * 1e6 < e[0] < t_end - skip * 1e6]
    t0 = win[0][0]
    byq = defaultdict(list)
    for e in win:
        byq[e[3]].append(e)
    busiest, most = None, 0
    for q, l in sorted(byq.items()):
        busy = sum(e[1] - e[0] for e in l)
        gaps = defaultdict(float)
        for a, b in zip(l, l[1:]):
            if b[0] > a[1]:
                gaps[b[2]] += b[0] - a[1]
        top = sorted(gaps.items(), key=lambda x: -x[1])[:5]
        print(f"queue {q}: {len(l)} kernels, span {(l[-1][1] - l[0][0]) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms; gaps before: "
              + ", ".join(f"{k} {v / 1e6:.1f}" for k, v in top))
        if busy > most:
            busiest, most = q, busy
    per = defaultdict(lambda: [0, 0])
    for s, e, n, _, _ in win:
        per[n][0] += 1
        per[n][1] += e - s
    for k, v in sorted(per.items(), key=lambda x: -x[1][1])[:14]:
        print(f"  {k:34s} n {v[0]:6d} busy {v[1] / 1e6:8.2f} ms avg {v[1] / v[0] / 1e3:7.1f} us")
    prev = None
    for s, e, n, _, z in byq[busiest][len(byq[busiest]) // 2:][:n_rows]:
        print(f"{(s - t0) / 1e3:10.1f} us dur {(e - s) / 1e3:7.1f} gap {((s - prev) / 1e3 if prev else 0):7.1f} {n} z={z}")
        prev = e


if __name__ == "__main__":
    main()
