"""Kernel time per BATCH of a sharded rank from a rocprofv3 --kernel-trace csv: the window that holds the last `k` batches
(a batch = `per_batch` finishes of the marker kernel, gang members counted), per kernel the summed durations per batch, the
wall time per batch, the share of it some kernel runs and the mean number running.
usage: trace_batches.py trace.csv marker per_batch [k]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rx, per_batch = re.compile(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
marks = []
for r in rows:
    if rx.search(r["Kernel_Name"]):
        z = max(1, int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r.get("Workgroup_Size_Z", 1) or 1)))
        marks += [int(r["End_Timestamp"])] * z
marks.sort()
need = k * per_batch
assert len(marks) > need + per_batch, (len(marks), need)
a, b = marks[-need - 1], marks[-1]
tot, cnt, pts = collections.Counter(), collections.Counter(), []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= a or s >= b:
        continue
    n = r["Kernel_Name"].replace("void ", "").replace("sfmloc::(anonymous namespace)::", "").replace("sfmloc::", "").split("(")[0]
    g = re.match(r"k_gang<(\w+?)Body", n)
    n = ("gang:" + g.group(1) if g else n)[:44]
    tot[n] += min(e, b) - max(s, a)
    cnt[n] += 1
    pts += [(max(s, a), 1), (min(e, b), -1)]
pts.sort()
busy = area = conc = 0
last = a
for t, d in pts:
    if conc > 0:
        busy += t - last
    area += conc * (t - last)
    last = t
    conc += d
print(f"last {k} batches of {per_batch} x {sys.argv[2]}: {(b - a) / 1e6 / k:.2f} ms of wall time per batch; some kernel runs "
      f"{100 * busy / (b - a):.1f} % of the time, {area / (b - a):.2f} kernels on average")
for n, v in tot.most_common(16):
    print(f"  {n:46s} {cnt[n] / k:6.1f} launches and {v / 1e6 / k:7.2f} ms of kernel time per batch")
