"""GPU occupancy of a traced run (rocprofv3 --kernel-trace csv): in the densest window of `marker` kernels -- the timed
region of a throughput run -- the fraction of wall time in which any kernel runs, the mean number running, and per
kernel the summed durations per marker (= per frame / per query).  `marker` is a regular expression over kernel names; a gang
launch of a marker kernel counts as many markers as it has members (grid z).  usage: trace_busy.py trace.csv marker [window_ms]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2]
win_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 300e6
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
rx = re.compile(marker)
marks = []
for r in rows:
    if rx.search(r["Kernel_Name"]):
        z = max(1, int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r.get("Workgroup_Size_Z", 1) or 1)))
        marks += [int(r["Start_Timestamp"])] * z
marks.sort()
best, j = (0, 0), 0
for i, t in enumerate(marks):                       # the window of win_ns with most markers
    while marks[j] < t - win_ns:
        j += 1
    if i - j + 1 > best[0]:
        best = (i - j + 1, j)
n_mark, j0 = best
a = marks[j0]
b = a + win_ns
pts = []
tot = collections.Counter()
cnt = collections.Counter()
for s, e, n in ev:
    if e <= a or s >= b:
        continue
    pts.append((max(s, a), 1))
    pts.append((min(e, b), -1))
    k = n.replace("void ", "").replace("sfmloc::(anonymous namespace)::", "").replace("sfmloc::", "").split("(")[0]
    g = re.match(r"k_gang<(\w+?)Body", k)
    k = ("gang:" + g.group(1) if g else k)[:44]
    tot[k] += min(e, b) - max(s, a)
    cnt[k] += 1
pts.sort()
busy = area = conc = 0
last = a
for t, d in pts:
    if conc > 0:
        busy += t - last
    area += conc * (t - last)
    last = t
    conc += d
print(f"window {win_ns / 1e6:.0f} ms with {n_mark} x {marker}: {win_ns / 1e3 / n_mark:.1f} us of wall time each; some kernel runs "
      f"{100 * busy / win_ns:.1f} % of the time, {area / win_ns:.2f} kernels on average")
for k, v in tot.most_common(28):
    print(f"  {k:46s} {cnt[k] / n_mark:6.1f} launches and {v / 1e3 / n_mark:8.1f} us of kernel time per {marker}")
