"""A window of a rocprofv3 --kernel-trace csv as one line per launch, grouped by hardware queue: start offset, duration,
kernel, grid z (gang members).  usage: trace_queues_window.py trace.csv start_ms length_ms  (start counted back from the
trace's last kernel when negative)"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
t_end = max(int(r["End_Timestamp"]) for r in rows)
t0 = int(rows[0]["Start_Timestamp"])
start = float(sys.argv[2]) * 1e6
a = (t_end + start) if start < 0 else (t0 + start)
b = a + float(sys.argv[3]) * 1e6
by = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= a or s >= b:
        continue
    n = r["Kernel_Name"].replace("void ", "").replace("sfmloc::(anonymous namespace)::", "").replace("sfmloc::", "").split("(")[0]
    g = re.match(r"k_gang<(\w+?)Body", n)
    n = ("gang:" + g.group(1) if g else n)[:28]
    by[r.get("Queue_Id", "?")].append((s, e, n, r.get("Grid_Size_Z", "1")))
for q in sorted(by, key=lambda q: by[q][0][0]):
    print(f"-- queue {q}: {len(by[q])} launches, busy {sum(e - s for s, e, _, _ in by[q]) / 1e6:.2f} ms")
    for s, e, n, z in sorted(by[q]):
        print(f"   {(s - a) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {n} x{z}")
