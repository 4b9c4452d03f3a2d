"""The last extraction of tools/akaze_trace.py from a rocprofv3 kernel trace: launch, start offset, duration, gap."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "sfmloc" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run of launches that starts with k_pre_rows
starts = [i for i, r in enumerate(rows) if "k_pre_rows" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("sfmloc::(anonymous namespace)::", "").replace("sfmloc::", "").split("(")[0]
    print(f"{(s - t0) / 1e3:8.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {name[:70]}  grid {r.get('Grid_Size_X', '')}x{r.get('Grid_Size_Y', '')} wg {r.get('Workgroup_Size_X', '')}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"total {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, launches {len(rows)}")
