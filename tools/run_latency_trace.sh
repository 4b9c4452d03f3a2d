mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lat -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 16 --in-flight 1 --no-cpu-baseline --no-roofline-phase $BENCH_ARGS > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/prof_lat/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last query: find last k_hamming_screen
import os
first="k_hamming_screen" if "--bow-knn 0" in os.environ.get("BENCH_ARGS","") else "k_bow_dist"
idx=[i for i,r in enumerate(rows) if first in r["Kernel_Name"]]
start=idx[-2]
t0=int(rows[start]["Start_Timestamp"])
prev_end=t0
for r in rows[start:idx[-1]]:
    n=re.sub(r"\(.*","",r["Kernel_Name"].replace("(anonymous namespace)::","")).replace("sfmloc::","").replace("void ","")
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{n[:28]:28s} start {(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us  gap {(s-prev_end)/1e3:6.1f}  grid {r['Grid_Size_X']}")
    prev_end=e
PY
