mkdir -p gpurun_out
rm -rf gpurun_out/prof_lane
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lane -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --in-flight 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/lane_trace.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY' > gpurun_out/lane_timeline.txt
import csv,glob,re
f=glob.glob("gpurun_out/prof_lane/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_hamming_screen" in r["Kernel_Name"]]
start=idx[-6]
t0=int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n=re.sub(r"[<(].*","",r["Kernel_Name"].replace("(anonymous namespace)::","")).replace("sfmloc::","").replace("void ","")
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f"{n[:24]:24s} q{r['Queue_Id']:>3s} start {(s-t0)/1e3:9.1f} end {(e-t0)/1e3:9.1f} dur {(e-s)/1e3:8.1f} grid {r['Grid_Size_X']}")
PY
head -150 gpurun_out/lane_timeline.txt
