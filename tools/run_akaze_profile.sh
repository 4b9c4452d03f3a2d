mkdir -p gpurun_out
timeout -k 10 200 python tools/akaze_time.py 2>&1 | tee gpurun_out/akaze_time.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_akaze -- python3 $GRAFT_REPO_ROOT/tools/akaze_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/prof_akaze/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n=re.sub(r"\(.*","",r["Name"].replace("(anonymous namespace)::","")).replace("sfmloc::","").replace("void ","")
    print(f"{n:28s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
