mkdir -p gpurun_out; rm -rf gpurun_out/prof_math
timeout -k 10 300 python -m pytest tests/test_gpu_geom.py -m gpu -q -k "minimal or polynomial" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_math -- python3 $GRAFT_REPO_ROOT/tools/math_latency.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_math/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "debug" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=["log10","sqrt+div","cubic","quartic","seven-point (lane)","P3P","seven-point (wave)"]
for i,r in enumerate(rows):
    print(names[i%7], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us")
PY
