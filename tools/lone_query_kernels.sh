# per-kernel median duration over the lone headline queries of a short bench run (one query in flight)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pb2
rocprofv3 --kernel-trace --output-format csv -d /tmp/pb2 -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 64 --in-flight 1 --no-image-in --no-cpu-baseline --no-real-stats --no-roofline-phase > /dev/null 2>&1
python3 - <<PY
import csv, glob, statistics
f = glob.glob("/tmp/pb2/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void ", "").replace("sfmloc::(anonymous namespace)::", "").replace("sfmloc::", "").split("(")[0][:40]
    d.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda x: -sum(x[1]))[:16]:
    print(f"{n:42s} n={len(v):5d} median {statistics.median(v):8.1f} us  min {min(v):7.1f}")
PY
