# per-kernel times of one rank of 8 (tools/rank_emulation.py) under rocprofv3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_rank
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rank -o rank -- python3 tools/rank_emulation.py --of 8 --steps 4 --warmup 1 $EXTRA > gpurun_out/prof_rank.log 2>&1 || { tail -20 gpurun_out/prof_rank.log; exit 1; }
grep rank_rate gpurun_out/prof_rank.log | cut -c1-330
f=$(find gpurun_out/prof_rank -name "*kernel_stats.csv" | head -1); python3 tools/trace_queues.py $(find gpurun_out/prof_rank -name "*kernel_trace.csv" | head -1) > gpurun_out/prof_rank_queues.txt
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):7d} total {float(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
