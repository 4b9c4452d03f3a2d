# per-queue timeline (one line per launch) of the last 28 ms of an emulated rank of 8 -> gpurun_out/rank8_queues.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_rank
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rank -- python3 $GRAFT_REPO_ROOT/tools/rank_emulation.py --of 8 --steps 8 --warmup 2 --gang 32 --in-flight 96 > /dev/null 2>&1
T=$(find /tmp/prof_rank -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT; python3 tools/trace_queues_window.py $T -45 28 > gpurun_out/rank8_queues.txt; wc -l gpurun_out/rank8_queues.txt
