# kernel trace of one emulated rank of N (default 8) of the sharded path: kernel time per batch of 256 and occupancy
# -> gpurun_out/rank_trace_batches_N.txt
N=${N:-8}; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_rank
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rank -- python3 $GRAFT_REPO_ROOT/tools/rank_emulation.py --of $N --steps 8 --warmup 2 ${EXTRA:---gang 32 --in-flight 96} > $GRAFT_REPO_ROOT/gpurun_out/rank_trace_run.log 2>&1 || { tail -n 20 $GRAFT_REPO_ROOT/gpurun_out/rank_trace_run.log; exit 1; }
T=$(find /tmp/prof_rank -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python3 tools/trace_batches.py $T "P3pFinish|k_p3p_finish" $((256 / N)) 4 > gpurun_out/rank_trace_batches_$N.txt; cat gpurun_out/rank_trace_batches_$N.txt
