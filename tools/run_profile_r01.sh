mkdir -p gpurun_out
./tools/peaks > gpurun_out/peaks.json 2>&1; cat gpurun_out/peaks.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/bench_prof.log | cut -c1-300; find gpurun_out/prof_r01 -name "*stats*" | head
