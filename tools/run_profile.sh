# Round-2 evidence run: GPU tests, default bench (BASELINE configs[2]), rocprofv3 kernel-trace stats of the same command,
# PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*) each in its own run, latency trace, AKAZE time.  Copy the results into
# profiles/ afterwards (gpurun_out/ is scratch).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1 || exit 1
tail -1 gpurun_out/bench_default.log | cut -c1-200
timeout -k 10 300 python bench.py --views 1000 --bow-knn 0 --no-real-stats > gpurun_out/bench_cfg1.log 2>&1 || exit 1
tail -1 gpurun_out/bench_cfg1.log | cut -c1-200
timeout -k 10 300 python bench.py --from-images --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_image_in.log 2>&1 || exit 1
tail -1 gpurun_out/bench_image_in.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
PB="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-real-stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_run -- python3 $PB > $R/gpurun_out/bench_prof_run.log 2>&1 || exit 1
PP="$R/bench.py --steps 1 --warmup 0 --batch 16 --no-cpu-baseline --no-real-stats"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_run -- python3 $PP > $R/gpurun_out/pmc_fetch_run.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_run -- python3 $PP > $R/gpurun_out/pmc_write_run.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq_run -- python3 $PP > $R/gpurun_out/pmc_sq_run.log 2>&1 || exit 1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_summary_run.json gpurun_out/pmc_fetch_run gpurun_out/pmc_write_run gpurun_out/pmc_sq_run
find gpurun_out/prof_run -name "*kernel_stats*" | head -3
bash tools/run_latency_trace.sh > gpurun_out/latency_trace.txt 2>&1
timeout -k 10 120 python tools/akaze_time.py > gpurun_out/akaze_time.jsonl 2>/dev/null
cat gpurun_out/akaze_time.jsonl
