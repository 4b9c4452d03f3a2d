# Evidence run: GPU tests, default bench, single-query bench, rocprofv3 kernel-trace stats, then the PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ_*) each in its own run, summarised by tools/pmc_summary.py.  Copy the results into
# profiles/ afterwards (gpurun_out/ is scratch).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1 || exit 1
tail -1 gpurun_out/bench_default.log
timeout -k 10 300 python bench.py --in-flight 1 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/bench_latency.log 2>&1 || exit 1
tail -1 gpurun_out/bench_latency.log | cut -c1-200
# BASELINE configs[2]: 10 k views, BoW shortlist k = 100, with its own CPU baseline
timeout -k 10 400 python bench.py --views 10000 --bow-knn 100 --steps 384 --warmup 16 > gpurun_out/bench_cfg3.log 2>&1 || exit 1
tail -1 gpurun_out/bench_cfg3.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_run -- python3 $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline > $R/gpurun_out/bench_prof_run.log 2>&1 || exit 1
# the same with one query in flight: a launch's duration is then the kernel's own (in the default run four queries share
# the chip and every launch is stretched by the others' work), and rocprof's average agrees with bench.py's events
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lat_run -- python3 $R/bench.py --in-flight 1 --steps 40 --warmup 4 --no-cpu-baseline > $R/gpurun_out/bench_prof_lat_run.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_run -- python3 $R/bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_run.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_run -- python3 $R/bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline > $R/gpurun_out/pmc_write_run.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq_run -- python3 $R/bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq_run.log 2>&1 || exit 1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_summary_run.json gpurun_out/pmc_fetch_run gpurun_out/pmc_write_run gpurun_out/pmc_sq_run
find gpurun_out/prof_run -name "*kernel_stats*" | head
