# rocprofv3 evidence for the bench configuration: kernel trace stats, then PMC passes (separate runs)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01b -- python3 $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline > $R/gpurun_out/bench_prof_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
find gpurun_out/prof_r01b gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq -name "*.csv" | head -20
tail -2 gpurun_out/pmc_sq.log | cut -c1-300
