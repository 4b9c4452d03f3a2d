mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || { tail -80 gpurun_out/pytest_gpu.log; exit $rc; }
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1 || { tail -30 gpurun_out/bench_default.log; exit 1; }
tail -1 gpurun_out/bench_default.log | cut -c1-300
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_forced_sharded.log 2>&1 || { tail -30 gpurun_out/bench_forced_sharded.log; exit 1; }
tail -1 gpurun_out/bench_forced_sharded.log | cut -c1-120
