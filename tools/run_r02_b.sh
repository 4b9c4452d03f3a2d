mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/pytest_gpu.log; exit $rc; }
timeout -k 10 120 ./tools/valu_rates > gpurun_out/valu_rates.jsonl 2>&1 || { tail -5 gpurun_out/valu_rates.jsonl; exit 1; }
grep -E '"waves_per_simd": (4|8)' gpurun_out/valu_rates.jsonl | grep -E 'xor' | cut -c1-420
for n in 4 8 16; do
  timeout -k 10 300 python bench.py --in-flight $n --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_inflight$n.log 2>&1 || { tail -30 gpurun_out/bench_inflight$n.log; exit 1; }
  tail -1 gpurun_out/bench_inflight$n.log | cut -c1-120
done
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_forced_sharded.log 2>&1 || { tail -30 gpurun_out/bench_forced_sharded.log; exit 1; }
tail -1 gpurun_out/bench_forced_sharded.log | cut -c1-120
for n in 2 4; do
SFMLOC_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus $n --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_world$n.log 2>&1 || { tail -30 gpurun_out/bench_world$n.log; exit 1; }
tail -1 gpurun_out/bench_world$n.log | cut -c1-120
done
