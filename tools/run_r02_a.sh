# round-2 first GPU session: parity suite, the reconciled VALU ceiling, the K1 ladder, the new default bench, the
# sharded path on one rank (nccl) and on two ranks sharing the GPU (gloo)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/pytest_gpu.log; exit $rc; }
timeout -k 10 120 ./tools/valu_rates > gpurun_out/valu_rates.jsonl 2>&1 || { tail -5 gpurun_out/valu_rates.jsonl; exit 1; }
grep -E '"waves_per_simd": 8' gpurun_out/valu_rates.jsonl | cut -c1-200
timeout -k 10 120 ./tools/k1_ladder > gpurun_out/k1_ladder.jsonl 2>&1 || { tail -5 gpurun_out/k1_ladder.jsonl; exit 1; }
cut -c1-330 gpurun_out/k1_ladder.jsonl
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1 || { tail -30 gpurun_out/bench_default.log; exit 1; }
tail -1 gpurun_out/bench_default.log | cut -c1-1500
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_forced_sharded.log 2>&1 || { tail -30 gpurun_out/bench_forced_sharded.log; exit 1; }
tail -1 gpurun_out/bench_forced_sharded.log | cut -c1-900
SFMLOC_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_world2.log 2>&1 || { tail -30 gpurun_out/bench_world2.log; exit 1; }
tail -1 gpurun_out/bench_world2.log | cut -c1-900
