mkdir -p gpurun_out
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_forced_sharded.log 2>&1 || { tail -30 gpurun_out/bench_forced_sharded.log; exit 1; }
tail -1 gpurun_out/bench_forced_sharded.log | cut -c1-120
