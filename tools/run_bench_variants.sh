mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log
timeout -k 10 300 python bench.py --steps 40 --warmup 4 --in-flight 1 --no-cpu-baseline > gpurun_out/bench_lat.log 2>&1; tail -1 gpurun_out/bench_lat.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-flight 1:', d['value'], d['latency_ms'], d['stage_ms'], d['roofline']['kernel_ms'])"
timeout -k 10 300 python bench.py --steps 60 --warmup 6 --in-flight 3 --no-cpu-baseline > gpurun_out/bench_if3.log 2>&1; tail -1 gpurun_out/bench_if3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-flight 3:', d['value'], d['latency_ms'], d['stage_ms'], d['roofline']['kernel_ms'])"
# world-size-1 run of the sharded code path under torchrun (RCCL init, all_gather plumbing) on the one GPU
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 32 --warmup 16 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1; tail -1 gpurun_out/bench_torchrun1.log | cut -c1-400
