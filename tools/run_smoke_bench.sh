mkdir -p gpurun_out
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout -k 10 400 python bench.py --steps 30 --warmup 3 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
