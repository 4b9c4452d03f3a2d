mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 40 --warmup 4 --in-flight 1 --no-cpu-baseline > gpurun_out/bench_lat.log 2>&1; tail -1 gpurun_out/bench_lat.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-flight 1:', d['value'], d['latency_ms'], d['stage_ms'], d['roofline']['kernel_ms'])"
for n in 2 4 8; do timeout -k 10 300 python bench.py --steps 60 --warmup 8 --in-flight $n --no-cpu-baseline > gpurun_out/bench_if$n.log 2>&1; tail -1 gpurun_out/bench_if$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-flight $n:', d['value'], d['latency_ms'], d['stage_ms'], d['roofline']['kernel_ms'])"; done
