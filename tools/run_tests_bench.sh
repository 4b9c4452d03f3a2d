mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for n in 1 4; do
  timeout -k 10 300 python bench.py --steps 80 --warmup 8 --in-flight $n --no-cpu-baseline > gpurun_out/bench_if$n.log 2>&1 || { tail -5 gpurun_out/bench_if$n.log; exit 1; }
  tail -1 gpurun_out/bench_if$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-flight $n:', round(d['value'],1), d['latency_ms'], round(d['roofline']['kernel_ms'],3), {k:round(v,2) for k,v in d['stage_ms'].items()})"
done
