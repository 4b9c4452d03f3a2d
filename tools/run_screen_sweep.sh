mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_hamming.py tests/test_gpu_end_to_end.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for cfg in "10 64" "10 128" "10 256" "10 512" "11 128" "11 32"; do
  set -- $cfg
  SFMLOC_K1_SCREEN_NW=$1 SFMLOC_K1_SCREEN_HEAD=$2 timeout -k 10 300 python bench.py --steps 40 --warmup 4 --in-flight 1 --no-cpu-baseline > gpurun_out/bench_nw$1_h$2.log 2>&1 || exit 1
  tail -1 gpurun_out/bench_nw$1_h$2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nw $1 head $2:', round(d['value'],1), d['latency_ms']['p50'], d['roofline']['kernel_ms'], d['roofline']['valu'])"
done
