mkdir -p gpurun_out
for cfg in "8 4" "12 4" "12 6" "16 6" "16 8" "24 8"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 300 python bench.py --steps 96 --warmup 8 --in-flight $2 --no-cpu-baseline > gpurun_out/bench_hwq$1_if$2.log 2>&1 || { tail -5 gpurun_out/bench_hwq$1_if$2.log; exit 1; }
  tail -1 gpurun_out/bench_hwq$1_if$2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq $1 in-flight $2:', round(d['value'],1), d['latency_ms']['p50_at_throughput'])"
done
SFMLOC_BENCH_FORCE_SHARDED=1 GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python bench.py --steps 96 --warmup 16 --in-flight 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded path hwq16 ctx 2x4:', round(d['value'],1))"
SFMLOC_BENCH_FORCE_SHARDED=1 GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --steps 96 --warmup 16 --in-flight 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded path hwq8 ctx 2x4:', round(d['value'],1))"
