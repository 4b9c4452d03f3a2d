mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python tests/tools/fuzz_parity.py 150 99000 | tail -1
timeout -k 10 300 python tests/tools/fuzz_sharded.py 100 98000 | tail -1
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase $EXTRA_ARGS > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$label:', round(d['value'],1), 'q/s  at-load p50', round(d['latency_ms']['p50_at_throughput'],2), '| alone p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3), d['config']['queries_localised'])"
}
run "whole chain" X=1
run "whole chain again" X=1
