mkdir -p gpurun_out
for r in 9 10 11 12; do
  SFMLOC_P3P_ROUNDS=$r timeout -k 10 300 python bench.py --in-flight 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('rounds queued $r:', round(d['value'],1), 'q/s one in flight; p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3))"
done
