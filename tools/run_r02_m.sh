mkdir -p gpurun_out
run() { # label, env..., args
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase $EXTRA_ARGS > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$label:', round(d['value'],1), 'q/s  p50', round(d['latency_ms']['p50'],3), 'at-load p50', round(d['latency_ms']['p50_at_throughput'],2), '| brackets', ' '.join(f'{k.split(chr(40))[0]}={v:.2f}' for k,v in s.items() if k!='note'))"
}
run "12ctx/4thr" X=1
EXTRA_ARGS="--in-flight 16 --threads 4" run "16ctx/4thr" GPU_MAX_HW_QUEUES=20
EXTRA_ARGS="--in-flight 8 --threads 4" run "8ctx/4thr" X=1
EXTRA_ARGS="--in-flight 4 --threads 1" run "4ctx/1thr" X=1
EXTRA_ARGS="--in-flight 4 --threads 4" run "4ctx/4thr" X=1
run "p3p iter 40" SFMLOC_DIAG_P3P_ITER=40
run "ransac 1" SFMLOC_DIAG_RANSAC_ROUND=1
