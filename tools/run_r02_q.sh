mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase $EXTRA_ARGS > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$label:', round(d['value'],1), 'q/s  at-load p50', round(d['latency_ms']['p50_at_throughput'],2), '| alone p50', round(d['latency_ms']['p50'],3))" | tee -a gpurun_out/chain_prefix_rates.txt
}
rm -f gpurun_out/chain_prefix_rates.txt
run "chain up to putative matches (K8, K1, K2)" SFMLOC_DIAG_STOP_AFTER=1
run "... + geometric filter (K3)" SFMLOC_DIAG_STOP_AFTER=2
run "... + 2D-3D set (K4)" SFMLOC_DIAG_STOP_AFTER=3
run "whole chain" X=1
EXTRA_ARGS="--in-flight 4 --threads 1" run "K1 only, 4 in flight" SFMLOC_DIAG_STOP_AFTER=1
EXTRA_ARGS="--in-flight 8 --threads 4" run "K1 only, 8 in flight" SFMLOC_DIAG_STOP_AFTER=1
