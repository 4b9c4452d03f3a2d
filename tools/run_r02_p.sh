mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase $EXTRA_ARGS > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$label:', round(d['value'],1), 'q/s  at-load p50', round(d['latency_ms']['p50_at_throughput'],2), d['config']['queries_localised'], '| PnP bracket', round(s['PnP(K5)'],2))" | tee -a gpurun_out/p3p_adaptive_policy.txt
}
rm -f gpurun_out/p3p_adaptive_policy.txt
for q in 4 6 8 12; do for f in 16 32 64; do
run "next batch = max($f, $q/4 t)" SFMLOC_P3P_ADAPT_QUARTERS=$q SFMLOC_P3P_ADAPT_FLOOR=$f
done; done
