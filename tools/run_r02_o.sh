mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase $EXTRA_ARGS > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$label:', round(d['value'],1), 'q/s  p50', round(d['latency_ms']['p50'],3), 'at-load p50', round(d['latency_ms']['p50_at_throughput'],2), d['config']['queries_localised'], '| PnP bracket', round(s['PnP(K5)'],2))" | tee -a gpurun_out/p3p_adaptive_sweep.txt
}
rm -f gpurun_out/p3p_adaptive_sweep.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run "adaptive 0, rounds 9" SFMLOC_P3P_ADAPTIVE=0
run "adaptive 1, rounds 9" SFMLOC_P3P_ADAPTIVE=1
run "adaptive 1, rounds 12" SFMLOC_P3P_ADAPTIVE=1 SFMLOC_P3P_ROUNDS=12
run "adaptive 1, rounds 15" SFMLOC_P3P_ADAPTIVE=1 SFMLOC_P3P_ROUNDS=15
run "adaptive 1, rounds 18" SFMLOC_P3P_ADAPTIVE=1 SFMLOC_P3P_ROUNDS=18
run "auto" X=1
