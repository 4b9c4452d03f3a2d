mkdir -p gpurun_out
rm -f gpurun_out/stage_cost_diag.txt
run() { env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json,sys; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print(' '.join(sys.argv[1:]) or 'as is', ':', round(d['value'],1), 'q/s  p50', round(d['latency_ms']['p50'],3), d['config']['queries_localised'])" "$@" | tee -a gpurun_out/stage_cost_diag.txt; }
run X=1
run SFMLOC_DIAG_P3P_ITER=40
run SFMLOC_DIAG_RANSAC_ROUND=1
run SFMLOC_DIAG_P3P_ITER=40 SFMLOC_DIAG_RANSAC_ROUND=1
