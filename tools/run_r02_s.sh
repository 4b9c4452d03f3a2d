mkdir -p gpurun_out
rm -f gpurun_out/k1_qsplit_fine.txt
for qs in 3 4 5 6 7 10; do
  SFMLOC_K1_QSPLIT=$qs timeout -k 10 300 python bench.py --in-flight 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('slices $qs:', 'p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3), d['config']['queries_localised'])" | tee -a gpurun_out/k1_qsplit_fine.txt
done
