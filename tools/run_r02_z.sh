mkdir -p gpurun_out
rm -f gpurun_out/inflight_sweep_final.txt
for cfg in "8 4 12" "12 4 14" "16 4 18" "16 8 18" "20 4 22" "22 4 24"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout -k 10 300 python bench.py --in-flight $1 --threads $2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('contexts $1, host threads $2, hardware queues $3:', round(d['value'],1), 'q/s  at-load p50', round(d['latency_ms']['p50_at_throughput'],2), d['config']['queries_localised'])" | tee -a gpurun_out/inflight_sweep_final.txt
done
