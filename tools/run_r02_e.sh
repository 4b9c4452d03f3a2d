mkdir -p gpurun_out
rm -f gpurun_out/threads_sweep.txt
for cfg in "4 0" "4 2" "4 4" "8 4" "8 8" "12 4" "12 6"; do set -- $cfg
  SFMLOC_P3P_BATCH=256 GPU_MAX_HW_QUEUES=24 timeout -k 10 300 python bench.py --in-flight $1 --threads $2 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('inflight $1 threads $2:', round(d['value'],1), 'q/s  p50', round(d['latency_ms']['p50'],3), d['config']['queries_localised'])" | tee -a gpurun_out/threads_sweep.txt
done
bash tools/run_latency_trace.sh > gpurun_out/latency_trace.txt 2>&1; tail -50 gpurun_out/latency_trace.txt
