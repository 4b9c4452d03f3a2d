mkdir -p gpurun_out
rm -f gpurun_out/p3p_batch_sweep.txt
for cfg in "512 9" "256 9" "128 12" "64 16" "32 24"; do set -- $cfg
for n in 1 4; do
  SFMLOC_P3P_BATCH=$1 SFMLOC_P3P_ROUNDS=$2 timeout -k 10 300 python bench.py --in-flight $n --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('p3p batch $1 rounds $2 inflight $n:', round(d['value'],1), 'q/s  p50', round(d['latency_ms']['p50'],3), d['config']['queries_localised'], 'PnP ms', round(d['stage_ms']['PnP(K5)'],3))" | tee -a gpurun_out/p3p_batch_sweep.txt
done; done
BENCH_ARGS="" bash tools/run_latency_trace.sh > gpurun_out/latency_trace.txt 2>&1; tail -60 gpurun_out/latency_trace.txt
