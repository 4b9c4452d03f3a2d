mkdir -p gpurun_out
timeout -k 10 120 ./tools/valu_rates > gpurun_out/valu_rates.jsonl 2>&1 || { tail -5 gpurun_out/valu_rates.jsonl; exit 1; }
grep -E '"waves_per_simd": 8' gpurun_out/valu_rates.jsonl | grep -E 'xor' | cut -c1-200
rm -f gpurun_out/inflight_sweep.txt
for hwq in 8 12 24; do for n in 3 4 5 6 8 12; do
  GPU_MAX_HW_QUEUES=$hwq timeout -k 10 300 python bench.py --in-flight $n --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('hwq $hwq inflight $n', round(d['value'],1), 'q/s p50', round(d['latency_ms']['p50'],3))" | tee -a gpurun_out/inflight_sweep.txt
done; done
