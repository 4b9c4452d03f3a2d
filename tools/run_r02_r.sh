mkdir -p gpurun_out
rm -f gpurun_out/image_in_sweep.txt
for share in 1 0; do for n in 4 8 12; do
  SFMLOC_BENCH_SHARE_STREAM=$share timeout -k 10 300 python bench.py --from-images --in-flight $n --steps 6 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('image in, $n workers, extractor on the context stream = $share:', round(d['value'],1), 'img/s', d['config']['queries_localised'], 'at-load p50', round(d['latency_ms']['p50_at_throughput'],2))" | tee -a gpurun_out/image_in_sweep.txt
done; done
