mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/ab.txt
rm -f $OUT
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-image-in > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1]); r=d['roofline']; print('run $i:', round(d['value'],1), 'q/s | alone p50', round(d['latency_ms']['p50'],3), '| full scan', round(r['kernel_ms'],3), 'ms, valu frac', round(r['valu']['frac'],4), 'finished frac', r['valu']['pairs_finished_frac'], 'flagged', r['valu']['rows_flagged_per_scan'], '| hbm regime', round(r['hbm_bound_regime']['achieved']), '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
done
