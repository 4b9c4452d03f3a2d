mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/ab.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in $EXTRA_ARGS > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s | alone p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3), '| geoMatch', round(d['latency_ms']['stage_seconds_last_query']['geoMatch']*1e3,3), '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
}
rm -f $OUT
run "first batch = one pass of the waves" X=1
run "again" X=1
