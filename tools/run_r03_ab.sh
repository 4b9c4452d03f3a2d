mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/ab.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in $EXTRA_ARGS > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s |', round(1e6/d['value'],1), 'us/query | alone p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3), '| stage', {k: round(v*1e3,3) for k,v in d['latency_ms']['stage_seconds_last_query'].items() if v}, '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
}
rm -f $OUT
run "screen<8,10,4> for short scans (default)" X=1
run "screen<8,10,1> always" SFMLOC_K1_SCREEN_BATCH=1
run "default again" X=1
run "screen<8,10,1> again" SFMLOC_K1_SCREEN_BATCH=1
