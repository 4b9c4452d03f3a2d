mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/ab.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in $EXTRA_ARGS > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s |', round(1e6/d['value'],1), 'us/query | alone p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3), '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
}
rm -f $OUT
run "K5 small round capped at 128 VGPRs" X=1
run "again" X=1
EXTRA_ARGS="--in-flight 22" run "22 in flight" X=1
