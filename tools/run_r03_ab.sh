mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/ab.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in $EXTRA_ARGS > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s |', round(1e6/d['value'],1), 'us/query | alone p50', round(d['latency_ms']['p50'],3), '| at throughput', round(d['latency_ms']['p50_at_throughput'],2), '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
}
rm -f $OUT
run "default" X=1
run "no prepare-ahead" SFMLOC_P3P_PREP_AHEAD=0
run "floor 32" SFMLOC_P3P_ADAPT_FLOOR=32
run "floor 128" SFMLOC_P3P_ADAPT_FLOOR=128
run "quarters 8" SFMLOC_P3P_ADAPT_QUARTERS=8
run "rounds queued 7" SFMLOC_P3P_ROUNDS=7
run "rounds queued 12" SFMLOC_P3P_ROUNDS=12
run "batch 128" SFMLOC_P3P_BATCH=128
run "K3 8 waves shared" SFMLOC_K3_WAVES_SHARED=8
run "default again" X=1
