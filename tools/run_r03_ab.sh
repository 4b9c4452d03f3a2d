mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/ab.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in $EXTRA_ARGS > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s |', round(1e6/d['value'],1), 'us/query | alone p50', round(d['latency_ms']['p50'],3), '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
}
rm -f $OUT
run "lean scan, 4 workgroups per CU (default)" X=1
run "lean scan padded to 3 workgroups per CU" SFMLOC_K1_LDS_PAD=12000
run "default again" X=1
run "padded again" SFMLOC_K1_LDS_PAD=12000
