mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/image_sweep5.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --image-in-only --no-cpu-baseline > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1])['image_in']; print('$label:', round(d['value'],1), 'images/s | alone p50', round(d['latency_ms']['p50'],2), '| PnP alone', round(d['path_stage_ms_one_frame_alone']['PnP'],3), '| identical', d['identical_to_single_flight'], '| oracle', d['oracle_end_to_end']['frames_identical'])" | tee -a $OUT
}
rm -f $OUT
run "costly floor 16 (default)" X=1
run "costly floor 32" SFMLOC_P3P_COSTLY_FLOOR=32
run "costly floor 64" SFMLOC_P3P_COSTLY_FLOOR=64
run "costly floor 8" SFMLOC_P3P_COSTLY_FLOOR=8
