mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/image_sweep2.txt
run() { label=$1; shift
  timeout -k 10 300 python bench.py --image-in-only --no-cpu-baseline "$@" > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1])['image_in']; print('$label:', round(d['value'],1), 'images/s | alone p50', round(d['latency_ms']['p50'],2), '| at throughput', round(d['latency_ms']['p50_at_throughput'],1), '| identical', d['identical_to_single_flight'], '| oracle', d['oracle_end_to_end']['frames_identical'])" | tee -a $OUT
}
rm -f $OUT
run "12 workers x 4" --image-workers 12
run "16 workers x 4, BoW on the extraction stream" --image-bow-shared-stream --image-workers 16
run "20 workers x 2, BoW on the extraction stream" --image-bow-shared-stream --image-workers 20 --image-batch 2
run "20 workers x 4, BoW on the extraction stream" --image-bow-shared-stream --image-workers 20
run "14 workers x 4" --image-workers 14
