mkdir -p gpurun_out/r03_ab
OUT=gpurun_out/r03_ab/image_sweep.txt
run() { label=$1; shift
  timeout -k 10 300 python bench.py --image-in-only --no-cpu-baseline "$@" > gpurun_out/r03_ab/b.log 2> gpurun_out/r03_ab/b.err || { tail -30 gpurun_out/r03_ab/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_ab/b.log').read().strip().splitlines()[-1])['image_in']; print('$label:', round(d['value'],1), 'images/s | alone p50', round(d['latency_ms']['p50'],2), '| at throughput', round(d['latency_ms']['p50_at_throughput'],1), '| identical', d['identical_to_single_flight'], '| oracle', d['oracle_end_to_end']['frames_identical'])" | tee -a $OUT
}
rm -f $OUT
run "4 workers x 4 frames, a stream per BoW chain (default)"
run "4 workers, BoW chains on one stream per worker" --image-bow-worker-stream
run "6 workers, one BoW stream per worker" --image-bow-worker-stream --image-workers 6
run "8 workers, one BoW stream per worker" --image-bow-worker-stream --image-workers 8
run "10 workers, one BoW stream per worker" --image-bow-worker-stream --image-workers 10
run "8 workers x 2 frames, one BoW stream per worker" --image-bow-worker-stream --image-workers 8 --image-batch 2
run "8 workers x 8 frames, one BoW stream per worker" --image-bow-worker-stream --image-workers 8 --image-batch 8
run "8 workers, BoW on the worker's stream" --image-bow-shared-stream --image-workers 8
