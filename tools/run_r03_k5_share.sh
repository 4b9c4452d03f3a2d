# what K5 costs the throughput, and what it scales with (round 3): bench.py's headline leg under diagnostic settings
mkdir -p gpurun_out/r03_k5
OUT=gpurun_out/r03_k5/k5_share.txt
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in $EXTRA_ARGS > gpurun_out/r03_k5/b.log 2> gpurun_out/r03_k5/b.err || { tail -30 gpurun_out/r03_k5/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r03_k5/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s |', round(1e6/d['value'],1), 'us of the chip per query | alone p50', round(d['latency_ms']['p50'],3), '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
}
rm -f $OUT
run "whole chain" X=1
run "chain up to putative matches (K8, K1, K2)" SFMLOC_DIAG_STOP_AFTER=1
run "... + geometric filter (K3)" SFMLOC_DIAG_STOP_AFTER=2
run "... + 2D-3D set (K4)" SFMLOC_DIAG_STOP_AFTER=3
run "whole chain, P3P budget 64 iterations (one round)" SFMLOC_DIAG_P3P_ITER=64
run "whole chain, P3P budget 256" SFMLOC_DIAG_P3P_ITER=256
run "whole chain, P3P budget 1024" SFMLOC_DIAG_P3P_ITER=1024
run "whole chain, rounds of 64" SFMLOC_P3P_BATCH=64
run "whole chain, rounds of 128" SFMLOC_P3P_BATCH=128
run "whole chain, shared-GPU floor 16" SFMLOC_P3P_ADAPT_FLOOR=16
run "whole chain, shared-GPU floor 32, quarters 8" SFMLOC_P3P_ADAPT_FLOOR=32 SFMLOC_P3P_ADAPT_QUARTERS=8
run "whole chain, F-matrix budget 5 rounds" SFMLOC_DIAG_RANSAC_ROUND=5
