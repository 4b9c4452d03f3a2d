# what each latency-bound stage costs the throughput: the headline with the chain cut after K1 / K3 / K4 (SFMLOC_DIAG_STOP_AFTER)
R=${ROUND:-r04}; O=gpurun_out/${R}_final; mkdir -p $O
OUT=$O/chain_prefix_rates.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in > $O/b.log 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('$O/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s |', round(1e6/d['value'],1), 'us of the chip per query | alone p50', round(d['latency_ms']['p50'],3), '| at throughput p50', round(d['latency_ms']['p50_at_throughput'],2), '| identical', d.get('identical_to_single_flight'))" | tee -a $OUT
}
rm -f $OUT
run "chain up to putative matches (K8, K1, K2)" SFMLOC_DIAG_STOP_AFTER=1
run "... + geometric filter (K3)" SFMLOC_DIAG_STOP_AFTER=2
run "... + 2D-3D set (K4)" SFMLOC_DIAG_STOP_AFTER=3
run "whole chain" X=1
