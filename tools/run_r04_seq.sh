# round 4: the sequential form of K5 (k_p3p_seq) -- parity with the form forced, then what it does to the headline
mkdir -p gpurun_out/r04_seq
O=gpurun_out/r04_seq
for w in ${SEQ_WAVES:-8 16 4}; do
  SFMLOC_P3P_SEQ=2 SFMLOC_P3P_SEQ_WAVES=$w timeout -k 10 300 python tests/tools/fuzz_parity.py 40 41000 > $O/fuzz_parity_w$w.txt 2>&1 || { tail -20 $O/fuzz_parity_w$w.txt; exit 1; }
  tail -1 $O/fuzz_parity_w$w.txt
  SFMLOC_P3P_SEQ=2 SFMLOC_P3P_SEQ_WAVES=$w timeout -k 10 300 python tests/tools/fuzz_p3p_large.py 10 42000 > $O/fuzz_large_w$w.txt 2>&1 || { tail -20 $O/fuzz_large_w$w.txt; exit 1; }
  tail -1 $O/fuzz_large_w$w.txt
done
run() { label=$1; shift; extra=""
  case "$label" in *"24 in flight"*) extra="--in-flight 24";; *"28 in flight"*) extra="--in-flight 28";; esac
  env "$@" timeout -k 10 300 python bench.py $extra --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase --no-image-in > $O/b.log 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('$O/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s |', round(1e6/d['value'],1), 'us of the chip per query | alone p50', round(d['latency_ms']['p50'],3), '| at throughput p50', round(d['latency_ms']['p50_at_throughput'],2), '| identical', d.get('identical_to_single_flight'), '| PnP alone', round(d['latency_ms']['stage_seconds_last_query']['PnP']*1e3,3))" | tee -a $O/rates.txt
}
rm -f $O/rates.txt
run "rounds (SFMLOC_P3P_SEQ=0)" SFMLOC_P3P_SEQ=0
run "seq when shared, 8 waves" SFMLOC_P3P_SEQ_WAVES=8
run "seq when shared, 16 waves" SFMLOC_P3P_SEQ_WAVES=16
run "seq always (alone too), 8 waves" SFMLOC_P3P_SEQ=2 SFMLOC_P3P_SEQ_WAVES=8
run "seq always (alone too), 16 waves" SFMLOC_P3P_SEQ=2 SFMLOC_P3P_SEQ_WAVES=16
run "seq always, 16 waves, no filter" SFMLOC_P3P_SEQ=2 SFMLOC_P3P_SEQ_WAVES=16 SFMLOC_P3P_FILTER=0
