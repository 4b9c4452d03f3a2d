# round 4: one workgroup per MODEL for a query alone (p3p_eval_coop4) -- parity, then the latency of a query alone
mkdir -p gpurun_out/r04_c
O=gpurun_out/r04_c
timeout -k 10 300 python tests/tools/fuzz_parity.py 60 43000 > $O/fuzz_parity.txt 2>&1 || { tail -20 $O/fuzz_parity.txt; exit 1; }
tail -1 $O/fuzz_parity.txt
timeout -k 10 300 python tests/tools/fuzz_p3p_large.py 16 44000 > $O/fuzz_large.txt 2>&1 || { tail -20 $O/fuzz_large.txt; exit 1; }
tail -1 $O/fuzz_large.txt
SFMLOC_P3P_WIDE_ALONE=2 timeout -k 10 300 python tests/tools/fuzz_gang.py 40 45000 > $O/fuzz_gang.txt 2>&1 || { tail -20 $O/fuzz_gang.txt; exit 1; }
tail -1 $O/fuzz_gang.txt
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline-phase --no-image-in > $O/b.log 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
  python -c "
import json; d=json.loads(open('$O/b.log').read().strip().splitlines()[-1]); print('$label:', round(d['value'],1), 'q/s | alone p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3), '| identical', d.get('identical_to_single_flight'), '| PnP alone', round(d['latency_ms']['stage_seconds_last_query']['PnP']*1e3,3))" | tee -a $O/rates.txt
}
rm -f $O/rates.txt
run "default" X=1
run "one workgroup per model when alone (SFMLOC_P3P_WIDE_ALONE=1)" SFMLOC_P3P_WIDE_ALONE=1
im() { label=$1; shift
  env "$@" timeout -k 10 400 python bench.py --image-in-only --image-steps 3 > $O/im.log 2> $O/im.err || { tail -30 $O/im.err; exit 1; }
  python -c "
import json; d=json.loads(open('$O/im.log').read().strip().splitlines()[-1])['image_in']; print('$label:', round(d['value']), 'images/s | p50 alone', round(d['latency_ms']['p50'],2), '| under load', round(d['latency_ms']['p50_at_throughput'],1), '| identical', d['identical_to_single_flight'], d.get('path_stage_ms_one_frame_alone'))" | tee -a $O/image.txt
}
rm -f $O/image.txt
im "image-in, default" X=1
im "image-in, SFMLOC_P3P_COOP=0" SFMLOC_P3P_COOP=0
im "image-in, 128 groups" SFMLOC_P3P_WIDE_GROUPS=128
im "image-in, 32 groups" SFMLOC_P3P_WIDE_GROUPS=32
SFMLOC_LIB_PATH=$PWD/sfmlocalization_amd/lib/libsfmloc_hip_stamps.so timeout -k 10 250 python tools/image_lab.py 64 8 > $O/lab8.txt 2>&1
