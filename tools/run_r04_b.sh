# round 4: the sequential form of K5 where queries come in gangs (one rank of N) and on the image-in leg
mkdir -p gpurun_out/r04_b
O=gpurun_out/r04_b
rm -f $O/rank.txt $O/image.txt
re() { label=$1; n=$2; shift 2
  env "$@" timeout -k 10 400 python tools/rank_emulation.py --of $n > $O/re.log 2>&1 || { tail -30 $O/re.log; exit 1; }
  python -c "
import json; d=json.loads(open('$O/re.log').read().strip().splitlines()[-1]); print('$label N=$n:', round(d['rank_rate_queries_per_s']), 'q/s | ms per batch', round(d['ms_per_batch'],2), d['host_ms_per_batch_in'], d.get('same_as_unsharded'))" | tee -a $O/rank.txt
}
re "rounds" 8 SFMLOC_P3P_SEQ=0
re "seq 8 waves" 8 SFMLOC_P3P_SEQ_WAVES=8
re "seq 16 waves" 8 SFMLOC_P3P_SEQ_WAVES=16
re "seq 16 waves" 2 SFMLOC_P3P_SEQ_WAVES=16
re "rounds" 2 SFMLOC_P3P_SEQ=0
im() { label=$1; shift
  env "$@" timeout -k 10 400 python bench.py --image-in-only --image-steps 4 > $O/im.log 2> $O/im.err || { tail -30 $O/im.err; exit 1; }
  python -c "
import json; d=json.loads(open('$O/im.log').read().strip().splitlines()[-1])['image_in']; print('$label:', round(d['value']), 'images/s | p50 alone', round(d['latency_ms']['p50'],2), '| under load', round(d['latency_ms']['p50_at_throughput'],1), '| identical', d['identical_to_single_flight'], d.get('path_stage_ms_one_frame_alone'))" | tee -a $O/image.txt
}
im "image-in, rounds" SFMLOC_P3P_SEQ=0
im "image-in, seq 16 waves" SFMLOC_P3P_SEQ_WAVES=16
im "image-in, seq 8 waves" SFMLOC_P3P_SEQ_WAVES=8
