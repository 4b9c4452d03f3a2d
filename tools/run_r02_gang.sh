# gang sessions on the N>1 path: randomised parity (ganged shards and merges against the oracle), the forced-sharded
# world-1 rate with and without gangs, the world-2 / world-4 rehearsals over gloo on the one GPU
mkdir -p gpurun_out
SFMLOC_FUZZ_GANG=1 timeout -k 10 900 python tests/tools/fuzz_sharded.py ${FUZZ_N:-200} 21000 > gpurun_out/fuzz_sharded_gang.txt 2>&1 || { tail -20 gpurun_out/fuzz_sharded_gang.txt; exit 1; }
tail -1 gpurun_out/fuzz_sharded_gang.txt
for g in 1 16; do
  SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --gang $g --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_forced_sharded_gang$g.log 2>&1 || { tail -30 gpurun_out/bench_forced_sharded_gang$g.log; exit 1; }
  tail -1 gpurun_out/bench_forced_sharded_gang$g.log | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('forced sharded, world 1, queries per launch $g:', round(d['value'], 1), 'q/s', d['config']['queries_localised'])"
done
bash tools/run_world2_rehearsal.sh
