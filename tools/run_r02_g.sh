mkdir -p gpurun_out
rm -f gpurun_out/sharded_sweep.txt
for n in 4 6 8 10 11; do
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --in-flight $n --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); s=d.get('stage_ms') or {}; print('forced sharded world 1, contexts per slot $n:', round(d['value'],1), 'q/s', d['config']['queries_localised'], {k.split('(')[0]:round(v,2) for k,v in s.items() if k!='note'})" | tee -a gpurun_out/sharded_sweep.txt
done
