mkdir -p gpurun_out
rm -f gpurun_out/sharded_batch_sweep.txt
for b in 256 128 64 32; do
steps=$((2048 / b)); warm=$((512 / b))
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --batch $b --steps $steps --warmup $warm --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('forced sharded world 1, batch $b:', round(d['value'],1), 'q/s', d['config']['queries_localised'], json.dumps(d['exchange'])[:200])" | tee -a gpurun_out/sharded_batch_sweep.txt
done
