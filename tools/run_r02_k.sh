# round-2 evidence, second part: issue-rate census, K1 ladder, N>1 rehearsal on one card, randomised parity campaigns
mkdir -p gpurun_out
timeout -k 10 180 ./tools/valu_rates > gpurun_out/valu_rates.jsonl 2>&1 || { tail -5 gpurun_out/valu_rates.jsonl; exit 1; }
timeout -k 10 180 ./tools/k1_ladder > gpurun_out/k1_ladder.jsonl 2>&1 || { tail -5 gpurun_out/k1_ladder.jsonl; exit 1; }
tail -3 gpurun_out/k1_ladder.jsonl | cut -c1-300
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_forced_sharded.log 2>&1 || { tail -30 gpurun_out/bench_forced_sharded.log; exit 1; }
tail -1 gpurun_out/bench_forced_sharded.log | cut -c1-400
bash tools/run_world2_rehearsal.sh || exit 1
N_PARITY=500 N_SHARDED=250 N_MAPSIDE=250 bash tools/run_fuzz.sh
