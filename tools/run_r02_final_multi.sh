# multi-GPU evidence on the final code: gang fuzz, adaptive-round fuzz, forced-sharded world 1 (gangs of 1 / 16), world
# 2 / 4 rehearsals, one rank of 2 / 4 / 8 with one and with sixteen queries per launch
mkdir -p gpurun_out
FUZZ_N=400 bash tools/run_r02_gang.sh || exit 1
bash tools/run_r02_u.sh || exit 1
WORLDS="2 4 8" TAG=gang1 EXTRA="--gang 1 --in-flight 8" bash tools/run_rank_emulation.sh > /dev/null || exit 1
WORLDS="2 4 8" TAG=gang16 EXTRA="--gang 16 --in-flight 32" bash tools/run_rank_emulation.sh > /dev/null || exit 1
cat gpurun_out/rank_emulation_gang1.jsonl gpurun_out/rank_emulation_gang16.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['emulated'], 'queries per launch', d['queries_per_launch_stage1'], round(d['rank_rate_queries_per_s']), 'q/s', d['same_as_unsharded'], d['localised_of_own'])"
