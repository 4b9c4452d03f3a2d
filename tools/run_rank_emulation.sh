# per-rank batch rate of the sharded path for N = 2, 4, 8, measured on one GPU (tools/rank_emulation.py)
mkdir -p gpurun_out
out=gpurun_out/rank_emulation${TAG:+_$TAG}.jsonl
rm -f $out
# (EXTRA: the bench's sharded defaults -- 32 queries per launch, 96 contexts per slot)
EXTRA=${EXTRA:---gang 32 --in-flight 96}
for n in ${WORLDS:-2 4 8}; do
  timeout -k 10 400 python tools/rank_emulation.py --of $n $EXTRA > gpurun_out/re.log 2>&1 || { tail -30 gpurun_out/re.log; exit 1; }
  tail -1 gpurun_out/re.log | tee -a $out
done
