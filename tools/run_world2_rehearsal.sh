# two and four ranks sharing the one GPU of the box, gloo for the exchange: exercises view sharding, candidate
# parts, query ownership and the two-slot pipeline of the N>1 path with the real kernels (not a performance number)
mkdir -p gpurun_out
for n in 2 4; do
  SFMLOC_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_world$n.log 2>&1; rc=$?
  tail -1 gpurun_out/bench_world$n.log | cut -c1-700
  [ $rc -eq 0 ] || { tail -20 gpurun_out/bench_world$n.log; exit $rc; }
done
