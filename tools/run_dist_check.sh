mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_geom.py -m gpu -x -q -k "sharded or concurrent" > gpurun_out/pytest_dist.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_dist.log; tail -4 gpurun_out/pytest_dist.log
[ $rc -eq 0 ] || exit $rc
# world-1 rehearsal of the N>1 launch line (one rank, RCCL backend, sharded code path forced)
SFMLOC_BENCH_FORCE_SHARDED=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1; rc=$?
tail -1 gpurun_out/bench_torchrun1.log | cut -c1-900
exit $rc
