mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || { tail -80 gpurun_out/pytest_gpu.log; exit $rc; }
