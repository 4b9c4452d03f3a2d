mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_akaze.py tests/test_gpu_extfeat.py -x -q > gpurun_out/pytest_akaze.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_akaze.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/pytest_akaze.log; exit $rc; }
timeout -k 10 400 python tests/tools/fuzz_akaze.py ${N_AKAZE:-150} 91000 > gpurun_out/fuzz_akaze.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_akaze.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/akaze_time.py 2>/dev/null
SFMLOC_AKAZE_RESIDENT=0 timeout -k 10 120 python tools/akaze_time.py 2>/dev/null
