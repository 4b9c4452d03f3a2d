# randomised parity campaigns on the final code of the round (copy the .txt files into profiles/)
mkdir -p gpurun_out
timeout -k 10 1000 python tests/tools/fuzz_parity.py ${N_PARITY:-600} 52000 > gpurun_out/fuzz_parity.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_parity.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_sharded.py ${N_SHARDED:-300} 58000 > gpurun_out/fuzz_sharded.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_sharded.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_mapside.py ${N_MAPSIDE:-300} > gpurun_out/fuzz_mapside.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_mapside.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_akaze.py ${N_AKAZE:-1000} 64000 > gpurun_out/fuzz_akaze.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_akaze.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_bow.py ${N_BOW:-1000} > gpurun_out/fuzz_bow.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_bow.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_undistort.py ${N_UNDIST:-1000} > gpurun_out/fuzz_undistort.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_undistort.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_gang.py ${N_GANG:-1000} 31000 > gpurun_out/fuzz_gang.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_gang.txt; [ $rc -eq 0 ] || exit $rc
SFMLOC_FUZZ_BATCH=1 timeout -k 10 600 python tests/tools/fuzz_akaze.py ${N_AKAZE_BATCH:-300} 71000 > gpurun_out/fuzz_akaze_batch.txt 2>&1; rc=$?; tail -2 gpurun_out/fuzz_akaze_batch.txt; [ $rc -eq 0 ] || exit $rc
