mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_geom.py -x -q -k "shared_gpu" 2>&1 | tail -3
SFMLOC_P3P_ADAPTIVE=1 timeout -k 10 900 python tests/tools/fuzz_parity.py 400 81000 > gpurun_out/fuzz_parity_adaptive.txt 2>&1; tail -2 gpurun_out/fuzz_parity_adaptive.txt
SFMLOC_P3P_ADAPTIVE=1 timeout -k 10 600 python tests/tools/fuzz_sharded.py 150 83000 > gpurun_out/fuzz_sharded_adaptive.txt 2>&1; tail -2 gpurun_out/fuzz_sharded_adaptive.txt
