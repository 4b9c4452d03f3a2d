bash tools/run_akaze_profile.sh
timeout -k 10 300 python -m pytest tests/test_gpu_akaze.py -m gpu -x -q 2>&1 | tail -3
