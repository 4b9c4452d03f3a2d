# final numbers of the round on the current code: default bench, configs[1], image-in, latency trace, stamps
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1 || { tail -20 gpurun_out/bench_default.log; exit 1; }
tail -1 gpurun_out/bench_default.log | cut -c1-160
timeout -k 10 300 python bench.py --views 1000 --bow-knn 0 --no-real-stats > gpurun_out/bench_cfg1.log 2>&1 || exit 1
tail -1 gpurun_out/bench_cfg1.log | cut -c1-160
timeout -k 10 300 python bench.py --from-images --no-cpu-baseline --no-roofline-phase > gpurun_out/bench_image_in.log 2>&1 || exit 1
tail -1 gpurun_out/bench_image_in.log | cut -c1-160
bash tools/run_latency_trace.sh > gpurun_out/latency_trace.txt 2>&1
bash tools/run_stamps.sh > /dev/null 2>&1
timeout -k 10 120 python tools/akaze_time.py > gpurun_out/akaze_time.jsonl 2>/dev/null; cat gpurun_out/akaze_time.jsonl
