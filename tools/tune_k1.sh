# sweeps the K1 geometry on the bench workload; prints kernel ms per variant
mkdir -p gpurun_out
: > gpurun_out/tune_k1.txt
for g in "2,8,2048" "4,16,2048" "2,16,1024" "2,16,2048" "1,16,1024" "1,16,512" "2,8,1024" "1,8,1024" "1,8,512" "4,8,1024" "2,4,512" "1,4,512" "4,4,512" "2,16,512"; do
  SFMLOC_K1_GEOM=$g timeout -k 10 200 python bench.py --steps 16 --warmup 3 --in-flight 1 --no-cpu-baseline > gpurun_out/tune_tmp.log 2>&1
  tail -1 gpurun_out/tune_tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$g', 'k1_ms', round(d['roofline']['kernel_ms'],3), 'qps', round(d['value'],1))" >> gpurun_out/tune_k1.txt 2>&1 || echo "$g failed" >> gpurun_out/tune_k1.txt
done
cat gpurun_out/tune_k1.txt
