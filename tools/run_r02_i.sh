mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || { tail -80 gpurun_out/pytest_gpu.log; exit $rc; }
timeout -k 10 600 python tests/tools/fuzz_parity.py 120 61000 > gpurun_out/fuzz_parity.txt 2>&1; rc=$?; tail -1 gpurun_out/fuzz_parity.txt; [ $rc -eq 0 ] || exit $rc
rm -f gpurun_out/k1_qsplit_sweep.txt
for qs in 1 2 4 8 16 0; do
  SFMLOC_K1_QSPLIT=$qs timeout -k 10 300 python bench.py --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print('K1 qsplit $qs (0 = automatic): p50', round(d['latency_ms']['p50'],3), 'ms', d['config']['queries_localised'])" | tee -a gpurun_out/k1_qsplit_sweep.txt
done
bash tools/run_latency_trace.sh > gpurun_out/latency_trace.txt 2>&1; head -12 gpurun_out/latency_trace.txt
