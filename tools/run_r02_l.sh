mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_geom.py -x -q -k "wave_sort or pipeline or sampling or minimal" > gpurun_out/pytest_sort.log 2>&1; rc=$?; tail -5 gpurun_out/pytest_sort.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/pytest_sort.log; exit $rc; }
bash tools/run_tests.sh || exit 1
bash tools/run_stamps.sh | grep -E "query|K5 round  [0-2]|wg|end times" | head -40
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline-phase > gpurun_out/b.log 2>&1 || { tail -30 gpurun_out/b.log; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1]); print(round(d['value'],1), 'q/s  p50', round(d['latency_ms']['p50'],3), 'p95', round(d['latency_ms']['p95'],3), d['config']['queries_localised'])"
