# Round-3 evidence run (one gpurun call): the GPU suite, the default bench (configs[2] headline + image_in), rocprofv3
# kernel-trace stats of the same command (incl. the image-in leg: first timing of the A5a-c kernels), PMC passes of the
# roofline kernel (separate runs), AKAZE times, K5 on extracted descriptors.  Copy what is to be judged into profiles/.
mkdir -p gpurun_out/r03_final
O=gpurun_out/r03_final
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("headline", round(d["value"]), d["identical_to_single_flight"], "p50", round(d["latency_ms"]["p50"],3), "valu frac", round(d["roofline"]["valu"]["frac"],3), "| image_in", round(d["image_in"]["value"]), d["image_in"]["identical_to_single_flight"], "p50", round(d["image_in"]["latency_ms"]["p50"],2))
PY
cd /tmp && export TMPDIR=/tmp
PB="$R/bench.py --steps 2 --warmup 1 --image-steps 2 --no-cpu-baseline --no-real-stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o bench -- python3 $PB > $R/$O/bench_prof_run.log 2>&1 || exit 1
PP="$R/bench.py --steps 1 --warmup 0 --batch 16 --no-cpu-baseline --no-real-stats --no-image-in"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $PP > $R/$O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $PP > $R/$O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/$O/pmc_sq -- python3 $PP > $R/$O/pmc_sq.log 2>&1 || exit 1
cd $R
python tools/pmc_summary.py $O/pmc_summary_fullscan.json $O/pmc_fetch $O/pmc_write $O/pmc_sq | grep hamming
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout -k 10 200 python tools/akaze_time.py > $O/akaze_time.jsonl 2>/dev/null; cat $O/akaze_time.jsonl
timeout -k 10 200 python tools/image_lab.py 64 2 2>/dev/null | grep "profile=1" > $O/image_lab_k5_on_extracted_descriptors.txt; tail -3 $O/image_lab_k5_on_extracted_descriptors.txt
