# The round's evidence run (one gpurun call): the GPU suite, the default bench (headline + image_in + image_in_1080p),
# rocprofv3 kernel-trace stats of the same command, PMC passes of the roofline kernel (separate runs, as the guide
# prescribes), per-frame / per-query kernel time from the trace, AKAZE times.  ROUND=r04 tags the output; copy what is to
# be judged into profiles/ under that prefix.  (GPU_MAX_HW_QUEUES is exported here: under rocprofv3 the runtime is up
# before bench.py's own setdefault runs -- ADVICE r03.  The value is the one bench.plan() chooses for the default run, 22 =
# 20 contexts + 2: with 24 exported the image-in legs lose 8 - 15 % and even the lone full scan of the real bank runs at
# 29.6 instead of 26.7 ms -- two runs each way, profiles/r04_hw_queues_22_vs_24.txt.)
R=${ROUND:-r04}; O=gpurun_out/${R}_final; mkdir -p $O
ROOT=$GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-22}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log || exit 1
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("headline", round(d["value"]), d["identical_to_single_flight"], "p50", round(d["latency_ms"]["p50"],3), "| roofline", r["bound"], round(r["frac"],3),
      "real bank", round(r["real_bank"]["valu"]["frac"],3), "| image_in", round(d["image_in"]["value"]), d["image_in"]["identical_to_single_flight"],
      "p50", round(d["image_in"]["latency_ms"]["p50"],2), "| 1080p", round(d["image_in_1080p"]["value"]), d["image_in_1080p"]["identical_to_single_flight"])
PY
cd /tmp && export TMPDIR=/tmp
PB="$ROOT/bench.py --steps 2 --warmup 1 --image-steps 2 --image-steps-1080p 1 --no-cpu-baseline --no-real-stats --image-oracle-frames 1"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python3 $PB > $ROOT/$O/bench_prof_run.log 2>&1 || exit 1
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $ROOT/$O/bench_kernel_stats.csv
T=$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_busy.py $T "P3pFinish|k_p3p_finish" > $ROOT/$O/kernel_time_per_result_busiest_window.txt 2>&1
PP="$ROOT/bench.py --steps 1 --warmup 0 --batch 16 --no-cpu-baseline --no-real-stats --no-image-in"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $PP > $ROOT/$O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $PP > $ROOT/$O/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/pmc_sq -- python3 $PP > $ROOT/$O/pmc_sq.log 2>&1 || exit 1
cd $ROOT
python tools/pmc_summary.py $O/pmc_summary_fullscan.json /tmp/pmc_fetch /tmp/pmc_write /tmp/pmc_sq | grep hamming
timeout -k 10 200 python tools/akaze_time.py > $O/akaze_time.jsonl 2>/dev/null; cat $O/akaze_time.jsonl
