mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/gpurun_out/prof_akaze_tl_batch8; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/akaze_trace_batch.py > $GRAFT_REPO_ROOT/gpurun_out/akaze_trace_batch8.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/akaze_trace_report.py $(ls $d/*/*kernel_trace.csv | head -1) > $GRAFT_REPO_ROOT/gpurun_out/akaze_timeline_batch8.txt
