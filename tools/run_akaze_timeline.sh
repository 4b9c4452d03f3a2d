# One frame's K9 launches with start, gap and duration (rocprofv3 kernel trace of tools/akaze_trace.py):
# plain and image-world VGA frame, image-world 1080p frame -> gpurun_out/akaze_timeline_*.txt
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for v in "plain" "rich" "rich 1080p"; do
  tag=$(echo $v | tr ' ' '_'); d=$GRAFT_REPO_ROOT/gpurun_out/prof_akaze_tl_$tag; rm -rf $d
  rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/akaze_trace.py $v > $GRAFT_REPO_ROOT/gpurun_out/akaze_trace_$tag.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/tools/akaze_trace_report.py $(ls $d/*/*kernel_trace.csv | head -1) > $GRAFT_REPO_ROOT/gpurun_out/akaze_timeline_$tag.txt
done
