# The launches of ONE image-world frame taken alone (K9 -> BoW chain -> shortlist -> path), from a rocprofv3 kernel trace
# of the image-in leg on a small map: start, duration and gap of every launch -> gpurun_out/image_frame_alone_timeline.txt
# (LEG=1080p: the 1080p leg runs after the VGA one and the LAST frame alone is a 1080p frame -> ..._1080p.txt)
if [ "$LEG" = 1080p ]; then LEG_ARGS="--image-views-1080p 300 --image-steps-1080p 1 --batch-1080p 32"; SFX=_1080p; else LEG_ARGS="--no-image-in-1080p"; SFX=""; fi
export SFX
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
d=$GRAFT_REPO_ROOT/gpurun_out/prof_img_lat; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --image-in-only $LEG_ARGS --views 1000 --image-views 1000 --image-steps 1 --batch 64 --image-oracle-frames 0 --no-cpu-baseline $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/image_lat_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/image_lat_bench.log || exit 1
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/prof_img_lat/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r):
    n = r["Kernel_Name"]
    m = re.search(r"k_gang<sfmloc::\(anonymous namespace\)::(\w+)", n)
    if m: return "gang:" + m.group(1)
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("sfmloc::", "").replace("void ", "")
starts = [i for i, r in enumerate(rows) if "k_pre_rows" in r["Kernel_Name"] or "PreRowsBody" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0; busy = 0
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:7.1f} gap {(e - s) / 1e3:8.1f} us  {name(r)[:60]}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
    busy += e - s; prev = max(prev, e)
out.append(f"total {(prev - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, launches {len(rows)}")
import os
open("gpurun_out/image_frame_alone_timeline" + os.environ.get("SFX", "") + ".txt", "w").write("\n".join(out) + "\n")
PY
