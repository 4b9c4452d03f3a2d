mkdir -p gpurun_out; rm -rf gpurun_out/prof_akaze_tr
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_akaze_tr -- python3 $GRAFT_REPO_ROOT/tools/akaze_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/prof_akaze_tr/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the VGA images come first: find runs of kernels separated by host gaps > 300us -> one image each
imgs=[]; cur=[rows[0]]
for a,b in zip(rows,rows[1:]):
    if int(b["Start_Timestamp"])-int(a["End_Timestamp"])>300000: imgs.append(cur); cur=[]
    cur.append(b)
imgs.append(cur)
for k,im in enumerate(imgs):
    if len(im)<50: continue
    t0=int(im[0]["Start_Timestamp"]); t1=int(im[-1]["End_Timestamp"])
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in im)
    gaps=[int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a,b in zip(im,im[1:])]
    print(f"run {k}: kernels {len(im)} span {(t1-t0)/1e3:.0f} us busy {busy/1e3:.0f} us gaps>2us {sum(1 for g in gaps if g>2000)} gap_total {sum(g for g in gaps if g>0)/1e3:.0f} us max_gap {max(gaps)/1e3:.0f} us")
PY
