mkdir -p gpurun_out; rm -rf gpurun_out/prof_akaze_tr
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_akaze_tr -- python3 $GRAFT_REPO_ROOT/tools/akaze_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob,re,collections
f=glob.glob("gpurun_out/prof_akaze_tr/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the VGA images come first: find runs of kernels separated by host gaps > 300us -> one image each
imgs=[]; cur=[rows[0]]
for a,b in zip(rows,rows[1:]):
    if int(b["Start_Timestamp"])-int(a["End_Timestamp"])>300000: imgs.append(cur); cur=[]
    cur.append(b)
imgs.append(cur)
shown=0
for k,im in enumerate(imgs):
    if len(im)<30: continue
    t0=int(im[0]["Start_Timestamp"]); t1=int(im[-1]["End_Timestamp"])
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in im)
    gaps=[int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a,b in zip(im,im[1:])]
    print(f"run {k}: kernels {len(im)} span {(t1-t0)/1e3:.0f} us busy {busy/1e3:.0f} us gaps>2us {sum(1 for g in gaps if g>2000)} gap_total {sum(g for g in gaps if g>0)/1e3:.0f} us max_gap {max(gaps)/1e3:.0f} us")
    if shown<1 and k>=4:
        shown+=1
        by=collections.OrderedDict()
        for r in im:
            n=re.sub(r"\(.*","",r["Kernel_Name"].replace("(anonymous namespace)::","")).replace("sfmloc::","").replace("void ","")
            d=by.setdefault(n,[0,0]); d[0]+=1; d[1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        for n,(c,t) in by.items(): print(f"    {n[:40]:40s} x{c:3d}  {t/1e3:8.1f} us")
        prev=None
        for r in im:
            n=re.sub(r"\(.*","",r["Kernel_Name"].replace("(anonymous namespace)::","")).replace("sfmloc::","").replace("void ","")
            s_,e_=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
            print(f"      {n[:30]:30s} start {(s_-t0)/1e3:8.1f} dur {(e_-s_)/1e3:7.1f} gap {((s_-prev)/1e3 if prev else 0):6.1f} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
            prev=e_
PY
