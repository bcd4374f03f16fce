cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk -- $R/tools/microbench/sweep_lds > $R/gpurun_out/clk.log 2>&1
f=$(find $R/gpurun_out/clk -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for r in rows:
    name=r.get("Kernel_Name") or r.get("Kernel Name")
    val=float(r["Counter_Value"]); 
    st=float(r.get("Start_Timestamp",0)); en=float(r.get("End_Timestamp",0))
    a=agg[name]; a[0]+=1; a[1]+=val; a[2]+=(en-st)
for k,(n,v,d) in agg.items():
    if d>0: print("%-70s n=%3d  GUI_ACTIVE/launch %.0f  dur %.1f us  -> %.0f MHz" % (k[:70], n, v/n, d/n/1e3, v/d*1e3))
PY
