cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xpmc; mkdir -p $O
for x in 0 1; do for r in 1 2; do
  PROBE_XMAP=$x PROBE_RING=$r timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/x${x}_r$r -- python $R/tools/pmc_probe.py 96 > $O/x${x}_r$r.log 2>&1; echo "rc=$?"
  python - <<PY
import csv,glob
for f in glob.glob("$O/x${x}_r$r/**/*counter_collection.csv", recursive=True):
    tot={}
    for row in csv.DictReader(open(f)):
        if "sweepw_ring" in row["Kernel_Name"] and row["Counter_Name"]=="FETCH_SIZE":
            tot.setdefault(row["Dispatch_Id"],0.0); tot[row["Dispatch_Id"]]+=float(row["Counter_Value"])
    v=sorted(tot.values())
    print("xmap=$x ring=$r launches",len(v),"FETCH_SIZE KiB each", [round(a) for a in v], "-> read MB (x2 x1024)", [round(a*2*1024/1e6,1) for a in v])
PY
done; done
