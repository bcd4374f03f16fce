set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/run5; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/shard_step_cost.py 336 24 "two launches" > $O/stats.log 2>&1; echo "rc=$?"; tail -2 $O/stats.log
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -40 | cut -c1-220
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc -- python $R/tools/shard_step_cost.py 48 24 "two launches" > $O/pmc.log 2>&1; echo "pmc rc=$?"
python - <<'PY'
import csv,collections,glob
f=glob.glob('/root/repo/gpurun_out/run5/pmc/*/*counter_collection.csv')[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'shard' in k or 'sweepw' in k:
        m={a:sum(b)/len(b) for a,b in v.items()}
        print(k,len(v['SQ_WAVES']),{a:int(b) for a,b in m.items()})
PY
