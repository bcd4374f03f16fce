cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
for impl in 4 22; do
for set in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SALU SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_GATE_EN2 TA_BUSY TD_BUSY" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_EA0_RD_UNCACHED_32B TCC_TAG_STALL TCC_BUSY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/i${impl}_$tag -- python $R/tools/pmc_probe.py 64 16 $impl > $O/i${impl}_$tag.log 2>&1; echo "impl $impl $tag rc=$?"
done; done
python - <<'PY'
import csv, collections, glob, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc2'
for f in sorted(glob.glob(O+'/i*/**/*counter_collection.csv', recursive=True)):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('pmc2/')[1].split('/')[0])
    for k,v in d.items():
        if 'sweep16' in k:
            for c,x in v.items(): print("   %-28s %16.1f (n=%d)"%(c,sum(x)/len(x),len(x)))
PY
