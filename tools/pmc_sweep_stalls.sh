#!/bin/bash
# SQ-level counters of the sweep kernel (where do its waves spend their time?).  Every rocprofv3
# call runs under its own `timeout`: a counter set the profiler does not like can otherwise hang in
# its signal handler for the rest of the GPU lease (TCP_* / TCC_EA0_* sets did exactly that here).
#   bash tools/pmc_sweep_stalls.sh            -> gpurun_out/pmc2/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc2; mkdir -p $O
for set in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SALU SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$tag -- python $R/tools/pmc_probe.py 64 16 > $O/$tag.log 2>&1; echo "$tag rc=$?"
done
python - <<'PY'
import csv, collections, glob, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/pmc2'
for f in sorted(glob.glob(O + '/*/**/*counter_collection.csv', recursive=True)):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in d.items():
        if 'sweep16' in k:
            for c, x in v.items():
                print("%-28s %16.1f (n=%d)" % (c, sum(x) / len(x), len(x)))
PY
