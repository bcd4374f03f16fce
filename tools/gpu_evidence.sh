#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, the default bench line, the rocprofv3 kernel
# summary of the same bench command, and the two PMC passes (separate runs, kernel-trace only).
# Everything lands under gpurun_out/evidence/ ; copy what should be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-700
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats -- python $R/bench.py --no-cpu-baseline > $O/kernel_stats.log 2>&1; echo "rocprof stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/tools/pmc_probe.py 64 > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
