set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/run1; rm -rf $O; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
MI355X_E2E_TIMING=1 python tools/native_end_to_end.py --init > $O/native_e2e.log 2>&1; tail -30 $O/native_e2e.log
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
timeout 600 python tools/resident_lds_ab.py > $O/resident_lds_ab.log 2>&1; grep "strip mode\|identical" $O/resident_lds_ab.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats -- python $R/bench.py --no-cpu-baseline --no-per-pivot --no-other-configs > $O/kernel_stats.log 2>&1; echo "rocprof stats rc=$?"
find $O/kernel_stats -name "*kernel_stats.csv" | head -1 | xargs head -8
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_resident_SQ -- python $R/tools/pmc_probe_resident.py > $O/pmc_resident_SQ.log 2>&1; echo "pmc resident SQ rc=$?"
find $O/pmc_resident_SQ -name "*counter_collection.csv" | head -1 | xargs grep -i "k_resident" | head -40
