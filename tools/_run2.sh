set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/run2; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_resident.log 2>&1; echo "pytest resident rc=$?"; tail -5 $O/pytest_resident.log
timeout 600 python tools/resident_lds_ab.py > $O/lds_ab.log 2>&1; cat $O/lds_ab.log
timeout 300 python bench.py --workload cfg4 --batch-lps 1024 2>&1 | tail -1 | cut -c1-250
timeout 300 python bench.py --workload cfg4 2>&1 | tail -1 | cut -c1-250
