set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/run4; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py tests/test_gpu_property.py tests/test_gpu_nan_rules.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/resident_ab.py > $O/resident_ab.log 2>&1; grep "poll mode" $O/resident_ab.log
timeout 600 python tools/resident_lds_ab.py > $O/resident_lds_ab.log 2>&1; grep "strip mode\|identical" $O/resident_lds_ab.log; grep "rep 2" $O/resident_lds_ab.log | head -4
timeout 300 python bench.py --workload cfg2 --steps 128 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --workload cfg4 --batch-lps 1024 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --workload cfg4 2>&1 | tail -1 | cut -c1-200
timeout 600 python tools/fuzz_requests.py 3000 > $O/fuzz_requests.log 2>&1; echo "fuzz_requests rc=$?"; tail -2 $O/fuzz_requests.log
timeout 600 python tools/fuzz_extreme.py 8000 100 ordinary > $O/fuzz_ordinary.log 2>&1; echo "fuzz ordinary rc=$?"; tail -2 $O/fuzz_ordinary.log
timeout 600 python tools/fuzz_extreme.py 12000 107 extreme > $O/fuzz_extreme.log 2>&1; echo "fuzz extreme rc=$?"; tail -2 $O/fuzz_extreme.log
timeout 600 python tools/fuzz_colpart.py 1500 > $O/fuzz_colpart.log 2>&1; echo "fuzz colpart rc=$?"; tail -2 $O/fuzz_colpart.log
timeout 600 python tools/fuzz_two_phase.py 2000 > $O/fuzz_two_phase.log 2>&1; echo "fuzz two-phase rc=$?"; tail -2 $O/fuzz_two_phase.log
timeout 600 python tools/fuzz_batch_extreme.py 200 > $O/fuzz_batch_extreme.log 2>&1; echo "fuzz batch extreme rc=$?"; tail -2 $O/fuzz_batch_extreme.log
timeout 600 python tools/fuzz_colpart_extreme.py 600 > $O/fuzz_colpart_extreme.log 2>&1; echo "fuzz colpart extreme rc=$?"; tail -2 $O/fuzz_colpart_extreme.log
timeout 600 python tools/fuzz_colpart_two_phase.py 600 > $O/fuzz_colpart_two_phase.log 2>&1; echo "fuzz colpart two-phase rc=$?"; tail -2 $O/fuzz_colpart_two_phase.log
