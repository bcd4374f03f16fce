set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/run3; rm -rf $O; mkdir -p $O; cd $R
MI355X_E2E_TIMING=1 python tools/native_end_to_end.py --init > $O/native_e2e.log 2>&1; tail -8 $O/native_e2e.log
timeout 400 python tools/fuzz_requests.py 300 > $O/fuzz_requests.log 2>&1; echo "fuzz_requests rc=$?"; tail -3 $O/fuzz_requests.log
timeout 400 python tools/fuzz_extreme.py 1200 0 ordinary > $O/fuzz_ordinary.log 2>&1; echo "fuzz ordinary rc=$?"; tail -3 $O/fuzz_ordinary.log
timeout 400 python tools/fuzz_extreme.py 1600 7 extreme > $O/fuzz_extreme.log 2>&1; echo "fuzz extreme rc=$?"; tail -3 $O/fuzz_extreme.log
timeout 400 python tools/fuzz_colpart.py 200 > $O/fuzz_colpart.log 2>&1; echo "fuzz colpart rc=$?"; tail -3 $O/fuzz_colpart.log
timeout 400 python tools/fuzz_two_phase.py 300 > $O/fuzz_two_phase.log 2>&1; echo "fuzz two-phase rc=$?"; tail -3 $O/fuzz_two_phase.log
