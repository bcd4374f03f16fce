#!/bin/bash
# The config-3 part of tools/gpu_evidence.sh (what a change of the single-tableau sweep touches): the whole GPU
# suite, the default + short bench line, the rocprofv3 kernel summary of the same bench command, the PMC passes
# (separate runs), the steady-state A/Bs and the pass's timeline.  tools/summarize_evidence.py copies what should
# be judged into profiles/ (it leaves alone what this run did not produce).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
grep -A 14 "slowest 12 durations" $O/pytest_gpu.log > $O/pytest_gpu_summary.log; tail -1 $O/pytest_gpu.log >> $O/pytest_gpu_summary.log
python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-400
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.log 2>&1; echo "bench (driver flags) rc=$?"
MI355X_E2E_TIMING=1 python tools/native_end_to_end.py --init > $O/native_end_to_end.log 2>&1; echo "native end to end rc=$?"; tail -4 $O/native_end_to_end.log
(python tools/steady_gap.py --repeat 3 --pivots 4200; python tools/steady_gap.py --repeat 3 --pivots 4200 --skew 0; python tools/steady_gap.py --repeat 3 --pivots 4200 --load 0.25; python tools/steady_gap.py --repeat 3 --pivots 4200 --ring 0; python tools/steady_gap.py --repeat 4 --pivots 20 --events 0) 2>&1 | grep -v amdgpu.ids > $O/steady_gap.log; echo "steady gap rc=$?"
(for sk in -1 0; do echo "== mi355x_tune_set_sweep_skew($sk)"; python tools/sweep_timeline.py --skew $sk; done) 2>&1 | grep -v amdgpu.ids > $O/sweep_timeline.log; echo "sweep timeline rc=$?"
python tools/wide_block_ab.py 2>&1 | grep "us per pivot" > $O/wide_block_ab.log; echo "wide block A/B rc=$?"
(timeout 300 python tools/fuzz_requests.py 3000; timeout 300 python tools/fuzz_extreme.py 8000 100 ordinary) 2>&1 | grep "cases,\|batches,\|lists,\|MISMATCH" > $O/fuzz_totals.log; echo "fuzzers done"; cat $O/fuzz_totals.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats -- python $R/bench.py --no-cpu-baseline --no-per-pivot --no-other-configs --no-prime > $O/kernel_stats.log 2>&1; echo "rocprof stats rc=$?"
for ring in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ring_ab_$ring -- python $R/tools/steady_gap.py --repeat 2 --pivots 4200 --events 0 --prime 0 --ring $ring > /dev/null 2>&1
  f=$(find $O/ring_ab_$ring -name "*kernel_stats.csv" | head -1); echo "ring=$ring (tools/steady_gap.py --pivots 4200 --events 0 --ring $ring under rocprofv3 --kernel-trace --stats)"; head -4 $f
done > $O/ring_ab_kernel_stats.log 2>&1; echo "ring A/B kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/tools/pmc_probe.py 64 > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_SQ -- python $R/tools/pmc_probe.py 64 > $O/pmc_SQ.log 2>&1; echo "pmc SQ rc=$?"
