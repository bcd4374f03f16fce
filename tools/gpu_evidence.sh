#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, the default bench line and the driver's short
# one, the rocprofv3 kernel summary of the same bench command, the PMC passes (separate runs,
# kernel-trace only -- never combined with a sys / runtime trace), config 4 and the per-pivot path.
# Every rocprofv3 call runs under its own `timeout` (a profiler that dies can sit in its signal
# handler for the rest of the lease).  Everything lands under gpurun_out/evidence/ ; tools/summarize_evidence.py copies what should be
# judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
grep -A 14 "slowest 12 durations" $O/pytest_gpu.log > $O/pytest_gpu_summary.log; tail -1 $O/pytest_gpu.log >> $O/pytest_gpu_summary.log
python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-400
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.log 2>&1; echo "bench (driver flags) rc=$?"
python bench.py --workload cfg4 > $O/bench_cfg4.log 2>&1; echo "bench cfg4 rc=$?"; tail -1 $O/bench_cfg4.log | cut -c1-300
python bench.py --workload cfg4 --batch-lps 1024 > $O/bench_cfg4_1024.log 2>&1; echo "bench cfg4 x1024 rc=$?"
python bench.py --workload cfg2 --steps 128 --no-cpu-baseline > $O/bench_cfg2.log 2>&1; echo "bench cfg2 rc=$?"
MI355X_E2E_TIMING=1 python tools/native_end_to_end.py --init > $O/native_end_to_end.log 2>&1; echo "native end to end rc=$?"; tail -4 $O/native_end_to_end.log
python bench.py --workload colpart --steps 112 --warmup 28 > $O/bench_colpart_1gpu.log 2>&1; echo "bench colpart rc=$?"
(for b in 16 24 28; do python tools/shard_step_cost.py 336 $b; done) 2>&1 | grep -E "per sweep|us per pivot" > $O/shard_step_cost.log; echo "shard step cost rc=$?"
python tools/wide_block_ab.py 2>&1 | grep "us per pivot" > $O/wide_block_ab.log; echo "wide block A/B rc=$?"
python tools/la_wide_ab.py 2>&1 | grep "us/pivot" > $O/la_wide_ab.log; echo "wide persistent look-ahead A/B rc=$?"
(python tools/steady_gap.py --repeat 3 --pivots 4200; python tools/steady_gap.py --repeat 3 --pivots 4200 --load 0.25; python tools/steady_gap.py --repeat 3 --pivots 4200 --ring 0; python tools/steady_gap.py --repeat 3 --pivots 4200 --wait 0; python tools/steady_gap.py --repeat 4 --pivots 20 --events 0; python tools/steady_gap.py --repeat 3 --pivots 4200 --xmap 1) 2>&1 | grep -v amdgpu.ids > $O/steady_gap.log; echo "steady gap rc=$?"
python tools/la_timing.py 2>&1 | grep -v amdgpu.ids | head -26 > $O/la_timing.log; echo "la timing rc=$?"
(cd tools/microbench && for b in sweep_lds sweep32 shard_step_skel; do [ -x $b ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o $b $b.hip; done) > /dev/null 2>&1
(cd tools/microbench && ./sweep_lds && ./sweep_lds 32769 8208) 2>&1 | grep -v "^    " > $O/sweep_lds_microbench.log; echo "sweep_lds microbench rc=$?"
(cd tools/microbench && ./sweep32 && ./sweep32 32769 65552 && ./sweep32 4097 8208) > $O/sweep32_microbench.log 2>&1; echo "sweep32 microbench rc=$?"
(cd tools/microbench && timeout 250 ./shard_step_skel) > $O/shard_step_skeleton.log 2>&1; echo "shard step skeleton rc=$?"
(python tools/resident_timing.py; python tools/resident_timing.py 512 256) 2>&1 | grep -E "us/pivot|inside" > $O/resident_timing.log; echo "resident timing rc=$?"
python tools/resident_ab.py 2>&1 | grep "poll mode" > $O/resident_ab.log; echo "resident A/B rc=$?"
python tools/resident_lds_ab.py 2>&1 | grep -v "^/opt" > $O/resident_lds_ab.log; echo "resident LDS-strip A/B rc=$?"
# robustness: the fuzzers on every path (wide blocks, exchange modes, two-phase, batches), totals only
(timeout 300 python tools/fuzz_requests.py 3000; timeout 300 python tools/fuzz_extreme.py 8000 100 ordinary; timeout 300 python tools/fuzz_extreme.py 12000 107 extreme
 timeout 300 python tools/fuzz_colpart.py 1500; timeout 300 python tools/fuzz_two_phase.py 2000; timeout 300 python tools/fuzz_batch_extreme.py 200
 timeout 300 python tools/fuzz_colpart_extreme.py 600; timeout 300 python tools/fuzz_colpart_two_phase.py 600; timeout 300 python tools/fuzz_solve_problems.py 300) 2>&1 | grep "cases,\|batches,\|lists,\|MISMATCH" > $O/fuzz_totals.log; echo "fuzzers done"; cat $O/fuzz_totals.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats -- python $R/bench.py --no-cpu-baseline --no-per-pivot --no-other-configs --no-prime > $O/kernel_stats.log 2>&1; echo "rocprof stats rc=$?"
for ring in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ring_ab_$ring -- python $R/tools/steady_gap.py --repeat 2 --pivots 4200 --events 0 --prime 0 --ring $ring > /dev/null 2>&1
  f=$(find $O/ring_ab_$ring -name "*kernel_stats.csv" | head -1); echo "ring=$ring (tools/steady_gap.py --pivots 4200 --events 0 --ring $ring under rocprofv3 --kernel-trace --stats)"; head -4 $f
done > $O/ring_ab_kernel_stats.log 2>&1; echo "ring A/B kernel stats done"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats_cfg4 -- python $R/bench.py --workload cfg4 > $O/kernel_stats_cfg4.log 2>&1; echo "rocprof stats cfg4 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats_cfg2 -- python $R/bench.py --workload cfg2 --steps 128 --no-cpu-baseline > $O/kernel_stats_cfg2.log 2>&1; echo "rocprof stats cfg2 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats_perpivot -- python $R/bench.py --block 1 --steps 320 --no-cpu-baseline --no-per-pivot > $O/kernel_stats_perpivot.log 2>&1; echo "rocprof stats per-pivot rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/tools/pmc_probe.py 64 > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_perpivot_$c -- python $R/tools/pmc_probe.py 24 1 > $O/pmc_perpivot_$c.log 2>&1; echo "pmc per-pivot $c rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats_cfg5 -- python $R/bench.py --workload colpart --steps 112 --warmup 28 > $O/kernel_stats_cfg5.log 2>&1; echo "rocprof stats cfg5 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats_shard -- python $R/tools/pmc_probe.py 96 24 0 8192 32768 > $O/kernel_stats_shard.log 2>&1; echo "rocprof stats shard rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_shard_$c -- python $R/tools/pmc_probe.py 96 24 0 8192 32768 > $O/pmc_shard_$c.log 2>&1; echo "pmc shard $c rc=$?"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_resident_$c -- python $R/tools/pmc_probe_resident.py > $O/pmc_resident_$c.log 2>&1; echo "pmc resident $c rc=$?"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_SQ -- python $R/tools/pmc_probe.py 64 > $O/pmc_SQ.log 2>&1; echo "pmc SQ rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_resident_SQ -- python $R/tools/pmc_probe_resident.py > $O/pmc_resident_SQ.log 2>&1; echo "pmc resident SQ rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats_shard_step -- python $R/tools/shard_step_cost.py 336 24 "two launches" > $O/kernel_stats_shard_step.log 2>&1; echo "rocprof stats shard step rc=$?"
# round 6: the shard's look-ahead of a whole block as ONE persistent launch (k_shard_la_block) on the same shard
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernel_stats_shard_block -- python $R/tools/shard_step_cost.py 336 24 "persistent" > $O/kernel_stats_shard_block.log 2>&1; echo "rocprof stats shard block rc=$?"
cd $R
(python tools/shard_la_timing.py 1; python tools/shard_la_timing.py 0) 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $O/shard_la_timing.log; echo "shard la timing rc=$?"
bash tools/la_policy_ab.sh > /dev/null 2>&1; cp gpurun_out/la_mall_policy_ab.txt $O/la_policy_ab.log; echo "la policy A/B done"
bash tools/ring_waves_ab.sh > $O/ring_waves_ab.log 2>&1; echo "ring waves A/B done"
