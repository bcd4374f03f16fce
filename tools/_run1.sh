set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/blk
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked_pivoting" > gpurun_out/blk/pytest_blk.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/blk/pytest_blk.log
for cfg in "16 16" "16 12"; do
  set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --steps 1920 --warmup 32 --block $1 --sweep-tr $2 > gpurun_out/blk/bench_b$1_tr$2.log 2>&1; echo "block $1 tr $2 rc=$?"; tail -1 gpurun_out/blk/bench_b$1_tr$2.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_avg_us'], r['roofline']['achieved'])"
done
timeout 600 python bench.py --no-cpu-baseline --workload cfg5 --steps 64 --warmup 16 > gpurun_out/blk/bench_cfg5.log 2>&1; echo "cfg5 rc=$?"; tail -1 gpurun_out/blk/bench_cfg5.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --workload cfg2 --steps 200 --warmup 16 > gpurun_out/blk/bench_cfg2.log 2>&1; echo "cfg2 rc=$?"; tail -1 gpurun_out/blk/bench_cfg2.log | cut -c1-300
