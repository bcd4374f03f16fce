set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/blk
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/blk/pytest_full.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/blk/pytest_full.log | cut -c1-300
for b in 4 8 16; do
timeout 300 python bench.py --no-cpu-baseline --workload cfg2 --warmup 32 --block $b > gpurun_out/blk/bench_cfg2.log 2>&1; echo "cfg2 block $b rc=$?"; tail -1 gpurun_out/blk/bench_cfg2.log | cut -c1-100
done
