cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for ring in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ring$ring -- python $R/tools/steady_gap.py --repeat 2 --pivots 4200 --events 0 --ring $ring > $R/gpurun_out/ring$ring.log 2>&1
  f=$(find $R/gpurun_out/ring$ring -name "*kernel_stats.csv" | head -1)
  echo "ring=$ring"; head -8 $f | cut -c1-160
done
