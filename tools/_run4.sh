cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for ring in 1 2 3; do echo "== sweepw_ring mode $ring (1: by size = NT loads at config 3; 2: no NT; 3: NT stores only)"; python tools/steady_gap.py --repeat 2 --pivots 4200 --ring $ring 2>&1 | grep "kernels la"; done
  echo "== --nt 1 (NT loads + stores)"; python tools/steady_gap.py --repeat 2 --pivots 4200 --nt 1 2>&1 | grep "kernels la"
  echo "== --nt 0"; python tools/steady_gap.py --repeat 2 --pivots 4200 --nt 0 2>&1 | grep "kernels la" ) > gpurun_out/la_mall_policy_ab.txt 2>&1
cat gpurun_out/la_mall_policy_ab.txt | cut -c1-330
