#!/bin/bash
# Does the sweep's cache policy move the look-ahead?  (round-5 review, lever b: keep what the look-ahead reads next --
# objective row, RHS column, one column, one row of the tableau the sweep has just written -- resident in the
# Infinity Cache.)  Config 3 steady state, kernels by HIP events: look-ahead + sweep per block of 24, for the ring
# sweep's four policies.  Run on the GPU box: bash tools/la_policy_ab.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( for ring in 1 2 3; do echo "== sweepw_ring mode $ring (1: by size = NT loads at config 3; 2: no NT; 3: NT stores only)"; python tools/steady_gap.py --repeat 2 --pivots 4200 --ring $ring 2>&1 | grep "kernels la"; done
  echo "== --nt 1 (NT loads + stores)"; python tools/steady_gap.py --repeat 2 --pivots 4200 --nt 1 2>&1 | grep "kernels la"
  echo "== --nt 0"; python tools/steady_gap.py --repeat 2 --pivots 4200 --nt 0 2>&1 | grep "kernels la" ) > gpurun_out/la_mall_policy_ab.txt 2>&1
cat gpurun_out/la_mall_policy_ab.txt | cut -c1-330
