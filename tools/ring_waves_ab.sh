#!/bin/bash
# A/B of k_sweepw_ring at 3 (product) and 4 waves per SIMD (-DMI355X_RING_MIN_WAVES=4: the compiler must fit 128
# VGPRs, i.e. spill part of the 24 prow pairs), config 3 steady state; and the persistent look-ahead's per-phase
# clocks with its workgroups on one XCD (product) and spread over the chip (the column gather through eight miss
# paths instead of one).  Run on the GPU box: bash tools/ring_waves_ab.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
python - <<'PY'
import os, sys
sys.path.insert(0, "linear-programming_amd")
import build as b
b.build(extra_flags=["-DMI355X_RING_MIN_WAVES=4"], out="gpurun_out/libmi355x_simplex_ring4.so")
PY
echo "== product (3 waves per SIMD, 164 VGPRs)"; python tools/steady_gap.py --repeat 2 --pivots 4200 2>&1 | grep "kernels la" | cut -c1-330
echo "== -DMI355X_RING_MIN_WAVES=4"; MI355X_SIMPLEX_LIB=$PWD/gpurun_out/libmi355x_simplex_ring4.so python tools/steady_gap.py --repeat 2 --pivots 4200 2>&1 | grep "kernels la" | cut -c1-330
echo "== look-ahead per-phase clocks, one XCD"; python tools/la_timing.py 1 2>&1 | grep -v amdgpu.ids | head -12
echo "== look-ahead per-phase clocks, spread over the XCDs"; python tools/la_timing.py 0 2>&1 | grep -v amdgpu.ids | head -12
