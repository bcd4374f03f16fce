python -m pytest tests/test_gpu_wide_blocks.py tests/test_gpu_solve_problems.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_dispatch_boundaries.py -x -q -k "100_mib or 768 or mid_band" 2>&1 | tail -2
python -m pytest tests/test_gpu_fullsize.py -x -q -k "config3_400 or config5_full_size_64_pivots_vs" 2>&1 | tail -2
python tools/steady_gap.py --repeat 3 --pivots 4200 2>&1 | tail -2 | cut -c55-150,200-330
bash tools/_call5.sh 2>&1 | grep -E "ring=|sweepw|la_block<24>"
