python -m pytest tests/test_gpu_wide_blocks.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_dispatch_boundaries.py -x -q -k "100_mib or 768" 2>&1 | tail -2
python tools/steady_gap.py --repeat 3 --pivots 4200 2>&1 | tail -3
bash tools/_call5.sh 2>&1 | grep -E "ring=|sweepw|la_block<24>"
