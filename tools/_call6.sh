python -m pytest tests/test_gpu_fullsize.py -x -q -k "config3_400 or handoff or lost_exchange" 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -x -q -k "lookahead or persistent or block" 2>&1 | tail -3
python tools/steady_gap.py --repeat 3 --pivots 4200 2>&1 | tail -3
bash tools/_call5.sh
