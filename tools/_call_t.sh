python -m pytest tests/test_gpu_wait_and_prime.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | head -20
