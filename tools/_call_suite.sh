mkdir -p gpurun_out/spin
python -m pytest tests -m gpu -x -q > gpurun_out/spin/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/spin/pytest_gpu.log | tail -3
