python -m pytest tests/test_gpu_native_route.py tests/test_gpu_cancel.py tests/test_gpu_solve_problems.py tests/test_gpu_fullsize.py::test_plain_c_client_on_the_gpu -q 2>&1 | tail -15
python -m pytest tests/test_gpu_dispatch_boundaries.py -q 2>&1 | tail -30
mkdir -p gpurun_out/gap
python tools/steady_gap.py --repeat 3 --pivots 4200 > gpurun_out/gap/v2_unloaded.log 2>&1
python tools/steady_gap.py --repeat 3 --pivots 4200 --load 0.25 > gpurun_out/gap/v2_load025.log 2>&1
python tools/steady_gap.py --repeat 3 --pivots 4200 --load 1 > gpurun_out/gap/v2_load1.log 2>&1
tail -n 5 gpurun_out/gap/v2_*.log
