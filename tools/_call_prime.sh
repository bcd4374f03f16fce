python - <<'PY'
import ctypes, time, sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
for prime in (1, 0, 1):
  L.mi355x_tune_set_prime(prime)
  for rep in range(2):
    h = ctypes.c_void_p(); k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), 8192, 4096, lp.synth.seed_for(3, rep), 0, -1, 0), "c")
    lp.capi.check(L.mi355x_tab_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "s")
    t0 = time.perf_counter()
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 5, 1), "w"); L.mi355x_tab_sync(h, ctypes.byref(k))
    tw = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 20, 0), "r")
    t1 = time.perf_counter()
    L.mi355x_tab_sync(h, ctypes.byref(k))
    t2 = time.perf_counter()
    print("prime", prime, "rep", rep, "warm-up request %.1f us; enqueue %.1f us total %.1f us" % (tw*1e6, (t1-t0)*1e6, (t2-t0)*1e6), flush=True)
    L.mi355x_tab_destroy(h)
PY
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b20', d['value'], d['ms_per_step'], d['steady_state_pivots_per_s'], d['steady_state_pivots_per_s_gpu_clock'])"; done
python -m pytest tests -m gpu -x -q > gpurun_out/spin/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/spin/pytest_gpu.log | tail -3
