"""Host-side overhead of one blocking solve: wall time of mi355x_tab_solve against the kernels'
own durations (run under `rocprofv3 --kernel-trace` for the latter).
    python tools/solve_overhead.py [n_vars n_cons] [timing 0|1]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 512)
timing = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for rep in range(6):
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(2, rep % 2), 0, -1, 0), "create")
    k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 0, 1), "prepare")
    L.mi355x_tab_sync(h, ctypes.byref(k))
    if timing:
        L.mi355x_tab_timing_enable(h, 1)
    t0 = time.perf_counter()
    rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(k))
    dt = time.perf_counter() - t0
    print("rep %d: rc %d, %d pivots, wall %.1f us = %.3f us per pivot (%.0f pivots/s)" % (rep, rc, k.value, dt * 1e6, dt / k.value * 1e6, k.value / dt), flush=True)
    L.mi355x_tab_destroy(h)
