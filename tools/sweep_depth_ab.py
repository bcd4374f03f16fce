"""k_sweep16 on config 3 with a step's rows requested ONE step ahead (two sets of four rows per
thread: the default) against TWO steps ahead (three sets), per rows-per-workgroup.
mi355x_debug_repeat_sweep re-launches the sweep of ONE pending list.   python tools/sweep_depth_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
n, m = (8192, 4096) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
for rep in range(2):
    for impl, tr in ((5, 0), (6, 0), (6, 24), (6, 48), (6, 96), (5, 48), (5, 0), (6, 0)):
        L.mi355x_tune_set_sweep_impl(impl)
        L.mi355x_tune_set_sweep_shape(tr, -1)
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
        k = ctypes.c_int64(0)
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "4 blocks"); L.mi355x_tab_sync(h, ctypes.byref(k))
        us = ctypes.c_double(0)
        lp.capi.check(L.mi355x_debug_repeat_sweep(h, 50, ctypes.byref(us)), "repeat")
        print("%d step(s) ahead, %2d rows per workgroup (0 = default): %7.1f us per sweep of %d pending pivots"
              % (impl - 4, tr, us.value, L.mi355x_tab_block_size(h)), flush=True)
        L.mi355x_tab_destroy(h)
L.mi355x_tune_set_sweep_impl(5); L.mi355x_tune_set_sweep_shape(0, -1)
