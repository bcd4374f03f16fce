"""The sweep of a full block of 16 on config 3, per implementation: k_sweep16 (split issue / wait
col chunks, pend() path compiled in) against the two-launch form of the wide sweeps (k_sweepw<16>:
links as self-contained asm statements, scalar row base + lane offset addressing; k_sweepw_rest:
pivot-row groups + pricing).  mi355x_debug_repeat_sweep re-launches the sweep of ONE pending list.
    python tools/sweep_impl_ab.py [n m]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
n, m = (8192, 4096) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
for rep in range(2):
    for impl, tr in ((0, 0), (3, 0), (3, 16), (3, 64), (0, 0), (3, 0)):
        L.mi355x_tune_set_sweep_impl(impl)
        L.mi355x_tune_set_sweep_shape(tr, -1)
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
        k = ctypes.c_int64(0)
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "4 blocks"); L.mi355x_tab_sync(h, ctypes.byref(k))
        us = ctypes.c_double(0)
        lp.capi.check(L.mi355x_debug_repeat_sweep(h, 50, ctypes.byref(us)), "repeat")
        print("impl %d tr %2d: %7.1f us per sweep of %d pending pivots" % (impl, tr, us.value, L.mi355x_tab_block_size(h)), flush=True)
        L.mi355x_tab_destroy(h)
L.mi355x_tune_set_sweep_impl(0); L.mi355x_tune_set_sweep_shape(0, -1)
