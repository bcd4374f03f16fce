"""A request of n pivots that is not a whole number of blocks (the driver's bench run: 20 after 5):
remainder spread evenly over the blocks vs full blocks + remainder.
    python tools/tail_policy.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
for n in (20, 24, 17, 40):
    for pol in (0, 1):
        L.mi355x_tune_set_tail_policy(pol)
        ts = []
        for rep in range(7):
            h = ctypes.c_void_p()
            lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), 8192, 4096, lp.synth.seed_for(3), 0, -1, 0), "create")
            npv = ctypes.c_int64(0)
            lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 5, 1), "warm")
            L.mi355x_tab_sync(h, ctypes.byref(npv))
            t0 = time.perf_counter()
            lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, n, 0), "run")
            L.mi355x_tab_sync(h, ctypes.byref(npv))
            ts.append(time.perf_counter() - t0)
            L.mi355x_tab_destroy(h)
        ts.sort()
        print("n=%2d policy %d (%s): best %.1f us median %.1f us = %.0f pivots/s (median)" % (
            n, pol, "full blocks + remainder" if pol else "spread evenly", ts[0] * 1e6, ts[3] * 1e6, n / ts[3]), flush=True)
