"""k_sweep16 / k_sweep launch-shape sweep on config 3: rows per workgroup x non-temporal x implementation.
    python tools/sweep_shapes.py [rows,rows,...] [impl,...] [nt,...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m, K = 8192, 4096, 800
TRS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (16, 32, 64)
IMPLS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (8, 4)
NTS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0, 1)
for impl in IMPLS:
    for nt in NTS:
        for tr in TRS:
            L.mi355x_tune_set_sweep_impl(0); L.mi355x_tune_set_sweep_impl(impl)
            L.mi355x_tune_set_sweep_shape(tr, nt)
            h = ctypes.c_void_p()
            lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
            npv = ctypes.c_int64(0)
            lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
            L.mi355x_tab_sync(h, ctypes.byref(npv))
            L.mi355x_tab_timing_enable(h, 1)
            lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, K, 0), "run")
            L.mi355x_tab_sync(h, ctypes.byref(npv))
            nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
            L.mi355x_tab_timing_read_kind(h, 0, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
            print("impl=%s nt=%d tr=%3d: sweep avg %6.1f min %6.1f us" % ("k_sweep16<U=%d>" % impl if impl in (4, 8) else ("k_sweep16" if impl == 0 else "k_sweep  "), nt, tr, sm.value / nl.value * 1e3, mn.value * 1e3), flush=True)
            L.mi355x_tab_destroy(h)
