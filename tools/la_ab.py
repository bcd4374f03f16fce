"""Product-build timing of the blocked solve loop on config 3 (or n m given): HIP events around
every look-ahead launch and every sweep, plus wall-clock pivots/s -- the A/B harness for kernel
changes (tools/la_timing.py is the instrumented, intrusive view).

    python tools/la_ab.py [pivots] [n_vars n_cons]
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8192, 4096)


def run(label):
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
    npv = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    # untimed-by-events wall clock first (events add a little host work)
    t0 = time.perf_counter()
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, K, 0), "run")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    dt = time.perf_counter() - t0
    L.mi355x_tab_timing_enable(h, 1)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, K, 0), "run")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    out = []
    for kind in (1, 0):
        nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        L.mi355x_tab_timing_read_kind(h, kind, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
        out.append((nl.value, sm.value / max(nl.value, 1) * 1e3, mn.value * 1e3))
    print("%-26s %dx%d: %7.0f pivots/s | look-ahead avg %6.1f min %6.1f us (%d) | sweep avg %6.1f min %6.1f us | lost=%d"
          % (label, n, m, K / dt, out[0][1], out[0][2], out[0][0], out[1][1], out[1][2], L.mi355x_tab_la_lost(h)), flush=True)
    L.mi355x_tab_destroy(h)


for one_xcd in (1, 0):
    L.mi355x_tune_set_la_one_xcd(one_xcd)
    run("persistent one_xcd=%d" % one_xcd)
L.mi355x_tune_set_la_one_xcd(1)
L.mi355x_tune_set_lookahead_mode(1)
run("two launches per step")
L.mi355x_tune_set_lookahead_mode(0)
L.mi355x_tune_set_block(1)
run("per-pivot (block 1)")
L.mi355x_tune_set_block(0)
