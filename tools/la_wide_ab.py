"""Tableaux of 33 ... 64 look-ahead workgroups (8193 ... 16384 rows / up to 32768 stored columns): the
persistent look-ahead in its record-per-workgroup form (default since round 5) against the two-launch
look-ahead (mi355x_tune_set_lookahead_mode(1): what these shapes ran before), pivots per second by the
wall clock over whole blocks.

    python tools/la_wide_ab.py [pivots]
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 672


def run(n, m, mode):
    L.mi355x_tune_set_lookahead_mode(mode)
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
    npv = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 168, 1), "warm")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, K, 0), "run")
        L.mi355x_tab_sync(h, ctypes.byref(npv))
        best = min(best, time.perf_counter() - t0)
    bs, lost = L.mi355x_tab_block_size(h), L.mi355x_tab_la_lost(h)
    c = (ctypes.c_int64 * 8)(); L.mi355x_tab_path_counts(h, c)
    L.mi355x_tab_destroy(h)
    L.mi355x_tune_set_lookahead_mode(0)
    return best / K * 1e6, bs, lost, c[1], c[2]


for n, m in ((8192, 4096), (6000, 8200), (9000, 9000), (12000, 11000), (20000, 6000), (16000, 16000), (24000, 12000), (32000, 16000)):
    ld = (n + 1 + 15) // 16 * 16
    nw = (max(m + 1, ld // 2) + 255) // 256
    a = run(n, m, 0)
    b = run(n, m, 1)
    print("%5d x %5d  %2d workgroups  %6.0f MB stored | persistent %6.2f us/pivot (%d per pass, %d launches, lost=%d) | two-launch %6.2f us/pivot (%d per pass) | x %.2f"
          % (n, m, nw, (m + 1) * ld * 8 / 1e6, a[0], a[1], a[3], a[2], b[0], b[1], b[0] / a[0]), flush=True)
