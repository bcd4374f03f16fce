"""Two look-ahead blocks per sweep (mi355x_tune_set_double_block): bit-exactness against the oracle and throughput.
    python tools/pipeline_check.py [parity|speed|all]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def solve_vs_oracle(n, m, seed, cap=0):
    M, b = lp.synth.tableau(n, m, seed)
    Mo, bo = M.copy(), b.copy()
    so, no, trace = oracle.solve(Mo, bo, max_pivots=cap, trace_cap=1 << 16, omp=M.size > 1 << 20)
    h = ctypes.c_void_p()
    R, C = M.shape
    lp.capi.check(L.mi355x_tab_create(ctypes.byref(h), R, C, ptr(M), ptr(b), 0), "create")
    k = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(h, 1, 1024.0, cap, ctypes.byref(k))
    ec = np.empty(max(no, 1), dtype=np.int64); cr = np.empty(max(no, 1), dtype=np.int64); nn = ctypes.c_int64(0)
    L.mi355x_tab_trace(h, ptr(ec), ptr(cr), no, ctypes.byref(nn))
    got = np.stack([ec[:no], cr[:no]], axis=1)
    G = np.empty_like(M); bg = np.empty_like(b)
    lp.capi.check(L.mi355x_tab_download(h, ptr(G), ptr(bg), None, None), "download")
    ok = (rc, k.value) == (so, no) and np.array_equal(got, trace) and np.array_equal(G.view(np.int64), Mo.view(np.int64)) and np.array_equal(bg, bo)
    d = np.where((got != trace).any(axis=1))[0]
    print("  %5d x %-5d seed %-12d cap %-5d: status %d/%d pivots %d/%d  %s%s" % (n, m, seed, cap, rc, so, k.value, no, "bitwise ok" if ok else "MISMATCH",
          "" if ok else " first differing pivot %s, tableau words differing %d" % (d[:3], int((G.view(np.int64) != Mo.view(np.int64)).sum()))), flush=True)
    L.mi355x_tab_destroy(h)
    return ok


if what in ("parity", "all"):
    for on in (1,):
        L.mi355x_tune_set_double_block(on)
        print("double blocks=%d" % on, flush=True)
        good = True
        for (n, m, cfg, s, cap) in [(600, 300, 2, 0, 0), (600, 300, 2, 1, 100), (600, 300, 2, 1, 17), (600, 300, 2, 1, 33), (600, 300, 2, 1, 48),
                                    (1024, 512, 2, 3, 0), (2048, 1024, 2, 5, 333), (27, 27, 2, 7, 0), (40, 31, 2, 8, 0),
                                    (300, 290, 5, 2, 0), (8192, 4096, 3, 0, 400), (8192, 4096, 3, 1, 417)]:
            good &= solve_vs_oracle(n, m, lp.synth.seed_for(cfg, s), cap)
        print("parity:", "ALL OK" if good else "FAILED", flush=True)

if what in ("speed", "all"):
    n, m = 8192, 4096
    for on in (0, 1):
        L.mi355x_tune_set_double_block(on)
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
        npv = ctypes.c_int64(0)
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
        L.mi355x_tab_sync(h, ctypes.byref(npv))
        L.mi355x_tab_timing_enable(h, 4)
        K = 3200
        t0 = time.perf_counter()
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, K, 0), "run")
        rc = L.mi355x_tab_sync(h, ctypes.byref(npv))
        dt = time.perf_counter() - t0
        out = []
        for kind in (1, 0):
            nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
            L.mi355x_tab_timing_read_kind(h, kind, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
            out.append("%s avg %6.1f min %6.1f us (%d)" % ("look-ahead" if kind else "sweep", sm.value / max(nl.value, 1) * 1e3, mn.value * 1e3, nl.value))
        print("double blocks=%d: %d pivots (rc %d, total %d) in %.2f ms = %.0f pivots/s | %s | lost=%d" % (on, K, rc, npv.value, dt * 1e3, K / dt, " | ".join(out), L.mi355x_tab_la_lost(h)), flush=True)
        L.mi355x_tab_destroy(h)
