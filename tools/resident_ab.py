"""A/B of the resident solve's record polling (wave 0 polls + LDS hand-over vs every wave polls) in ONE run on one
box: config 2 (32 workgroups) and config-4 batches (8 workgroups per LP).  python tools/resident_ab.py"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, importlib
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
def cfg2(reps=5):
    best = 0
    for rep in range(reps):
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), 1024, 512, lp.synth.seed_for(2), 0, -1, 0), "create")
        k = ctypes.c_int64(0)
        L.mi355x_tab_solve_async(h, 1, 1024.0, 0, 1); L.mi355x_tab_sync(h, ctypes.byref(k))
        t0 = time.perf_counter()
        rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(k))
        dt = time.perf_counter() - t0
        L.mi355x_tab_destroy(h)
        best = max(best, k.value / dt)
    return best
def batch(nl, reps=3):
    n, m = 512, 256
    seeds = np.array([lp.synth.seed_for(4, i) for i in range(nl)], dtype=np.uint64)
    best = 0
    for rep in range(reps):
        b = lp.TableauBatch.synthetic(nl, n, m, seeds)
        lp.capi.check(L.mi355x_batch_prepare(b._h), "prepare")
        t0 = time.perf_counter()
        st, npv = b.solve()
        dt = time.perf_counter() - t0
        best = max(best, npv.sum() / dt)
    return best
for rnd in range(2):
    for mode in (1, 2):
        L.mi355x_tune_set_resident_poll(mode)
        print("poll mode %d: cfg2 %.0f pivots/s   batch128 %.2f M   batch1024 %.2f M" % (mode, cfg2(), batch(128) / 1e6, batch(1024) / 1e6))
L.mi355x_tune_set_resident_poll(0)
