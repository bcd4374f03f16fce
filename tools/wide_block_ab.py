"""Pivots per sweep on the big tableaux: config 5 as ONE column shard (17.2 GB stored) and as a single
tableau, per block size -- full blocks only, timed between solve_async and sync.
    python tools/wide_block_ab.py [blocks per measurement]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
lp = importlib.import_module("linear-programming_amd")
cp = importlib.import_module("linear-programming_amd.colpart")
L = lp.capi.lib()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, m = 65536, 32768
seed = lp.synth.seed_for(5)
for bk in (16, 24, 28, 0):
    got = L.mi355x_tune_set_block(bk)
    eff = got or 28
    tab = cp.NativeColumnPartition.synthetic(n, m, seed, 1)
    tab.solve_async(2 * eff, reset=True); tab.sync()
    t0 = time.perf_counter()
    tab.solve_async(NB * eff); st, done = tab.sync()
    dt = time.perf_counter() - t0
    print("one column shard, block %2d (knob %2d): %7.1f us per pivot = %6.0f pivots/s (%d pivots, status %d)"
          % (eff, got, dt / (NB * eff) * 1e6, NB * eff / dt, NB * eff, st), flush=True)
    tab.close()
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, 0), "create")
    k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 2 * eff, 1), "warm"); L.mi355x_tab_sync(h, ctypes.byref(k))
    t0 = time.perf_counter()
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, NB * eff, 0), "run"); st = L.mi355x_tab_sync(h, ctypes.byref(k))
    dt = time.perf_counter() - t0
    print("single tableau,   block %2d (knob %2d): %7.1f us per pivot = %6.0f pivots/s (block size reported %d, status %d)"
          % (eff, got, dt / (NB * eff) * 1e6, NB * eff / dt, L.mi355x_tab_block_size(h), st), flush=True)
    L.mi355x_tab_destroy(h)
L.mi355x_tune_set_block(0)
