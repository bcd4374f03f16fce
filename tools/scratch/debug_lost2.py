import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, importlib, oracle
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
n, m = 1500, 700
seed = lp.synth.seed_for(2, 77)
M0, b0 = lp.synth.tableau(n, m, seed)
for fault, first in ((-16, 16), (-16, 32), (-16, 48), (-15, 32)):
    M, b = M0.copy(), b0.copy()
    st_o, npiv, trace = oracle.solve(M, b, max_pivots=64, trace_cap=64)
    L.mi355x_tune_set_la_max_spins(20000)
    L.mi355x_tune_set_la_fault(fault)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    k = ctypes.c_int64(0)
    L.mi355x_tab_solve_async(t._h, 1, 1024.0, first, 1)
    rc = L.mi355x_tab_sync(t._h, ctypes.byref(k))
    print("fault", fault, "first", first, "-> rc", rc, "pivots", k.value, "lost", L.mi355x_tab_la_lost(t._h))
    L.mi355x_tab_solve_async(t._h, 1, 1024.0, 64 - k.value, 0)
    rc = L.mi355x_tab_sync(t._h, ctypes.byref(k))
    t._touch()
    L.mi355x_tune_set_la_max_spins(0)
    L.mi355x_tune_set_la_fault(0)
    got = t.pivot_trace()[:64]
    d = np.where((got != trace).any(axis=1))[0]
    print("   rc", rc, k.value, "first diffs", d[:6], got[d[:3]].tolist(), trace[d[:3]].tolist(),
          "matrix equal", np.array_equal(t.matrix.view(np.int64), M.view(np.int64)))
