import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, importlib, oracle
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
n, m = 1500, 700
seed = lp.synth.seed_for(2, 77)
M0, b0 = lp.synth.tableau(n, m, seed)
for fault in (-16, -15, -9):
    M, b = M0.copy(), b0.copy()
    st_o, npiv, trace = oracle.solve(M, b, max_pivots=200, trace_cap=200)
    L.mi355x_tune_set_la_max_spins(20000)
    L.mi355x_tune_set_la_fault(fault)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    k = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(t._h, 1, 1024.0, 200, ctypes.byref(k))
    t._touch()
    L.mi355x_tune_set_la_max_spins(0)
    L.mi355x_tune_set_la_fault(0)
    got = t.pivot_trace()
    d = np.where((got != trace).any(axis=1))[0]
    print("fault", fault, "rc", rc, k.value, "lost", L.mi355x_tab_la_lost(t._h), "first diffs", d[:6], got[d[:4]].tolist(), trace[d[:4]].tolist())
    print("   matrix equal", np.array_equal(t.matrix.view(np.int64), M.view(np.int64)))
