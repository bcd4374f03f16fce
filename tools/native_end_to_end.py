"""End-to-end timing of the native path (problem -> build-tableau -> upload -> solve -> light
solution) on a config-3-sized LP, to see what the host-side assembly costs next to the solve."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
_a = [a for a in sys.argv[1:] if not a.startswith('--')]
n, m = (8192, 4096) if len(_a) < 2 else (int(_a[0]), int(_a[1]))
A, b, c = lp.synth.lp_data(n, m, lp.synth.seed_for(3))
idx = np.arange(n, dtype=np.int64)
if "--init" in sys.argv:                      # the one-off costs out of the first solve (what a host does at load time)
    t0 = time.perf_counter()
    lp.capi.check(L.mi355x_init(0), "init")
    print("mi355x_init %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    p = ctypes.c_void_p()
    lp.capi.check(L.mi355x_problem_create(ctypes.byref(p), 1, n), "create")
    lp.capi.check(L.mi355x_problem_set_objective(p, idx.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p), n), "obj")
    for i in range(m):
        row = np.ascontiguousarray(A[i])
        lp.capi.check(L.mi355x_problem_add_constraint(p, 0, idx.ctypes.data_as(ctypes.c_void_p), row.ctypes.data_as(ctypes.c_void_p), n, float(b[i])), "row")
    t1 = time.perf_counter()
    s = ctypes.c_void_p()
    rc = L.mi355x_simplex_solver(p, 1024.0, 0, ctypes.byref(s))
    t2 = time.perf_counter()
    w = ctypes.c_double(0); p1 = ctypes.c_int64(0); p2 = ctypes.c_int64(0)
    L.mi355x_solution_objective_value(s, ctypes.byref(w)); L.mi355x_solution_pivots(s, ctypes.byref(p1), ctypes.byref(p2))
    print("rep %d: problem construction %.0f ms, mi355x_simplex_solver %.0f ms (rc %d, %d pivots, objective %.6f)"
          % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rc, p2.value, w.value), flush=True)
    L.mi355x_solution_destroy(s); L.mi355x_problem_destroy(p)
