"""Two-phase solves (mi355x_solve_two_phase, src/simplex.lisp:402-452) on random mixed problems
(<=, >= and = constraints, max and min), many more than tests/test_gpu_property.py runs.
    python tools/fuzz_two_phase.py [cases]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd, random_mixed_problem
lp = lp_amd(); L = lp.capi.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
meta = np.random.default_rng(31)
bad = 0
t0 = time.time()
for case in range(cases):
    n = int(meta.integers(2, 120)); mle = int(meta.integers(0, 40)); mge = int(meta.integers(0, 30)); meq = int(meta.integers(0, 20))
    seed = int(meta.integers(0, 2 ** 31 - 1)); kind = str(meta.choice(["max", "min"]))
    if mge + meq == 0:
        mge = 1
    problem = random_mixed_problem(lp, n, mle, mge, meq, seed, kind=kind)
    art, main = lp.build_tableau(problem, problem)
    A, ab = art.matrix.copy(), art.basis_columns.copy()
    Mm, mb = main.matrix.copy(), main.basis_columns.copy()
    st_o, npv = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max)
    npiv = (ctypes.c_int64 * 2)()
    L.mi355x_tune_set_lookahead_mode(int(meta.choice([0, 0, 1]))); L.mi355x_tune_set_block(int(meta.choice([0, 16, 4, 1])))
    rc = L.mi355x_solve_two_phase(art._h, main._h, int(main.is_max), 1024.0, npiv)
    art._touch(); main._touch()
    ok = rc == st_o and np.array_equal(art.matrix.view(np.int64), A.view(np.int64))
    if st_o == oracle.OPTIMAL:
        ok = ok and (npiv[0], npiv[1]) == (npv[0], npv[1])
    if st_o in (oracle.OPTIMAL, oracle.UNBOUNDED):
        ok = ok and np.array_equal(main.matrix.view(np.int64), Mm.view(np.int64)) and np.array_equal(main.basis_columns, mb)
    if not ok:
        bad += 1
        print("MISMATCH case %d: n=%d le=%d ge=%d eq=%d seed=%d %s: rc %d/%d pivots (%d,%d)/(%d,%d)" % (
            case, n, mle, mge, meq, seed, kind, rc, st_o, npiv[0], npiv[1], npv[0], npv[1]), flush=True)
        if bad >= 10:
            break
L.mi355x_tune_set_lookahead_mode(0); L.mi355x_tune_set_block(0)
print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
