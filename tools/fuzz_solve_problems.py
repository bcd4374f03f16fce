"""Random LISTS of problems through the list entry (mi355x-solve-problems: same-shape groups as
multi-device batches, two-phase groups as pairs of batches, the rest alone) against the one-problem
hook, member by member: same condition or bit-identical tableau, basis and pivot counts.  (Each
path is pinned against the oracle by the tests; this checks that grouping, chunking and the batched
hand-over change nothing.)
    python tools/fuzz_solve_problems.py [lists] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests.helpers import lp_amd, random_mixed_problem
lp = lp_amd()
lists = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def le_problem(rng, n, m, kind, degenerate):
    names = ["v%d" % i for i in range(n)]
    if degenerate:
        A = rng.integers(0, 4, (m, n)).astype(float); b = rng.integers(0, 5, m).astype(float)
        c = rng.integers(-2, 5, n).astype(float)
    else:
        A = rng.uniform(0.1, 1.0, (m, n)); b = rng.uniform(3, 9, m); c = rng.uniform(0.5, 1.5, n)
    if kind == "min":
        c = -c
    cons = [("<=", list(zip(names, A[i].tolist())), float(b[i])) for i in range(m)]
    return lp.Problem(type=kind, vars=names, objective_var="obj", objective_func=list(zip(names, c.tolist())), constraints=cons)


def drive_out_problem(rng):
    n = 5
    names = ["x%d" % i for i in range(n)]
    rows = [rng.integers(-2, 3, n).astype(float) for _ in range(3)]
    cons = [("=", list(zip(names, a.tolist())), 0.0) for a in rows if a.any()]
    cons.append(("<=", list(zip(names, [1.0] * n)), 5.0))
    return lp.Problem(type="max", vars=names, objective_var="obj",
                      objective_func=list(zip(names, rng.integers(1, 4, n).astype(float).tolist())), constraints=cons)


bad, members = 0, 0
t0 = time.time()
for li in range(lists):
    rng = np.random.default_rng(seed0 + li)
    shapes = [(int(rng.integers(3, 40)), int(rng.integers(2, 25))) for _ in range(int(rng.integers(1, 4)))]
    mixed = [(int(rng.integers(4, 16)), int(rng.integers(1, 5)), int(rng.integers(0, 4)), int(rng.integers(0, 3))) for _ in range(int(rng.integers(1, 3)))]
    ps = []
    for _ in range(int(rng.integers(4, 28))):
        r = rng.integers(0, 10)
        if r < 5:
            n, m = shapes[int(rng.integers(0, len(shapes)))]
            ps.append(le_problem(rng, n, m, "max" if rng.integers(0, 3) else "min", bool(rng.integers(0, 4) == 0)))
        elif r < 9:
            n, a, b, c = mixed[int(rng.integers(0, len(mixed)))]
            if b + c == 0:
                b = 1
            ps.append(random_mixed_problem(lp, n, a, b, c, int(rng.integers(0, 2 ** 31 - 1)), kind="max" if rng.integers(0, 3) else "min"))
        else:
            ps.append(drive_out_problem(rng))
    devices = int(rng.integers(1, 5))
    got = lp.solve_problems(ps, devices=devices, errorp=False)
    for k, (p, r) in enumerate(zip(ps, got)):
        members += 1
        try:
            one = lp.solve_problem(p, native=False)
        except Exception as e:                            # noqa: BLE001  (the reference's conditions)
            one = e
        if isinstance(one, Exception):
            ok = type(r) is type(one)
        else:
            ok = isinstance(r, lp.Tableau) and np.array_equal(r.matrix.view(np.int64), one.matrix.view(np.int64)) and \
                np.array_equal(r.basis_columns, one.basis_columns) and (getattr(r, 'n_pivots', None) is None or getattr(one, 'n_pivots', None) is None or
                 np.array_equal(np.atleast_1d(r.n_pivots), np.atleast_1d(one.n_pivots)))
        if not ok:
            bad += 1
            print("MISMATCH list %d (seed %d) member %d of %d, devices %d: list entry %r, one-problem hook %r"
                  % (li, seed0 + li, k, len(ps), devices, type(r).__name__, type(one).__name__), flush=True)
    if bad >= 10:
        break
print("%d lists, %d members, %d mismatches, %.0f s" % (li + 1, members, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
