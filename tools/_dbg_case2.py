import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, importlib, ctypes
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
cp = importlib.import_module("linear-programming_amd.colpart")
seed = 1
rng = np.random.default_rng(1000 + seed)
n = 5
names = ["x%d" % i for i in range(n)]
x0 = rng.integers(1, 4, n).astype(float)
rows = [rng.integers(0, 3, n).astype(float) for _ in range(2)]
cons = [("=", list(zip(names, a.tolist())), float(a @ x0)) for a in rows if a.any()]
cons.append(("<=", list(zip(names, [1.0] * n)), float(x0.sum() + 2)))
cons.append((">=", [(names[0], 1.0)], 1.0))
c = rng.integers(1, 4, n).astype(float)
c[int(rng.integers(0, n))] = np.inf if seed % 2 else -np.inf
problem = lp.Problem(type="max", vars=names, objective_var="obj", objective_func=list(zip(names, c.tolist())), constraints=cons)
tabs = lp.build_tableau(problem, problem)
art, main = tabs
A, ab = art.matrix.copy(), art.basis_columns.copy()
Mm, mb = main.matrix.copy(), main.basis_columns.copy()
with np.errstate(all="ignore"):
    st, npv = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max, factor=main.fp_tolerance_factor)
print("oracle", st, npv, "main obj row", main.matrix[-1])
print("oracle final obj row", Mm[-1]); print("oracle basis", mb)
for la_off in (0, 1):
    for blk in (0, 1):
        L.mi355x_tune_set_shard_la_block(la_off); L.mi355x_tune_set_block(blk)
        tab = cp.NativeColumnPartition.from_arrays(art.matrix.copy(), art.basis_columns.copy(), 1)
        rc, got, mt = tab.solve_two_phase(main.matrix[-1].copy(), main.is_max, main.fp_tolerance_factor)
        print("la_off", la_off, "block", blk, "rc", rc, got)
        if mt is not None:
            G, gb, lr, lc = mt.download(); print(" obj row", lr, "basis", gb, "trace", mt.trace(max(got[1],1)))
            mt.close()
        tab.close()
