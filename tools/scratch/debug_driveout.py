import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, importlib, oracle
lp = importlib.import_module("linear-programming_amd")
cp = importlib.import_module("linear-programming_amd.colpart")
np.set_printoptions(linewidth=250, precision=4)
shown = 0
for seed in range(40):
    rng = np.random.default_rng(seed)
    n = 6
    names = ["x%d" % i for i in range(n)]
    x0 = rng.integers(0, 3, n).astype(float) * (rng.uniform(size=n) < 0.5)
    rows = [rng.integers(0, 3, n).astype(float) for _ in range(4)]
    cons = [("=", list(zip(names, a.tolist())), float(a @ x0)) for a in rows if a.any()]
    cons.append(("<=", list(zip(names, [1.0] * n)), float(x0.sum() + 3)))
    problem = lp.Problem(type="max", vars=names, objective_var="obj",
                         objective_func=list(zip(names, rng.integers(1, 4, n).astype(float).tolist())), constraints=cons)
    tabs = lp.build_tableau(problem, problem)
    if not isinstance(tabs, list):
        continue
    art, main = tabs
    A, ab = art.matrix.copy(), art.basis_columns.copy()
    Mm, mb = main.matrix.copy(), main.basis_columns.copy()
    st, npv = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max)
    A1, b1 = art.matrix.copy(), art.basis_columns.copy()
    _, n_plain, _ = oracle.solve(A1, b1, is_max=False)
    for shards in (1, 2, 3):
        tab = cp.NativeColumnPartition.from_arrays(art.matrix.copy(), art.basis_columns.copy(), shards)
        rc, got, mt = tab.solve_two_phase(main.matrix[-1].copy(), main.is_max, 1024)
        G, gb, _, _ = tab.download()
        ok = rc == st and np.array_equal(G.view(np.int64), A.view(np.int64)) and np.array_equal(gb, ab)
        okm = None
        if mt is not None:
            GM, gm, _, _ = mt.download()
            okm = np.array_equal(GM.view(np.int64), Mm.view(np.int64)) and np.array_equal(gm, mb)
            mt.close()
        tab.close()
        if not ok or okm is False:
            print("seed", seed, "shards", shards, "rc", rc, "oracle", st, "npv", got, npv.tolist(), "plain phase-1 pivots", n_plain, "art ok", ok, "main ok", okm)
            if shown < 2:
                shown += 1
                print("basis got", gb, "oracle", ab)
                d = np.argwhere(G.view(np.int64) != A.view(np.int64))
                print("art diffs at", d[:20].tolist())
                for r, c in d[:6]:
                    print("   ", r, c, G[r, c], A[r, c])
                print("oracle art:\n", A); print("got art:\n", G)
