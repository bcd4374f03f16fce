"""Fuzz of the two-phase branch on the column partition (mi355x_colpart_solve_two_phase): random
LPs with <=, >= and = rows (degenerate integer data in half of them, so that artificials stay basic
at level zero and drive-out pivots happen), 1 .. 8 logical shards, exchange modes 0 (device-local
sum / all-reduce) and 2 (P2P push), against the oracle bit for bit.  A drive-out pivot on a negative
element must be DECLINED (MI_UNSUPPORTED) and nothing else may be.
    python tools/fuzz_colpart_two_phase.py [cases] [first_seed] [two-launch]
`two-launch`: every case as ONE shard over the forced one-rank loop in exchange mode 2 with the
multi-workgroup look-ahead step forced: the two-launch step (k_shard_p2p_step) at these sizes."""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd, random_mixed_problem
lp = lp_amd(); L = lp.capi.lib()
cp = importlib.import_module("linear-programming_amd.colpart")
TWO_LAUNCH = len(sys.argv) > 3 and sys.argv[3] == "two-launch"
if TWO_LAUNCH:
    os.environ["MI355X_COLPART_FORCE_RCCL"] = "1"
    L.mi355x_tune_set_shard_la_split(2)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
meta = np.random.default_rng(seed0)
bad = declined = drove = 0
t0 = time.time()
for case in range(cases):
    seed = int(meta.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    shards = int(meta.integers(1, 9))
    mode = int(meta.choice([0, 2]))
    if TWO_LAUNCH:                      # one shard over a forced one-rank loop: the two-launch P2P step at any size
        shards, mode = 1, 2
    if meta.integers(0, 2):
        n = int(meta.integers(3, 9))
        names = ["x%d" % i for i in range(n)]
        x0 = rng.integers(0, 3, n).astype(float) * (rng.uniform(size=n) < 0.5)
        rows = [rng.integers(-1, 3, n).astype(float) for _ in range(int(meta.integers(2, 6)))]
        cons = []
        for a in rows:
            if not a.any():
                continue
            rhs = float(a @ x0)
            if rhs < 0:
                a, rhs = -a, -rhs
            cons.append((str(meta.choice(["=", "=", ">="])), list(zip(names, a.tolist())), rhs))
        cons.append(("<=", list(zip(names, [1.0] * n)), float(x0.sum() + 3)))
        problem = lp.Problem(type=str(meta.choice(["max", "min"])), vars=names, objective_var="obj",
                             objective_func=list(zip(names, rng.integers(1, 4, n).astype(float).tolist())), constraints=cons)
    else:
        problem = random_mixed_problem(lp, int(meta.integers(2, 80)), int(meta.integers(0, 25)), int(meta.integers(0, 20)),
                                       int(meta.integers(0, 12)), seed, kind=str(meta.choice(["max", "min"])))
    try:
        tabs = lp.build_tableau(problem, problem)
    except lp.SolverError:                                   # (a draw without constraints: build-tableau's own special case, :153-186)
        continue
    if not isinstance(tabs, list):
        continue
    art, main = tabs
    A, ab = art.matrix.copy(), art.basis_columns.copy()
    Mm, mb = main.matrix.copy(), main.basis_columns.copy()
    st, npv = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max)
    A1, b1 = art.matrix.copy(), art.basis_columns.copy()
    _, n_plain, _ = oracle.solve(A1, b1, is_max=False)
    did_drive = st not in (oracle.INFEASIBLE,) and int(npv[0]) > n_plain
    drove += did_drive
    L.mi355x_tune_set_colpart_exchange(mode)
    try:
        tab = cp.NativeColumnPartition.from_arrays(art.matrix.copy(), art.basis_columns.copy(), shards)
    finally:
        L.mi355x_tune_set_colpart_exchange(0)
    ok, why = True, ""
    try:
        rc, got, mt = tab.solve_two_phase(main.matrix[-1].copy(), main.is_max, 1024)
        GA, ga, _, _ = tab.download()
        ok = rc == st and np.array_equal(GA.view(np.int64), A.view(np.int64)) and np.array_equal(ga, ab)
        if not ok:
            why = "status/art"
        if ok and st in (oracle.OPTIMAL, oracle.UNBOUNDED):
            GM, gm, _, _ = mt.download()
            ok = got[0] == int(npv[0]) and (st != oracle.OPTIMAL or got[1] == int(npv[1])) and \
                np.array_equal(GM.view(np.int64), Mm.view(np.int64)) and np.array_equal(gm, mb)
            why = "main"
        if mt is not None:
            mt.close()
    except lp.capi.Mi355xError as e:
        declined += 1
        ok = e.code == lp.capi.MI_UNSUPPORTED and "negative element" in str(e)
        why = "declined: " + str(e)
    tab.close()
    if not ok:
        bad += 1
        print("MISMATCH case %d seed %d shards %d mode %d (%s): oracle status %d pivots %s" % (case, seed, shards, mode, why, st, npv.tolist()), flush=True)
        if bad >= 10:
            break
print("%d cases, %d mismatches, %d declined (negative drive-out pivot), %d with drive-out pivots, %.0f s" % (
    case + 1, bad, declined, drove, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
