"""Column partition behind the C ABI (logical shards on one GPU): random LPs -- sparse, dense,
integer-degenerate (ties broken by lowest index ACROSS shards), max and min -- split over 1..8
shards, against the oracle.  Max problems only through mi355x_colpart_solve's is_max=1 ... both.
    python tools/fuzz_colpart.py [cases]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
cp = importlib.import_module("linear-programming_amd.colpart")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
meta = np.random.default_rng(99)
bad = 0
t0 = time.time()
for case in range(cases):
    n = int(meta.integers(8, 400)); m = int(meta.integers(1, 250)); seed = int(meta.integers(0, 2 ** 31 - 1))
    nd = int(meta.integers(1, 9))
    rng = np.random.default_rng(seed)
    is_max = int(meta.integers(0, 2))
    if meta.integers(0, 2) == 0:
        A = rng.integers(0, 4, (m, n)).astype(np.float64); bb = rng.integers(0, 5, m).astype(np.float64)
        c = rng.integers(-2, 5, n).astype(np.float64)
    else:
        A = rng.uniform(-0.5, 1.5, (m, n)); A[rng.uniform(size=(m, n)) > float(meta.choice([1.0, 0.5, 0.1]))] = 0.0
        bb = rng.uniform(0.5, 5.0, m); c = rng.uniform(-0.5, 2.0, n)
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = A
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = bb
    M0[m, :n] = -c if is_max else c
    b0 = np.arange(n, n + m, dtype=np.int64)
    cap = 200
    M, b = M0.copy(), b0.copy()
    st_o, npiv, trace = oracle.solve(M, b, is_max=bool(is_max), max_pivots=cap, trace_cap=cap)
    blk, exch, split = int(meta.choice([0, 0, 16, 24, 28])), int(meta.choice([0, 0, 2, 3])), int(meta.choice([0, 0, 2]))
    L.mi355x_tune_set_block(blk); L.mi355x_tune_set_colpart_exchange(exch); L.mi355x_tune_set_shard_la_split(split)
    tab = cp.NativeColumnPartition.from_arrays(M0, b0, nd)
    st, k = tab.solve(is_max=bool(is_max), max_pivots=cap)
    L.mi355x_tune_set_block(0); L.mi355x_tune_set_colpart_exchange(0); L.mi355x_tune_set_shard_la_split(0)
    got = tab.trace(npiv) if k == npiv else None
    G, bg, _, _ = tab.download()
    tab.close()
    ok = (st, k) == (st_o, npiv) and got is not None and np.array_equal(got, trace) and \
        np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)
    if not ok:
        bad += 1
        print("MISMATCH case %d: %d x %d seed %d max=%d shards %d block %d exchange %d split %d: status %d/%d pivots %d/%d" % (case, n, m, seed, is_max, nd, blk, exch, split, st, st_o, k, npiv), flush=True)
        if bad >= 10:
            break
print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
