"""Column partition on LPs that overflow (entries over hundreds of orders of magnitude): a compact
shard cannot reproduce what the reference does with a non-finite entering column or a NaN
quotient and says MI_NONFINITE instead (include/mi355x_simplex.h).  Every run must therefore end
EITHER exactly like the oracle (status, pivots, trace, bits) OR with MI_NONFINITE after a prefix of
the oracle's pivots -- never with anything else, never hang.
    python tools/fuzz_colpart_extreme.py [cases]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
cp = importlib.import_module("linear-programming_amd.colpart")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
meta = np.random.default_rng(4242)
bad = nonfinite = 0
t0 = time.time()
for case in range(cases):
    n = int(meta.integers(8, 61)); m = int(meta.integers(1, 41)); nd = int(meta.integers(1, 9))
    lo = int(meta.choice([-300, -160, -20])); hi = int(meta.choice([20, 160, 300]))
    rng = np.random.default_rng(int(meta.integers(0, 2 ** 31 - 1)))
    mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(lo, hi + 1, shape)   # noqa: E731
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = mag((m, n)) * rng.choice([1.0, 1.0, -1.0], (m, n))
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = mag(m)
    M0[m, :n] = -mag(n)
    b0 = np.arange(n, n + m, dtype=np.int64)
    M, b = M0.copy(), b0.copy()
    with np.errstate(all="ignore"):
        st_o, npiv, trace = oracle.solve(M, b, max_pivots=60, trace_cap=60)
    tab = cp.NativeColumnPartition.from_arrays(M0, b0, nd)
    st, k = tab.solve(max_pivots=60)
    got = tab.trace(k) if k > 0 else np.zeros((0, 2), dtype=np.int64)
    if st == lp.capi.MI_NONFINITE:
        nonfinite += 1
        ok = k <= npiv and np.array_equal(got, trace[:k])
    else:
        G, bg, _, _ = tab.download()
        nan_o, nan_g = np.isnan(M), np.isnan(G)
        ok = (st, k) == (st_o, npiv) and np.array_equal(got, trace) and np.array_equal(nan_o, nan_g) and \
            np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64)) and np.array_equal(bg, b)
    tab.close()
    if not ok:
        bad += 1
        print("MISMATCH case %d: %d x %d shards %d [%d,%d]: status %d/%d pivots %d/%d" % (case, n, m, nd, lo, hi, st, st_o, k, npiv), flush=True)
        if bad >= 10:
            break
print("%d cases, %d mismatches, %d ended with MI_NONFINITE, %.0f s" % (case + 1, bad, nonfinite, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
