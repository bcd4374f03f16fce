"""Batches whose members span hundreds of orders of magnitude (NaN / inf / subnormal corner) through
mi355x_batch_solve, every batch mode; every member against the oracle run on it alone.
    python tools/fuzz_batch_extreme.py [batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1        # run just this batch, in every mode, verbosely
meta = np.random.default_rng(123)
bad = 0
t0 = time.time()
for bi in range(batches):
    n = int(meta.integers(2, 61)); m = int(meta.integers(1, 41)); nl = int(meta.integers(2, 17))
    lo = int(meta.choice([-300, -160, -20])); hi = int(meta.choice([20, 160, 300]))
    mode = int(meta.choice([0, 1, 2, 3]))
    Ms, Bs, ref = [], [], []
    for k in range(nl):
        rng = np.random.default_rng(int(meta.integers(0, 2 ** 31 - 1)))
        mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(lo, hi + 1, shape)   # noqa: E731
        M0 = np.zeros((m + 1, n + m + 1))
        M0[:m, :n] = mag((m, n)) * rng.choice([1.0, 1.0, -1.0], (m, n))
        M0[np.arange(m), n + np.arange(m)] = 1.0
        M0[:m, -1] = mag(m)
        M0[m, :n] = -mag(n)
        b0 = np.arange(n, n + m, dtype=np.int64)
        Ms.append(M0); Bs.append(b0)
        M, b = M0.copy(), b0.copy()
        with np.errstate(all="ignore"):
            st, npiv, _ = oracle.solve(M, b, max_pivots=60)
        ref.append((st, npiv, M, b))
    if only >= 0 and bi != only:
        continue
    if only >= 0:
        for md in (0, 1, 2, 3):
            L.mi355x_tune_set_batch_mode(md)
            bt = lp.TableauBatch.from_arrays(np.stack(Ms), np.stack(Bs))
            st, npv = bt.solve(max_pivots=60)
            print("mode %d: gpu    %s" % (md, list(zip(st.tolist(), npv.tolist()))))
        print("        oracle %s" % [(r[0], r[1]) for r in ref])
        print("non-finite entries in the final oracle tableaux:", [int((~np.isfinite(r[2])).sum()) for r in ref])
    L.mi355x_tune_set_batch_mode(mode)
    batch = lp.TableauBatch.from_arrays(np.stack(Ms), np.stack(Bs))
    st, npv = batch.solve(max_pivots=60)
    for k in range(nl):
        G, bg = batch.download(k)
        so, no, M, b = ref[k]
        nan_o, nan_g = np.isnan(M), np.isnan(G)
        ok = (int(st[k]), int(npv[k])) == (so, no) and np.array_equal(nan_o, nan_g) and \
            np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64)) and np.array_equal(bg, b)
        if not ok:
            bad += 1
            print("MISMATCH batch %d (mode %d, %d x %d, %d LPs, [%d,%d]) member %d: status %d/%d pivots %d/%d" % (
                bi, mode, n, m, nl, lo, hi, k, st[k], so, npv[k], no), flush=True)
    batch.close() if hasattr(batch, "close") else None
L.mi355x_tune_set_batch_mode(0)
print("%d batches, %d mismatching members, %.0f s" % (batches, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
