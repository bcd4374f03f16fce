"""Quick bit-for-bit check of the persistent block launch of column shards (k_shard_la_block) on small
shapes: 1 / 2 / 3 / 8 logical shards in exchange mode 2, compact and dense shards, against the oracle.
    python tools/shard_block_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib
lp = importlib.import_module("linear-programming_amd")
cp = importlib.import_module("linear-programming_amd.colpart")
import oracle  # noqa: E402  (test infrastructure: the checker)
L = lp.capi.lib()
bad = 0
for (n, m, seed) in ((96, 64, 1), (700, 333, 2), (1500, 700, 3), (3000, 1200, 4)):
    for shards in (1, 2, 3, 8):
        for dense in (False, True):
            M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(5, seed))
            if dense:
                M0[:m, n:n + m] *= 2.0                       # basis columns != unit columns: dense shards
            M, b = M0.copy(), b0.copy()
            so, no, trace = oracle.solve(M, b, trace_cap=1 << 16)
            L.mi355x_tune_set_colpart_exchange(2)
            tab = cp.NativeColumnPartition.from_arrays(M0, b0, shards)
            L.mi355x_tune_set_colpart_exchange(0)
            st, k = tab.solve(max_pivots=23)
            st, k2 = tab.solve()
            G, bg, _, _ = tab.download()
            ok = (st, k + k2) == (so, no) and np.array_equal(tab.trace(no), trace) and \
                np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)
            stats = tab.la_stats()
            print("n=%d m=%d shards=%d %s: %s  pivots %d/%d status %d/%d  block %d  %s" % (
                n, m, shards, "dense" if dense else "compact", "OK" if ok else "MISMATCH", k + k2, no, st, so,
                tab.block_size(), stats), flush=True)
            bad += 0 if ok else 1
            tab.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
