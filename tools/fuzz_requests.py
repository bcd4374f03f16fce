"""Random request sequences: mi355x_tab_solve_async(n_1), (n_2), ... with arbitrary n_i (blocks, partial
blocks, remainders, single pivots) on random LPs, checked against the oracle stopped at the same
total (status, pivot count, trace, tableau bits).
    python tools/fuzz_requests.py [cases]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
meta = np.random.default_rng(2024)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


bad = 0
t0 = time.time()
for case in range(cases):
    n = int(meta.integers(2, 500)); m = int(meta.integers(1, 300)); seed = int(meta.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    is_max = int(meta.integers(0, 2))
    if meta.integers(0, 3) == 0:
        A = rng.integers(0, 4, (m, n)).astype(np.float64); bb = rng.integers(0, 5, m).astype(np.float64)
        c = rng.integers(-2, 5, n).astype(np.float64)
    else:
        A = rng.uniform(-0.5, 1.5, (m, n)); bb = rng.uniform(0.5, 5.0, m); c = rng.uniform(-0.5, 2.0, n)
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = A
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = bb
    M0[m, :n] = -c if is_max else c
    b0 = np.arange(n, n + m, dtype=np.int64)
    reqs = [int(x) for x in meta.choice([1, 2, 3, 5, 7, 15, 16, 17, 20, 23, 24, 25, 27, 28, 29, 31, 32, 33, 40, 48, 56, 64, 100], size=int(meta.integers(1, 6)))]
    total = sum(reqs)
    M, b = M0.copy(), b0.copy()
    st_o, npiv, trace = oracle.solve(M, b, is_max=bool(is_max), max_pivots=total, trace_cap=total)
    if npiv == total:
        st_o = 100        # exactly the requested pivots were made: the device has not looked at the tableau again
    L.mi355x_tune_set_lookahead_mode(int(meta.choice([0, 0, 1]))); L.mi355x_tune_set_block(int(meta.choice([0, 0, 16, 8, 1, 24, 28])))
    # (round 6) the wide ring pass with tiles of unequal heights: an explicit skew of the thirds on whatever grid
    # the tile height gives (mi355x_tune_set_sweep_skew; -1 = the library's own rule)
    L.mi355x_tune_set_sweep_skew(int(meta.choice([-1, 0, 4, 8, 12, 20]))); L.mi355x_tune_set_sweep_shape(int(meta.choice([0, 0, 8, 16, 32])), -1)
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create(ctypes.byref(h), m + 1, n + m + 1, ptr(M0), ptr(b0), 0), "create")
    k = ctypes.c_int64(0)
    rc = None
    for i, r in enumerate(reqs):
        lp.capi.check(L.mi355x_tab_solve_async(h, is_max, 1024.0, r, 1 if i == 0 else 0), "solve_async")
        rc = L.mi355x_tab_sync(h, ctypes.byref(k))
    ec = np.empty(total + 4, dtype=np.int64); cr = np.empty(total + 4, dtype=np.int64); nn = ctypes.c_int64(0)
    L.mi355x_tab_trace(h, ptr(ec), ptr(cr), total + 4, ctypes.byref(nn))
    got = np.stack([ec[:nn.value], cr[:nn.value]], axis=1)
    G = np.empty_like(M0); bg = np.empty_like(b0)
    lp.capi.check(L.mi355x_tab_download(h, ptr(G), ptr(bg), None, None), "download")
    L.mi355x_tab_destroy(h)
    ok = (rc, k.value) == (st_o, npiv) and np.array_equal(got, trace) and np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)
    if not ok:
        bad += 1
        print("MISMATCH case %d: %d x %d seed %d max=%d requests %s: rc %d/%d pivots %d/%d" % (case, n, m, seed, is_max, reqs, rc, st_o, k.value, npiv), flush=True)
        if bad >= 10:
            break
L.mi355x_tune_set_lookahead_mode(0); L.mi355x_tune_set_block(0)
print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
