import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, importlib, oracle
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
def one(n, m, seed, cap=0):
    M0, b0 = lp.synth.tableau(n, m, seed)
    M, b = M0.copy(), b0.copy()
    so, no, trace = oracle.solve(M, b, max_pivots=cap, trace_cap=1 << 14)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    k = ctypes.c_int64(0)
    L.mi355x_tab_solve_async(t._h, 1, 1024.0, 0, 1)     # representation change outside the timing
    L.mi355x_tab_sync(t._h, ctypes.byref(k))
    t0 = time.perf_counter()
    rc = L.mi355x_tab_solve(t._h, 1, 1024.0, cap, ctypes.byref(k))
    dt = time.perf_counter() - t0
    res = L.mi355x_tab_resident(t._h)
    t._touch()
    ok = (rc, k.value) == (so, no) and np.array_equal(t.pivot_trace()[:no], trace) and \
        np.array_equal(t.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(t.basis_columns, b)
    print("LP %dx%d cap %d: resident=%d rc=%d pivots=%d (oracle %d %d) ok=%s  %.3f ms -> %.0f pivots/s" % (
        n, m, cap, res, rc, k.value, so, no, ok, dt * 1e3, k.value / dt))
    return ok
allok = True
for (n, m) in [(5, 3), (40, 20), (64, 30), (65, 30), (200, 100), (300, 256), (300, 257), (700, 333), (1024, 512), (500, 700), (2048, 200)]:
    allok &= one(n, m, lp.synth.seed_for(2, n + m))
allok &= one(1024, 512, lp.synth.seed_for(2), cap=37)
# timing on config 2 (second run: clocks up)
for rep in range(3):
    allok &= one(1024, 512, lp.synth.seed_for(2))
# batch, config-4 shape
for nl in (16, 128, 1024):
    n, m = 512, 256
    seeds = np.array([lp.synth.seed_for(4, i) for i in range(nl)], dtype=np.uint64)
    for rep in range(2):
        batch = lp.TableauBatch.synthetic(nl, n, m, seeds)
        lp.capi.check(L.mi355x_batch_prepare(batch._h), "prepare")
        t0 = time.perf_counter()
        st, npv = batch.solve()
        dt = time.perf_counter() - t0
    good = True
    for i in range(min(nl, 24)):
        Mi, bi = lp.synth.tableau(n, m, int(seeds[i]))
        so, no, _ = oracle.solve(Mi, bi)
        Gi, gb = batch.download(i)
        good &= int(st[i]) == so and int(npv[i]) == no and np.array_equal(Gi.view(np.int64), Mi.view(np.int64)) and np.array_equal(gb, bi)
    allok &= good
    print("batch %d LPs: %.3f ms, %d pivots -> %.2f M pivots/s, parity(first 24)=%s" % (nl, dt * 1e3, npv.sum(), npv.sum() / dt / 1e6, good))
print("ALL OK" if allok else "FAILURES")
