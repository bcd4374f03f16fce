"""The optimum pinned at full size (round-3 review, item 4).  north_star's claim is about FINAL
values ("objective / variable values matching the reference to 1e-10"): rounds 1-3 pinned the
optimum bit for bit for configs 2 and 4 (128 LPs) only.  Here:

  * config 3 (8192 x 4096): the WHOLE solve -- every pivot of the trace, the basis and every entry
    of the final 4097 x 12289 tableau -- bit for bit against the OpenMP oracle (~6 000 pivots);
  * config 5 (65536 x 32768, 25.8 GB): solved to optimality on one GPU -- a few thousand blocks of
    the reference's loop (src/simplex.lisp:453-461) at 17 GB -- with the read-back properties the
    reference's accessors expose (src/simplex.lisp:74-120): dual-feasible objective row, primal-
    feasible RHS, c'x recomputed from the generator within 1e-10 of the tableau's objective, Ax <= b
    with A regenerated row chunk by row chunk (splitmix64 is counter-based);
  * config 4 as BASELINE specifies it: 1024 LPs of 512 x 256 behind ONE multi-device handle as 8
    sub-batches (logical ones on this GPU), EVERY LP's status, pivot count and final tableau
    against the oracle.
"""
import ctypes
import json
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
from tests.helpers import ROOT, lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _synthetic_tableau(n, m, seed):
    h = ctypes.c_void_p()
    lp.capi.check(lp.capi.lib().mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, 0),
                  "create_synthetic")
    return lp.Tableau(None, lp.Problem(type="max"), None, None, n + m, m, {}, _handle=h)


def _u01_at(seed, pos):
    """splitmix64 is counter-based: the u01 values at arbitrary stream positions (synth.splitmix_u01
    for a position array instead of a range)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (pos.astype(np.uint64) + np.uint64(1)) * lp.synth.GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def _note(name, rec):
    """Figures worth keeping (pivot counts, wall times) go to gpurun_out/ next to the test log."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "optimum_%s.json" % name), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError:
        pass


@pytest.mark.slow
@pytest.mark.timeout(1500, method="thread")
def test_config3_full_solve_bitwise_vs_the_oracle():
    n, m = 8192, 4096
    seed = lp.synth.seed_for(3)
    t = _synthetic_tableau(n, m, seed)
    t0 = time.perf_counter()
    lp.n_solve_tableau(t)                                        # raises unless OPTIMAL
    gpu_s = time.perf_counter() - t0
    npiv = t.n_pivots
    M, b = lp.synth.tableau(n, m, seed)
    t0 = time.perf_counter()
    st, no, trace = oracle.solve(M, b, trace_cap=1 << 16, omp=True)
    cpu_s = time.perf_counter() - t0
    _note("cfg3", {"pivots": int(npiv), "gpu_solve_s": gpu_s, "oracle_omp_s": cpu_s, "oracle_threads": oracle.omp_threads()})
    assert st == oracle.OPTIMAL and no == npiv and no > 4000
    got = t.pivot_trace()
    assert got.shape == trace.shape
    bad = np.where((got != trace).any(axis=1))[0]
    assert not len(bad), "first differing pivots %s: got %s, oracle %s" % (bad[:4], got[bad[:4]], trace[bad[:4]])
    assert np.array_equal(t.basis_columns, b)
    G = t.matrix
    assert np.array_equal(G.view(np.int64), M.view(np.int64))                    # every entry of 50.3 M
    # ... and what the reference's accessors read off it (src/simplex.lisp:74-120)
    assert lp.tableau_objective_value(t) == M[m, -1]


@pytest.mark.slow
@pytest.mark.timeout(2400, method="thread")
def test_config5_solved_to_optimality_on_one_gpu():
    L = lp.capi.lib()
    n, m = 65536, 32768
    vc = n + m
    seed = lp.synth.seed_for(5)
    t = _synthetic_tableau(n, m, seed)
    k = ctypes.c_int64(0)
    t0 = time.perf_counter()
    rc = L.mi355x_tab_solve(t._h, 1, 1024.0, 0, ctypes.byref(k))
    wall = time.perf_counter() - t0
    t._touch()
    npiv = int(k.value)
    _note("cfg5", {"status": int(rc), "pivots": npiv, "wall_s": wall, "pivots_per_s": npiv / wall,
                   "block": int(L.mi355x_tab_block_size(t._h))})
    assert rc == lp.capi.MI_OPTIMAL and npiv > 10000
    obj_row, rhs_col, basis = t._readback()                      # objective row, RHS column, basis only
    eps = oracle.EPSILON
    assert obj_row[:vc].min() >= -128 * eps                      # find-entering-column returns NIL
    assert rhs_col[:m].min() >= 0.0                              # the ratio test keeps the RHS non-negative
    assert len(set(basis.tolist())) == m and basis.min() >= 0 and basis.max() < vc
    assert not obj_row[basis].any()                              # basic columns: reduced cost +0
    # the pivot trace is one entering column per pivot, every pivot row in range
    tr = t.pivot_trace(cap=1 << 20)
    assert len(tr) == npiv and tr[:, 0].min() >= 0 and tr[:, 0].max() < vc and tr[:, 1].max() < m
    # c'x from the generator against the tableau's objective value
    x = np.zeros(vc)
    x[basis] = rhs_col[:m]
    c = 0.5 + lp.synth.splitmix_u01(seed, n * m + m, n)
    obj = rhs_col[m]
    assert obj > 0 and abs(c @ x[:n] - obj) <= 1e-10 * abs(obj)
    # Ax <= b, A regenerated in row chunks on the support of x
    bvec = float(n) * (0.25 + 0.5 * lp.synth.splitmix_u01(seed, n * m, m))
    sup = np.flatnonzero(x[:n])
    xs = x[:n][sup]
    worst = 0.0
    for r0 in range(0, m, 1024):
        rows = np.arange(r0, min(r0 + 1024, m), dtype=np.int64)
        A = 0.05 + _u01_at(seed, rows[:, None] * n + sup[None, :])   # stream position of A[i][j]: i * n + j
        slack = bvec[rows] - A @ xs
        worst = min(worst, float(slack.min()))
        # a slack variable that is basic carries exactly that slack (to rounding of the solve)
        assert np.allclose(slack, x[n + rows], rtol=0, atol=1e-7 * np.abs(bvec).max())
    assert worst >= -1e-8 * np.abs(bvec).max()


@pytest.mark.timeout(1500, method="thread")
def test_config4_1024_lps_as_8_sub_batches_every_lp_vs_the_oracle():
    n, m, nl = 512, 256, 1024
    seeds = np.array([lp.synth.seed_for(4, k) for k in range(nl)], dtype=np.uint64)
    mb = lp.MultiDeviceBatch.synthetic(nl, n, m, seeds, n_devices=8)
    info = mb.info()
    t0 = time.perf_counter()
    st, npv = mb.solve()
    wall = time.perf_counter() - t0

    def one(k):
        M, b = lp.synth.tableau(n, m, int(seeds[k]))
        so, no, _ = oracle.solve(M, b)                            # (ctypes releases the GIL)
        return so, no, M, b

    pivots = 0
    with ThreadPoolExecutor(max_workers=16) as pool:
        for k0 in range(0, nl, 64):
            res = list(pool.map(one, range(k0, k0 + 64)))
            for k, (so, no, M, b) in zip(range(k0, k0 + 64), res):
                assert (int(st[k]), int(npv[k])) == (so, no) and so == oracle.OPTIMAL, k
                G, gb = mb.download(k)
                assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(gb, b), k
                pivots += no
    _note("cfg4_1024", {"sub_batches": info, "pivots": pivots, "wall_s": wall, "pivots_per_s": pivots / wall})
    assert pivots == int(npv.sum())
