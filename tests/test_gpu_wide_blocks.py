"""Wide blocks (round 4): 24 / 28 pivots selected ahead and applied by ONE pass over the stored
tableau (k_sweepw + k_sweepw_rest), the default where the sweep dominates an iteration (tableaux of
240 MiB and column shards of 0.75 GB and more) and selectable everywhere with mi355x_tune_set_block.  Same
operands, same roundings, same order as 28 k_update launches (src/simplex.lisp:337-359), so the
pivots and every bit must be the oracle's -- through the blocking solve, through arbitrary
sequences of asynchronous requests (full blocks, remainders above and below 16, blocks cut short
by a cap or by optimality), on column shards in every exchange mode, and in the two-phase path."""
import ctypes

import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd, random_mixed_problem

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(params=[(24, 1, 0, 0), (28, 1, 0, 0), (24, 0, 0, 0), (28, 0, 0, 0), (24, 1, 1, 0), (28, 1, 1, 0), (24, 1, 0, 12), (28, 1, 0, 20)],
                ids=["24-ring", "28-ring", "24-registers", "28-registers", "24-ring-xcd-map", "28-ring-xcd-map", "24-ring-skew", "28-ring-skew"])
def wide(request):
    """Pivots per pass, the form of the streaming kernel -- the tile's rows through a per-wave LDS
    ring (k_sweepw_ring, the default since round 5) or through two register sets (k_sweepw) -- and,
    for the ring, which tile a workgroup takes: the grid's own order or one run of tiles per XCD
    (mi355x_tune_set_sweep_xcd_map: every (strip, tile) exactly once either way), and how tall the
    tiles are: all alike, or (round 6, mi355x_tune_set_sweep_skew: the default where one round of
    workgroups covers the tableau; forced here on every shape) the thirds of the tiles in dispatch
    order tr + skew / tr / tr - skew rows tall -- every row exactly once whatever the heights."""
    L = lp.capi.lib()
    k, ring, xmap, skew = request.param
    assert L.mi355x_tune_set_block(k) == k
    L.mi355x_tune_set_sweepw_ring(ring)
    L.mi355x_tune_set_sweep_xcd_map(xmap)
    L.mi355x_tune_set_sweep_skew(skew)
    L.mi355x_tune_set_select_mode(2)                    # small shapes too: the blocked path
    yield k
    L.mi355x_tune_set_block(0)
    L.mi355x_tune_set_sweepw_ring(1)
    L.mi355x_tune_set_sweep_xcd_map(0)
    L.mi355x_tune_set_sweep_skew(-1)
    L.mi355x_tune_set_select_mode(0)


def test_block_knob_values():
    L = lp.capi.lib()
    try:
        assert [L.mi355x_tune_set_block(k) for k in (0, 1, 7, 16, 17, 23, 24, 27, 28, 32, 99, -3)] == \
               [0, 1, 7, 16, 16, 16, 24, 24, 28, 28, 28, 0]
    finally:
        L.mi355x_tune_set_block(0)


@pytest.mark.parametrize("n,m,seed", [(700, 333, 6), (2000, 1100, 9), (1500, 2300, 11), (4100, 130, 12)])
def test_request_sequences_with_wide_blocks(n, m, seed, wide, request):
    """solve_async(n) for request sizes around the block size: full blocks, remainders of 1 .. 27
    pivots (split into a block of 16 and a short one above 16), and the LP ending inside a block."""
    L = lp.capi.lib()
    form = request.node.callspec.id
    if n == 1500 and ("xcd-map" in form or "28-registers" in form or "24-ring-skew" in form):
        pytest.skip("the largest shape (14 s: the oracle replays every request) on the default forms and one register form (suite time)")
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, seed))
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    assert L.mi355x_tab_block_size(t._h) in (wide, 1)      # (1: not yet on the compact representation)
    M, b = M0.copy(), b0.copy()
    done, first = 0, 1
    for req in (1, wide, wide - 1, wide + 1, 2 * wide, 17, 5, 3 * wide + 19, 16, wide + 16, 100000):
        lp.capi.check(L.mi355x_tab_solve_async(t._h, 1, 1024.0, req, first), "solve_async")
        assert L.mi355x_tab_block_size(t._h) == wide           # (on the compact representation now)
        first = 0
        k = ctypes.c_int64(0)
        rc = L.mi355x_tab_sync(t._h, ctypes.byref(k))
        st, npiv, _ = oracle.solve(M, b, max_pivots=req)
        done += npiv
        t._touch()
        assert k.value == done, (req, k.value, done)
        assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64)), req
        assert np.array_equal(t.basis_columns, b), req
        if st == oracle.OPTIMAL:
            assert rc == lp.capi.MI_OPTIMAL
            break
        assert rc == lp.capi.MI_RUNNING
    else:
        raise AssertionError("the LP did not finish")
    M2, b2 = M0.copy(), b0.copy()
    _, _, trace = oracle.solve(M2, b2, trace_cap=1 << 16)
    assert np.array_equal(t.pivot_trace(), trace)


@pytest.mark.parametrize("kind", ["max", "min"])
def test_min_problems_and_tolerance_factors_with_wide_blocks(kind, wide):
    rng = np.random.default_rng(5)
    n, m = 900, 400
    A = rng.integers(1, 9, (m, n)).astype(np.float64)          # integer data: ties, degenerate pivots
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = A
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = rng.integers(50, 90, m)
    M0[m, :n] = -rng.integers(1, 6, n) if kind == "max" else rng.integers(1, 6, n)
    if kind == "min":
        M0[:m, :n] *= rng.choice([1.0, 1.0, 1.0, -1.0], (m, n))
    b0 = np.arange(n, n + m, dtype=np.int64)
    for factor in (16.0, 1024.0, float(2 ** 20)):
        M, b = M0.copy(), b0.copy()
        st, npiv, trace = oracle.solve(M, b, is_max=(kind == "max"), factor=factor, max_pivots=700, trace_cap=700)
        t = lp.Tableau(None, lp.Problem(type=kind), M0, b0, n + m, m, {}, fp_tolerance_factor=factor)
        k = ctypes.c_int64(0)
        rc = lp.capi.lib().mi355x_tab_solve(t._h, int(kind == "max"), factor, 700, ctypes.byref(k))
        t._touch()
        assert (rc, k.value) == (st, npiv), factor
        assert np.array_equal(t.pivot_trace(), trace)
        assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64))


def test_two_phase_with_wide_blocks(wide):
    for seed in (1, 2, 3):
        p = random_mixed_problem(lp, 60, 25, 12, 6, seed)
        tabs = lp.build_tableau(p, p)
        art, main = tabs
        A, ab = art.matrix.copy(), art.basis_columns.copy()
        Mm, mb = main.matrix.copy(), main.basis_columns.copy()
        st, npv = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max, factor=main.fp_tolerance_factor)
        assert st == oracle.OPTIMAL
        lp.n_solve_tableau(tabs)
        assert np.array_equal(main.matrix.view(np.int64), Mm.view(np.int64)) and np.array_equal(main.basis_columns, mb)
        assert tuple(main.n_pivots) == tuple(int(x) for x in npv)


@pytest.mark.parametrize("exchange", [0, 2, 3], ids=["allreduce-semantics", "p2p-fused", "p2p-four-launches"])
@pytest.mark.parametrize("n_shards", [1, 3, 8])
def test_column_partition_with_wide_blocks(n_shards, exchange, wide):
    """mi355x_colpart_*: every shard's slice swept once per 24 / 28 pivots; full solve and a cap
    inside a block, one-workgroup and split look-ahead steps."""
    L = lp.capi.lib()
    for (n, m, seed, split) in ((700, 333, 2, 0), (1300, 600, 4, 2)):
        M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, seed))
        for cap in (0, wide + 9):
            M, b = M0.copy(), b0.copy()
            st, npiv, trace = oracle.solve(M, b, max_pivots=cap, trace_cap=1 << 16)
            h = ctypes.c_void_p()
            try:
                L.mi355x_tune_set_colpart_exchange(exchange)
                L.mi355x_tune_set_shard_la_split(split)
                lp.capi.check(L.mi355x_colpart_create(ctypes.byref(h), M0.shape[0], M0.shape[1], _ptr(M0), _ptr(b0),
                                                      n_shards), "colpart_create")
                k = ctypes.c_int64(0)
                rc = L.mi355x_colpart_solve(h, 1, 1024.0, cap, ctypes.byref(k))
                assert (rc, k.value) == (st, npiv), (n, cap)
                G, gb = np.empty_like(M0), np.empty_like(b0)
                lp.capi.check(L.mi355x_colpart_download(h, _ptr(G), _ptr(gb), None, None), "download")
                ec, cr, nn = np.empty(npiv, dtype=np.int64), np.empty(npiv, dtype=np.int64), ctypes.c_int64(0)
                lp.capi.check(L.mi355x_colpart_trace(h, _ptr(ec), _ptr(cr), npiv, ctypes.byref(nn)), "trace")
            finally:
                L.mi355x_tune_set_colpart_exchange(0)
                L.mi355x_tune_set_shard_la_split(0)
                if h:
                    L.mi355x_colpart_destroy(h)
            assert np.array_equal(np.stack([ec, cr], axis=1), trace[:npiv])
            assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(gb, b), (n, cap)


def test_wide_blocks_are_the_default_where_the_sweep_dominates():
    """A 240 MiB+ stored tableau that does not fit the persistent look-ahead takes 24 pivots per
    sweep by default (28 from 8 GB on: config 5, tests/test_gpu_fullsize.py); so does config 3 --
    the persistent look-ahead holds up to 24 pending pivots, and from 28 MiB of stored tableau on
    (48 MB at 3000 x 2000) the one-launch pass of 24 beats k_sweep16's 16; smaller tableaux stay at 16."""
    L = lp.capi.lib()
    for (n, m, want) in ((8192, 4096, 24), (200, 100, 16), (2000, 1500, 16), (3000, 2000, 24), (12000, 9000, 24)):
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, 12345, 0, -1, 0), "create")
        try:
            lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 0, 1), "to the compact representation")
            L.mi355x_tab_sync(h, None)
            assert L.mi355x_tab_block_size(h) == want, (n, m)
            if want == 24:                                   # ... and it is the oracle's solve: 100 pivots
                M, b = lp.synth.tableau(n, m, 12345)
                st, npiv, trace = oracle.solve(M, b, max_pivots=100, trace_cap=100, omp=True)
                k = ctypes.c_int64(0)
                rc = L.mi355x_tab_solve(h, 1, 1024.0, 100, ctypes.byref(k))
                assert (rc, k.value) == (st, npiv)
                last_col = np.empty(m + 1)
                gb = np.empty(m, dtype=np.int64)
                lp.capi.check(L.mi355x_tab_download(h, None, _ptr(gb), None, _ptr(last_col)), "download")
                assert np.array_equal(gb, b) and np.array_equal(last_col.view(np.int64), M[:, -1].copy().view(np.int64))
        finally:
            L.mi355x_tab_destroy(h)
