"""NaNs in the tableau (inf - inf after an overflow) and the reference's scan order.

find-entering-column (src/simplex.lisp:362-379) starts from column 0 and replaces the incumbent
only by a strictly smaller entry: a NaN entry never becomes the entering column, a NaN in column 0
ends the solve as "optimal".  find-pivoting-row (:382-389) takes the first eligible row and
replaces it only by a strictly smaller quotient: a NaN quotient wins iff its row is the first
eligible one.  The device reductions are trees over lanes, waves and workgroups; these cases pin
that they take the sequential decisions whatever the order -- on every code path (persistent and
two-launch look-ahead, per-pivot compact, dense single-workgroup and split select)."""
import ctypes

import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()
NAN, INF = float("nan"), float("inf")

MODES = {
    "default":              dict(),
    "two-launch lookahead": dict(lookahead=1),
    "per-pivot compact":    dict(block=1),
    "dense, one workgroup": dict(compact=0, select=1),
    "dense, split select":  dict(compact=0, select=2),
}


def _apply(mode):
    L = lp.capi.lib()
    L.mi355x_tune_set_lookahead_mode(mode.get("lookahead", 0))
    L.mi355x_tune_set_block(mode.get("block", 0))
    L.mi355x_tune_set_compact(mode.get("compact", 1))
    L.mi355x_tune_set_select_mode(mode.get("select", 0))


def _reset():
    _apply({})


def _tableau(A, b, c):
    m, n = A.shape
    M = np.zeros((m + 1, n + m + 1))
    M[:m, :n] = A
    M[np.arange(m), n + np.arange(m)] = 1.0
    M[:m, -1] = b
    M[m, :n] = c
    return M, np.arange(n, n + m, dtype=np.int64)


def _check(M0, b0, cap, is_max=True):
    M, b = M0.copy(), b0.copy()
    with np.errstate(all="ignore"):
        st_o, npiv, trace = oracle.solve(M, b, is_max=is_max, max_pivots=cap, trace_cap=cap)
    L = lp.capi.lib()
    try:
        for name, mode in MODES.items():
            _apply(mode)
            t = lp.Tableau(None, lp.Problem(type="max" if is_max else "min"), M0, b0, M0.shape[1] - 1, M0.shape[0] - 1, {})
            k = ctypes.c_int64(0)
            rc = L.mi355x_tab_solve(t._h, int(is_max), 1024.0, cap, ctypes.byref(k))
            t._touch()
            got = t.pivot_trace()
            assert (rc, k.value) == (st_o, npiv), "%s: status/pivots (%d, %d), oracle (%d, %d); trace %s vs %s" % (
                name, rc, k.value, st_o, npiv, got.tolist(), trace.tolist())
            assert np.array_equal(got, trace), "%s: trace %s, oracle %s" % (name, got.tolist(), trace.tolist())
            G = t.matrix
            nan_o, nan_g = np.isnan(M), np.isnan(G)
            assert np.array_equal(nan_o, nan_g), name
            assert np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64)), name
            assert np.array_equal(t.basis_columns, b), name
    finally:
        _reset()
    return st_o, npiv, trace


def test_nan_objective_entries_never_enter():
    """NaNs around the minimum, ties at the minimum: the lowest-index strict minimum among the
    numbers enters (here column 5), never a NaN and never a later tie."""
    rng = np.random.default_rng(7)
    A = rng.uniform(0.5, 2.0, (4, 12))
    c = np.array([3.0, 1.0, NAN, -2.0, NAN, -7.0, NAN, -7.0, -1.0, NAN, -7.0, 2.0])
    M0, b0 = _tableau(A, rng.uniform(1.0, 2.0, 4), c)
    st, npiv, trace = _check(M0, b0, cap=1)
    assert trace[0][0] == 5


def test_nan_in_objective_column_zero_ends_the_solve():
    """Column 0 holds a NaN: nothing replaces it, (fp< NaN 0) fails, the tableau counts as optimal."""
    rng = np.random.default_rng(8)
    A = rng.uniform(0.5, 2.0, (3, 9))
    c = np.array([NAN, -1.0, -5.0, 2.0, -INF, 0.0, -3.0, NAN, -2.0])
    M0, b0 = _tableau(A, rng.uniform(1.0, 2.0, 3), c)
    st, npiv, trace = _check(M0, b0, cap=5)
    assert (st, npiv) == (oracle.OPTIMAL, 0)


def test_nan_quotient_wins_only_in_the_first_eligible_row():
    """Entering column 0 (the only negative objective entry).  Case 1: the first eligible row's
    quotient is NaN (rhs NaN) -> that row is the pivot row.  Case 2: the NaN quotient sits in a
    later eligible row -> it is ignored, the smallest real quotient wins."""
    A = np.array([[0.0, 1.0, 2.0],      # row 0: not eligible (entry 0)
                  [2.0, 1.0, 0.5],      # row 1: first eligible
                  [4.0, 3.0, 1.0],      # row 2
                  [1.0, 0.5, 2.0],      # row 3
                  [8.0, 1.0, 1.0]])     # row 4
    c = np.array([-1.0, 2.0, 3.0])
    M0, b0 = _tableau(A, np.array([1.0, NAN, 8.0, 0.5, 4.0]), c)
    st, npiv, trace = _check(M0, b0, cap=1)
    assert trace.tolist() == [[0, 1]]
    M0, b0 = _tableau(A, np.array([1.0, 6.0, NAN, 0.5, NAN]), c)
    st, npiv, trace = _check(M0, b0, cap=1)
    assert trace.tolist() == [[0, 3]]


def test_nan_quotient_rows_across_workgroups():
    """The same two rules on a tableau tall enough for the split select and several look-ahead
    workgroups (1500 rows): NaN quotients far apart, the first eligible row deep in the tableau."""
    rng = np.random.default_rng(9)
    m, n = 1500, 40
    A = rng.uniform(0.5, 2.0, (m, n))
    A[:700, 3] = -1.0                                   # column 3: rows below 700 are not eligible
    b = rng.uniform(1.0, 2.0, m)
    c = rng.uniform(0.5, 1.0, n)
    c[3] = -9.0
    b1 = b.copy(); b1[700] = NAN; b1[1490] = NAN        # first eligible row 700 has the NaN quotient
    M0, b0 = _tableau(A, b1, c)
    st, npiv, trace = _check(M0, b0, cap=1)
    assert trace.tolist() == [[3, 700]]
    b2 = b.copy(); b2[701] = NAN; b2[1499] = NAN        # NaNs behind the first eligible row: ignored
    M0, b0 = _tableau(A, b2, c)
    st, npiv, trace = _check(M0, b0, cap=3)
    assert trace[0][0] == 3 and trace[0][1] not in (701, 1499)


def test_regression_extreme_magnitudes_case():
    """The hypothesis example that exposed the order dependence (ties at -inf next to NaNs in the
    objective row after four pivots)."""
    n, m, seed, lo, hi = 48, 5, 1, -300, 160
    rng = np.random.default_rng(seed)
    mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(lo, hi + 1, shape)   # noqa: E731
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = mag((m, n)) * rng.choice([1.0, 1.0, -1.0], (m, n))
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = mag(m)
    M0[m, :n] = -mag(n)
    b0 = np.arange(n, n + m, dtype=np.int64)
    st, npiv, trace = _check(M0, b0, cap=60)
    assert npiv == 5


def test_batches_with_overflowing_members_every_mode():
    """Batches whose members overflow at different pivots (each such member sends the whole batch
    to the dense tableaux while the others are in the middle of theirs), in every batch mode:
    every member ends where the oracle ends on it alone.  (Mode 1, the lockstep driver, used to
    change the representation with the other members' pivots half done.)"""
    L = lp.capi.lib()
    meta = np.random.default_rng(123)
    try:
        for bi in range(160):
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
                    st_o, npiv, _ = oracle.solve(M, b, max_pivots=60)
                ref.append((st_o, npiv, M, b))
            L.mi355x_tune_set_batch_mode(mode)
            batch = lp.TableauBatch.from_arrays(np.stack(Ms), np.stack(Bs))
            st, npv = batch.solve(max_pivots=60)
            for k in range(nl):
                G, bg = batch.download(k)
                so, no, M, b = ref[k]
                where = "batch %d (mode %d, %d x %d, %d LPs) member %d" % (bi, mode, n, m, nl, k)
                assert (int(st[k]), int(npv[k])) == (so, no), where
                nan_o, nan_g = np.isnan(M), np.isnan(G)
                assert np.array_equal(nan_o, nan_g), where
                assert np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64)), where
                assert np.array_equal(bg, b), where
    finally:
        L.mi355x_tune_set_batch_mode(0)


@pytest.mark.parametrize("split", [1, 2], ids=["one-workgroup-step", "split-step"])
def test_dense_column_shards_nan_in_a_later_shards_first_column(split):
    """Round-2 advisor finding: the NaN-in-column-0 rule is about GLOBAL column 0.  A dense column
    shard numbers its columns from 0, and the look-ahead step of a shard priced its slice without
    the shard's column offset -- a NaN objective entry in LOCAL column 0 of shard r > 0 became the
    unbeatable candidate and every shard stopped as optimal.  The reference simply never enters
    that column (find-entering-column, src/simplex.lisp:362-379)."""
    import importlib
    cp = importlib.import_module("linear-programming_amd.colpart")
    L = lp.capi.lib()
    rng = np.random.default_rng(5)
    m, n = 25, 40
    A = rng.uniform(0.1, 1.5, (m, n))
    M0, b0 = _tableau(A, rng.uniform(1.0, 4.0, m), -rng.uniform(0.5, 2.0, n))
    M0[:m, n:n + m] *= 2.0                               # basis columns != e_i: dense shards, all 65 columns distributed
    for shards in (2, 3):
        first = [(n + m) // shards * r + min(r, (n + m) % shards) for r in range(shards)]
        for r in range(1, shards):
            M1 = M0.copy()
            M1[m, first[r]] = NAN                        # local column 0 of shard r
            M, b = M1.copy(), b0.copy()
            with np.errstate(all="ignore"):
                so, no, trace = oracle.solve(M, b, trace_cap=4096)
            assert no > 3                                # the oracle pivots on: the NaN column just never enters
            try:
                L.mi355x_tune_set_shard_la_split(split)
                tab = cp.NativeColumnPartition.from_arrays(M1, b0, shards)
                st, k = tab.solve()
            finally:
                L.mi355x_tune_set_shard_la_split(0)
            assert (st, k) == (so, no), (shards, r, st, k, so, no)
            assert np.array_equal(tab.trace(no), trace)
            G, bg, _, _ = tab.download()
            nan_o, nan_g = np.isnan(M), np.isnan(G)
            assert np.array_equal(nan_o, nan_g)
            assert np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64)) and np.array_equal(bg, b)
            tab.close()
