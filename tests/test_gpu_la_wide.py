"""The persistent look-ahead on more than 32 workgroups (up to 64: tableaux of up to 16384 rows and
32768 stored columns) -- csrc/kernels_la_block.inc, k_la_block<., WGR = true> / la_exchange<., true>.

Up to 32 workgroups every WAVE publishes a record per exchange and every wave polls them all; above,
a workgroup's four wave winners meet in LDS, its first wave publishes ONE record, polls the nw records
and hands the result on through LDS (the poll traffic of the first form grows with the square of the
workgroups: DESIGN_experiments.md R5.16 / R5.18).  Same candidates, same decision rule
(find-entering-column / find-pivoting-row, src/simplex.lisp:362-397: the lexicographic minimum does not
depend on the shape of the reduction), same chains: the oracle's pivots and bits, whatever the form.
Here: 33 ... 64 workgroups by rows and by column pairs, both kernel forms (16 / 24 pending pivots),
requests that cut blocks short, and the recovery from a lost exchange behind blocks of 24 pivots in both
record forms (fault injection, test build)."""
import ctypes

import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()
LA_PERSISTENT, LA_TWO_LAUNCH = 1, 2


def _la_workgroups(n, m):
    ld = (n + 1 + 15) // 16 * 16
    return (max(m + 1, ld // 2) + 255) // 256


def _counts(L, h):
    out = (ctypes.c_int64 * 8)()
    lp.capi.check(L.mi355x_tab_path_counts(h, out), "path_counts")
    return list(out)


@pytest.fixture
def knobs():
    L = lp.capi.lib()
    yield L
    L.mi355x_tune_set_block(0)


@pytest.mark.parametrize("n,m,wg", [
    (700, 8300, 33),            # rows
    (500, 16383, 64),
    (20000, 1000, 40),          # column pairs
    (32767, 500, 64),
    (12000, 11000, 43),         # both large (1.06 GB stored): the objective row's and the RHS pair's owners far apart
], ids=["rows-33", "rows-64", "pairs-40", "pairs-64", "square-43"])
@pytest.mark.parametrize("block", [24, 16])
def test_workgroup_record_form_matches_the_oracle(knobs, n, m, wg, block):
    L = knobs
    assert _la_workgroups(n, m) == wg
    if block == 16 and wg not in (33, 40):
        pytest.skip("the 16-step form on two shapes only (suite time)")
    seed = lp.synth.seed_for(3, 8800 + wg + block)
    L.mi355x_tune_set_block(block if block == 16 else 0)
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, 0), "create_synthetic")
    t = lp.Tableau(None, lp.Problem(type="max"), None, None, n + m, m, {}, _handle=h)
    M = t.matrix
    b = t.basis_columns.copy()
    requests = [5, block, 2 * block + 3, 1]               # a short block, a full one, two full ones + a short one, a single pivot
    K = sum(requests)
    st, npiv, trace = oracle.solve(M, b, max_pivots=K, trace_cap=K, omp=True)
    assert (st, npiv) == (oracle.MAX_PIVOTS, K)
    t._touch()
    k = ctypes.c_int64(0)
    done = 0
    for i, q in enumerate(requests):
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, q, 1 if i == 0 else 0), "solve_async")
        rc = L.mi355x_tab_sync(h, ctypes.byref(k))
        done += q
        assert (rc, k.value) == (lp.capi.MI_RUNNING, done)
    c = _counts(L, h)
    assert L.mi355x_tab_block_size(h) == block
    assert c[LA_PERSISTENT] > 0 and c[LA_TWO_LAUNCH] == 0 and L.mi355x_tab_la_lost(h) == 0, c
    t._touch()
    tr = t.pivot_trace()
    bad = np.where((tr[:K] != trace[:K]).any(axis=1))[0]
    assert not len(bad), "first differing pivots %s: got %s, oracle %s" % (bad[:4], tr[bad[:4]], trace[bad[:4]])
    assert np.array_equal(t.basis_columns, b)
    G = t.matrix
    for r0 in range(0, m + 1, 2048):
        assert np.array_equal(G[r0:r0 + 2048].view(np.int64), M[r0:r0 + 2048].view(np.int64)), r0


@pytest.mark.parametrize("n,m,wg,fault_step", [
    (700, 8300, 33, 1), (700, 8300, 33, -1), (700, 8300, 33, -7), (700, 8300, 33, -24),
    (700, 8000, 32, -7),
    (4000, 3000, 12, 5), (4000, 3000, 12, -2), (4000, 3000, 12, -24),
    # ONE wave of the last workgroup gives up at the ratio exchange of step 2 / 23 while its workgroup's other
    # waves see the records and go on (every-wave-polls form; round-5 advisor finding: the workgroup's `done`
    # entry must be the minimum over its waves, or the sweep applies a pivot whose prow entries that wave never stored)
    (4000, 3000, 12, -1003), (4000, 3000, 12, -1024), (700, 8000, 32, -1008),
])
def test_lost_exchange_behind_blocks_of_24(n, m, wg, fault_step, hooks_lib):
    """As tests/test_gpu_fullsize.py test_lost_exchange_falls_back_to_two_launch_lookahead (a shape of 16
    pivots per pass), on shapes that run 24 per pass through the ring sweep -- the record-per-workgroup
    form (33 workgroups) and the record-per-wave form (32, 12): the last workgroup stops publishing
    (fault_step > 0) or gives up alone right after its ratio record (< 0: the leader commits a pivot that
    workgroup never completed); the sweep applies what EVERY workgroup completed (BlockCtl::done),
    k_la_rollback takes the leader's extra pivot back, the solve carries on on the two-launch look-ahead
    and ends with the oracle's pivots and bits.
    (Round 5 found k_sweepw_ring starting a slot column's chain at the hand-over of that extra, unapplied
    pivot -- the masks hold it --: wrong pivots from there on, on every shape of 24 per pass.  The block-16
    sweep reads the mask bit by bit and was right; the fault tests only had a block-16 shape.)"""
    L = hooks_lib
    assert _la_workgroups(n, m) == wg
    seed = lp.synth.seed_for(3, 8877)
    M0, b0 = lp.synth.tableau(n, m, seed)
    M, b = M0.copy(), b0.copy()
    K = 60
    st_o, npiv, trace = oracle.solve(M, b, max_pivots=K, trace_cap=K, omp=True)
    assert npiv == K
    try:
        L.mi355x_tune_set_la_max_spins(20000)
        L.mi355x_tune_set_la_fault(fault_step)
        t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
        k = ctypes.c_int64(0)
        rc = L.mi355x_tab_solve(t._h, 1, 1024.0, K, ctypes.byref(k))
        t._touch()
    finally:
        L.mi355x_tune_set_la_max_spins(0)
        L.mi355x_tune_set_la_fault(0)
    assert (rc, k.value) == (st_o, npiv)
    assert L.mi355x_tab_la_lost(t._h) == 1
    c = _counts(L, t._h)
    assert c[LA_PERSISTENT] > 0 and c[LA_TWO_LAUNCH] > 0 and c[4] > 0, c          # (c[4]: wide sweeps)
    assert np.array_equal(t.pivot_trace()[:npiv], trace[:npiv])
    assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64))
    assert np.array_equal(t.basis_columns, b)


def test_nonfinite_entering_column_is_flagged_across_workgroup_records():
    """An inf in the entering column, in a row the LAST of 33 workgroups owns: the ratio records carry a flag
    (OR over the lanes of a wave, over the four waves of a workgroup in LDS, over the nw records) that sends
    the solve back to the dense tableau, where the reference's arithmetic on inf / NaN is reproduced entry by
    entry (src/simplex.lisp:382-397 takes the pivot on whatever the quotients say).  Same pivots, same NaN
    pattern, same bits as the oracle."""
    L = lp.capi.lib()
    n, m = 700, 8300
    assert _la_workgroups(n, m) == 33
    seed = lp.synth.seed_for(3, 8899)
    M0, b0 = lp.synth.tableau(n, m, seed)
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, max_pivots=4, trace_cap=4, omp=True)
    assert npiv == 4
    col, row = int(trace[3][0]), 8250                     # the column that enters fourth; a row of workgroup 32
    assert row not in [int(r) for r in trace[:, 1]] and col < n
    M1 = M0.copy()
    M1[row, col] = float("inf")
    M, b = M1.copy(), b0.copy()
    K = 9
    with np.errstate(all="ignore"):
        st_o, npiv, trace = oracle.solve(M, b, max_pivots=K, trace_cap=K, omp=True)
    t = lp.Tableau(None, lp.Problem(type="max"), M1, b0, n + m, m, {})
    k = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(t._h, 1, 1024.0, K, ctypes.byref(k))
    c = _counts(L, t._h)
    t._touch()
    assert (rc, k.value) == (st_o, npiv), (rc, k.value, st_o, npiv)
    assert c[LA_PERSISTENT] > 0, c
    assert np.array_equal(t.pivot_trace()[:npiv], trace[:npiv])
    G = t.matrix
    nan_o, nan_g = np.isnan(M), np.isnan(G)
    assert nan_o.any() or np.isinf(M).any()
    assert np.array_equal(nan_o, nan_g)
    assert np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64))
    assert np.array_equal(t.basis_columns, b)
