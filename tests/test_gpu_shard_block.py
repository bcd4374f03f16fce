"""A column shard's look-ahead of a whole block as ONE persistent launch -- k_shard_la_block
(csrc/kernels_shard_block.inc; exchange mode 2 of mi355x_colpart_*, the default form of that mode since
round 6).  What the reference does per pivot (find-entering-column -> find-pivoting-row -> n-pivot-row,
src/simplex.lisp:362-389 inside the loop 453-461) happens here for up to 24 pivots inside one kernel per
device: thread g owns row g and column pair g of the shard's slice, the shard's winner is exchanged between
its workgroups through 64-byte record lines, the shards' winners (exchange A) and the owner's entering column
(exchange B) through the self-validating granules the shards push into each other's buffers -- polled INSIDE
the launch.  Whatever the number of shards and the form of their storage, the pivots and every bit must be
the oracle's.

Here: shapes beyond 64 workgroups (the 8-GPU shard of config 5 has 129: three record groups per polling
lane, BlockCtl::done read four entries per lane by the sweeps), blocks of 16 and 24, logical shards of one
device in ONE launch, requests that cut blocks short, and the recovery from lost exchanges (fault injection,
test build): on a lone shard every loss is recovered bit for bit; between shards only the loss at the
launch's first exchange (the co-residency check, which precedes a shard's first push) -- anything later
must end in an ERROR, never in a wrong answer."""
import ctypes
import importlib

import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _cp():
    return importlib.import_module("linear-programming_amd.colpart")


def _last_error():
    return lp.capi.lib().mi355x_last_error().decode("utf-8", "replace")


def _workgroups(rows, local_cols):
    ld = (local_cols + 1 + 15) // 16 * 16
    return (max(rows, ld // 2) + 255) // 256


@pytest.fixture
def knobs():
    L = lp.capi.lib()
    yield L
    L.mi355x_tune_set_block(0)
    L.mi355x_tune_set_colpart_exchange(0)
    L.mi355x_tune_set_shard_la_block(0)
    L.mi355x_tune_set_shard_self_hop(0)
    L.mi355x_tune_set_p2p_spins(0)


def _solve_and_compare(L, tab, M0, b0, cap_first, K=0):
    M, b = M0.copy(), b0.copy()
    so, no, trace = oracle.solve(M, b, max_pivots=K, trace_cap=1 << 16, omp=M0.size > 4_000_000)
    done = 0
    if cap_first:
        st, k = tab.solve(max_pivots=cap_first)                 # ends inside a block
        assert (st, k) == (lp.capi.MI_MAX_PIVOTS, cap_first)
        done = k
    st, k = tab.solve(max_pivots=(K - done) if K else 0)
    assert (st, done + k) == (so, no), (st, done + k, so, no)
    assert np.array_equal(tab.trace(no), trace)
    G, bg, last_row, last_col = tab.download()
    assert np.array_equal(bg, b)
    assert np.array_equal(last_col.view(np.int64), M[:, -1].view(np.int64))
    assert np.array_equal(last_row.view(np.int64), M[-1].view(np.int64))
    rows = M.shape[0]
    for r0 in range(0, rows, 4096):
        assert np.array_equal(G[r0:r0 + 4096].view(np.int64), M[r0:r0 + 4096].view(np.int64)), r0
    return no


@pytest.mark.parametrize("block", [16, 24])
@pytest.mark.parametrize("shards", [1, 3, 8])
@pytest.mark.parametrize("kind", ["compact", "dense"])
@pytest.mark.parametrize("n,m", [(700, 333), (2000, 900)])
def test_logical_shards_in_one_launch_match_the_oracle(knobs, n, m, kind, shards, block):
    L = knobs
    if block == 24 and (n, shards) not in ((2000, 1), (2000, 3), (700, 8)):
        pytest.skip("blocks of 24 on three combinations (suite time)")
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(5, 600 + shards + block))
    if kind == "dense":
        M0[:m, n:n + m] *= 2.0                                  # basis columns != unit columns: dense shards
    L.mi355x_tune_set_block(block)
    L.mi355x_tune_set_colpart_exchange(2)
    tab = _cp().NativeColumnPartition.from_arrays(M0, b0, shards)
    assert tab.block_size() == block and tab.la_stats()["live"]
    _solve_and_compare(L, tab, M0, b0, cap_first=block + 7)
    stats = tab.la_stats()
    assert stats["blocks"] > 0 and stats["losses"] == 0, stats
    tab.close()


@pytest.mark.parametrize("n,m,shards,wg,hop,K", [
    (66000, 100, 1, 129, 1, 52),     # 129 workgroups by column pairs (all of them hold pairs): three record groups per polling
                                     # lane, BlockCtl::done read beyond one entry per lane by the sweeps
    (600, 17000, 1, 67, 1, 52),      # 67 by rows: 3 hold pairs and read the ratio records, 64 take the leader's decision line
    (900, 16500, 3, 65, 1, 52),      # three shards of 65 workgroups in ONE launch (195 CUs)
    (256, 32768, 1, 129, 1, 28),     # the 8-GPU shard's row count: 129 workgroups, one holds the pairs
], ids=["pairs-129", "rows-67", "three-shards-of-65", "rows-129"])
def test_many_workgroups_per_shard(knobs, n, m, shards, wg, hop, K):
    L = knobs
    cp = _cp()
    assert _workgroups(m + 1, -(-n // shards)) == wg
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(5, 700 + wg + shards))
    L.mi355x_tune_set_block(24)
    L.mi355x_tune_set_colpart_exchange(2)
    L.mi355x_tune_set_shard_self_hop(hop)
    tab = cp.NativeColumnPartition.from_arrays(M0, b0, shards)
    assert tab.la_stats()["live"]
    _solve_and_compare(L, tab, M0, b0, cap_first=0, K=K)
    stats = tab.la_stats()
    assert stats["blocks"] > 0 and stats["losses"] == 0, stats
    tab.close()


def test_more_workgroups_than_cus_keep_the_step_kernels(knobs):
    """Logical shards whose workgroups would not all be resident together (each takes a CU's LDS) must not
    wait for each other inside one launch: the handle stays on the step kernels."""
    L = knobs
    n, m = 40000 * 9, 40                                         # nine shards of 79 workgroups (by column pairs)
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(5, 911))
    L.mi355x_tune_set_block(16)
    L.mi355x_tune_set_colpart_exchange(2)
    tab = _cp().NativeColumnPartition.from_arrays(M0, b0, 9)
    assert not tab.la_stats()["live"]
    _solve_and_compare(L, tab, M0, b0, cap_first=0, K=20)
    assert tab.la_stats()["blocks"] == 0
    tab.close()


def _fault_case(L, n, m, shards, fault, K, kind="compact", block=24):
    cp = _cp()
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(5, 800 + shards))
    if kind == "dense":
        M0[:m, n:n + m] *= 2.0
    M, b = M0.copy(), b0.copy()
    so, no, trace = oracle.solve(M, b, max_pivots=K, trace_cap=1 << 14)
    try:
        L.mi355x_tune_set_block(block)
        L.mi355x_tune_set_colpart_exchange(2)
        L.mi355x_tune_set_la_max_spins(20000)
        L.mi355x_tune_set_p2p_spins(40000)
        L.mi355x_tune_set_shard_la_fault(fault)
        tab = cp.NativeColumnPartition.from_arrays(M0, b0, shards)
        assert tab.la_stats()["live"]
        n_out = ctypes.c_int64(0)
        rc = L.mi355x_colpart_solve(tab._h, 1, 1024.0, K, ctypes.byref(n_out))
    finally:
        L.mi355x_tune_set_la_max_spins(0)
        L.mi355x_tune_set_p2p_spins(0)
        L.mi355x_tune_set_shard_la_fault(0)
        L.mi355x_tune_set_block(0)
        L.mi355x_tune_set_colpart_exchange(0)
    return tab, rc, n_out.value, (so, no, trace, M, b)


@pytest.mark.parametrize("kind", ["compact", "dense"])
@pytest.mark.parametrize("fault", [1, 5, 24, -1, -7, -24])
def test_lone_shard_recovers_from_every_lost_exchange(hooks_lib, fault, kind):
    """The last workgroup stops publishing from step fault - 1 on (> 0), or gives up alone right after its
    ratio record (< 0: the leader commits a pivot that workgroup never completed).  The sweep applies what
    EVERY workgroup completed, k_shard_la_rollback takes the leader's extra pivot back (basis, column maps,
    pivot count, trace), the handle goes on with the two-launch step: the oracle's pivots and bits."""
    L = hooks_lib
    tab, rc, k, (so, no, trace, M, b) = _fault_case(L, 1500, 1100, 1, fault, K=70, kind=kind)
    assert (rc, k) == (so, no)
    stats = tab.la_stats()
    assert stats["losses"] == 1 and stats["demoted"] and stats["blocks"] > 0, stats
    assert np.array_equal(tab.trace(no), trace)
    G, bg, _, _ = tab.download()
    assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)
    # re-armed after the clean blocks it was told to wait for: the persistent form runs again, bit for bit
    lp.capi.check(L.mi355x_colpart_debug_set_la_rearm(tab._h, 1), "rearm")
    L.mi355x_tune_set_shard_la_fault(0)
    so2, no2, _ = oracle.solve(M, b, max_pivots=48)
    st, k2 = tab.solve(max_pivots=48)
    assert (st, k2) == (so2, no2)
    G, bg, _, _ = tab.download()
    assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)
    after = tab.la_stats()
    assert after["blocks"] > stats["blocks"] and not after["demoted"], after
    tab.close()


@pytest.mark.parametrize("shards", [2, 3])
def test_loss_at_the_first_exchange_between_shards_is_recovered(hooks_lib, shards):
    """A shard whose workgroups are not all there at the launch's first exchange (the co-residency check)
    never pushes anything: every shard times out with NOTHING committed -- consistent, so all of them go on
    with the two-launch step and end with the oracle's pivots and bits."""
    L = hooks_lib
    tab, rc, k, (so, no, trace, M, b) = _fault_case(L, 900, 500, shards, 1, K=60)
    assert (rc, k) == (so, no), (rc, k, so, no, _last_error())
    stats = tab.la_stats()
    assert stats["losses"] == 1 and stats["demoted"], stats
    assert np.array_equal(tab.trace(no), trace)
    G, bg, _, _ = tab.download()
    assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)
    tab.close()


@pytest.mark.parametrize("fault", [6, -3])
def test_loss_inside_a_block_between_shards_is_an_error_not_a_wrong_answer(hooks_lib, fault):
    """Past the first exchange the shards of a partition cannot agree on what was committed without another
    exchange (a shard that lost its own workgroup stops a pivot earlier than the peers that still got its
    pair): the handle reports an error -- or, should the shards happen to agree, carries on correctly."""
    L = hooks_lib
    tab, rc, k, (so, no, trace, M, b) = _fault_case(L, 900, 500, 3, fault, K=60)
    if rc < 0:
        assert "disagree" in _last_error() or "lost" in _last_error() or "never arrived" in _last_error(), _last_error()
    else:
        assert (rc, k) == (so, no)
        G, bg, _, _ = tab.download()
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)
    tab.close()
