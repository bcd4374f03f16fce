"""How a request ENDS and how a handle's first request STARTS must not change what it computes.

  * csrc/capi_tab_impl.inc read_ctls: the status read-back behind a request is, by default, a kernel
    that publishes the control block to pinned memory (k_ctl_publish) and a host poll of its
    sequence number; mi355x_tune_set_ctl_wait selects the two older forms (copy + polled
    hipStreamQuery, copy + hipStreamSynchronize).  Same status, same pivot count, same tableau.
  * csrc/capi_tab_impl.inc prime_block_kernels: a handle's first block is preceded by one EMPTY block
    of every kernel form its requests can pick (a look-ahead of 0 steps + the sweep for its empty
    list).  Primed or not, split into requests of any lengths or not: the oracle's pivots and bits
    (src/simplex.lisp:453-461 has no notion of a request)."""
import ctypes

import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture
def knobs():
    L = lp.capi.lib()
    yield L
    L.mi355x_tune_set_ctl_wait(2)
    L.mi355x_tune_set_prime(1)


def _solve_in_requests(n, m, seed, requests, is_batch=False):
    """-> (trace, basis, last_row, last_col, tableau) after the given request lengths, one sync each."""
    L = lp.capi.lib()
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, 0), "create_synthetic")
    t = lp.Tableau(None, lp.Problem(type="max"), None, None, n + m, m, {}, _handle=h)
    k = ctypes.c_int64(0)
    done = 0
    for i, q in enumerate(requests):
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, q, 1 if i == 0 else 0), "solve_async")
        rc = L.mi355x_tab_sync(h, ctypes.byref(k))
        done += q
        assert (rc, k.value) == (lp.capi.MI_RUNNING, done), (rc, k.value, done)
    t._touch()
    return t.pivot_trace().copy(), t.basis_columns.copy(), t.matrix.copy()


@pytest.mark.parametrize("n,m,requests", [
    (4095, 3200, [5, 20, 24, 3, 44]),        # 24 per sweep (100 MiB stored): short, the block of 20, a full one, <= 16, 24 + 20
    (1500, 1200, [5, 20, 16, 1, 33]),        # 16 per sweep
], ids=["block24", "block16"])
def test_wait_modes_and_priming_leave_the_bits_alone(knobs, n, m, requests):
    L = knobs
    seed = lp.synth.seed_for(3, 9100 + n)
    M, b = lp.synth.tableau(n, m, seed)
    K = sum(requests)
    st, npiv, trace = oracle.solve(M, b, max_pivots=K, trace_cap=K, omp=True)
    assert (st, npiv) == (oracle.MAX_PIVOTS, K)
    for wait, prime in [(2, 1), (1, 1), (0, 1), (2, 0), (0, 0)]:
        L.mi355x_tune_set_ctl_wait(wait)
        L.mi355x_tune_set_prime(prime)
        tr, basis, G = _solve_in_requests(n, m, seed, requests)
        assert np.array_equal(tr, trace), (wait, prime)
        assert np.array_equal(basis, b), (wait, prime)
        assert np.array_equal(G.view(np.int64), M.view(np.int64)), (wait, prime)


def test_a_request_longer_than_the_poll_limit_still_returns(knobs):
    """The memory poll gives up after MI355X_SPIN_WAIT_US and the thread sleeps in
    hipStreamSynchronize; the sequence number must be there when that returns.  (The limit is read
    once per process: here only the default -- 200 ms -- can be exercised, with a request that takes
    longer: 4 400 pivots of config 3 are ~50 ms, so five such requests back to back on one stream.)"""
    L = knobs
    hs = []
    k = ctypes.c_int64(0)
    for i in range(5):
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), 8192, 4096, lp.synth.seed_for(3, 9200 + i), 0, -1, 0), "create")
        hs.append(h)
    # all five on ONE stream: the first sync -- on the LAST handle -- waits for ~5 x 50 ms of queued work
    import torch
    s = torch.cuda.Stream()
    for h in hs:
        lp.capi.check(L.mi355x_tab_set_stream(h, ctypes.c_void_p(s.cuda_stream), 0), "set_stream")
    for h in hs:
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 4400, 1), "solve_async")
    rc = L.mi355x_tab_sync(hs[-1], ctypes.byref(k))
    assert (rc, k.value) == (lp.capi.MI_RUNNING, 4400)
    for h in hs[:-1]:
        assert (L.mi355x_tab_sync(h, ctypes.byref(k)), k.value) == (lp.capi.MI_RUNNING, 4400)
    for h in hs:
        L.mi355x_tab_destroy(h)


def test_batch_read_back_through_the_published_block(knobs):
    """read_ctls with n > 1: a batch's control blocks arrive together (csrc/capi_batch.inc)."""
    L = knobs
    seeds = np.array([lp.synth.seed_for(4, 40 + i) for i in range(24)], dtype=np.uint64)
    res = []
    for wait in (2, 0):
        L.mi355x_tune_set_ctl_wait(wait)
        b = lp.TableauBatch.synthetic(24, 96, 64, seeds)
        st, pv = b.solve()
        res.append((st.copy(), pv.copy(), [b.download(i) for i in (0, 7, 23)]))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert (res[0][0] == 0).all()
    for x, y in zip(res[0][2], res[1][2]):
        for u, v in zip(x, y):
            assert np.array_equal(np.asarray(u), np.asarray(v))
