"""A way out of a solve.  The reference's loop (src/simplex.lisp:453-461) has no iteration cap and
no anti-cycling rule: Dantzig pricing + lowest-index ties cycle on the textbook examples, whose
coefficients are dyadic, so they cycle EXACTLY in double-float.  In Lisp such a solve can be
interrupted; a blocking foreign call cannot -- hence mi355x_*_cancel and the bounded chunks of the
drivers.  The GPU must (i) follow the oracle's periodic pivot sequence bit for bit under max_pivots,
(ii) return MI_CANCELLED when another thread cancels, with whole pivots only: the tableau equals
the oracle's after exactly the reported number of pivots, and a further solve call carries on."""
import ctypes
import threading
import time

import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _tab(A, b, c):
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    M = np.zeros((m + 1, n + m + 1))
    M[:m, :n] = A
    M[:m, n:n + m] = np.eye(m)
    M[:m, -1] = b
    M[m, :n] = -np.asarray(c, dtype=np.float64)
    return M, np.arange(n, n + m, dtype=np.int64)


def beale():
    """Beale's example with dyadic coefficients: cycles with period 6 under Dantzig's rule and
    lowest-index ties (every ratio is 0 / a)."""
    return _tab([[0.25, -8, -1, 9], [0.5, -12, -0.5, 3], [0, 0, 1, 0]], [0, 0, 1], [0.75, -20, 0.5, -6])


def chvatal():
    """Chvatal's example (Linear Programming, 1983, p. 31): period 6 as well."""
    return _tab([[0.5, -5.5, -2.5, 9], [0.5, -1.5, -0.5, 1], [1, 0, 0, 0]], [0, 0, 1], [10, -57, -9, -24])


def embedded(base, n_extra, m_extra, seed):
    """The cycling LP with extra columns that never enter (objective coefficient <= 0 in max form ->
    reduced cost >= 0, and zero in the cycling rows, so it stays what it is) and extra rows that
    never bind (zero on the cycling columns): the same pivots on a tableau large enough for the
    multi-workgroup paths."""
    M0, b0 = base
    m, n = M0.shape[0] - 1, M0.shape[1] - 1 - (M0.shape[0] - 1)
    rng = np.random.default_rng(seed)
    A = np.zeros((m + m_extra, n + n_extra))
    A[:m, :n] = M0[:m, :n]
    A[m:, n:] = rng.integers(1, 5, (m_extra, n_extra)).astype(np.float64)
    b = np.concatenate([M0[:m, -1], np.full(m_extra, 2.0 ** 40)])
    c = np.concatenate([-M0[m, :n], -rng.integers(0, 4, n_extra).astype(np.float64)])
    return _tab(A, b, c)


CASES = {"beale": beale, "chvatal": chvatal,
         "beale-wide": lambda: embedded(beale(), 700, 300, 1),
         "chvatal-tall": lambda: embedded(chvatal(), 260, 1200, 2)}


@pytest.fixture(params=["default", "blocked", "two-launch", "per-pivot", "dense"])
def path(request):
    """The solve loop's implementations (knobs are snapshotted when a handle is created)."""
    L = lp.capi.lib()
    knobs = {"default": [], "blocked": [(L.mi355x_tune_set_resident, 1, 0)],
             "two-launch": [(L.mi355x_tune_set_resident, 1, 0), (L.mi355x_tune_set_lookahead_mode, 1, 0),
                            (L.mi355x_tune_set_select_mode, 2, 0)],
             "per-pivot": [(L.mi355x_tune_set_resident, 1, 0), (L.mi355x_tune_set_block, 1, 0)],
             "dense": [(L.mi355x_tune_set_compact, 0, 1)]}[request.param]
    for fn, on, _ in knobs:
        fn(on)
    yield request.param
    for fn, _, off in knobs:
        fn(off)


@pytest.mark.parametrize("case", sorted(CASES))
def test_cycling_lps_follow_the_oracle_under_a_pivot_cap(case, path):
    M0, b0 = CASES[case]()
    cap = 6 * 37 + 4
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, max_pivots=cap, trace_cap=cap)
    assert (st, npiv) == (oracle.MAX_PIVOTS, cap)
    assert np.array_equal(trace[:6], trace[6:12]) and np.array_equal(trace[:cap - 6], trace[6:])   # it cycles
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, M0.shape[1] - 1, M0.shape[0] - 1, {})
    k = ctypes.c_int64(0)
    rc = lp.capi.lib().mi355x_tab_solve(t._h, 1, 1024.0, cap, ctypes.byref(k))
    t._touch()
    assert (rc, k.value) == (lp.capi.MI_MAX_PIVOTS, cap)
    assert np.array_equal(t.pivot_trace(), trace)
    assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(t.basis_columns, b)


def _cancel_after(fn, delay):
    th = threading.Thread(target=lambda: (time.sleep(delay), fn()))
    th.start()
    return th


@pytest.mark.parametrize("case", ["beale", "chvatal-tall"])
def test_cancel_from_a_second_thread_leaves_whole_pivots(case, path):
    L = lp.capi.lib()
    M0, b0 = CASES[case]()
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, M0.shape[1] - 1, M0.shape[0] - 1, {})
    h = t._h
    k = ctypes.c_int64(0)
    # (the oracle replays every pivot the GPU got through before the cancel -- single thread, 1 201 x 1 461 for
    # the tall case: 0.15 s of GPU pivots there instead of 0.4 s, which was 12-25 s of replay per variant)
    th = _cancel_after(lambda: L.mi355x_tab_cancel(h), 0.15 if case == "chvatal-tall" else 0.4)
    t0 = time.perf_counter()
    rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(k))          # no cap: only the cancel ends it
    dt = time.perf_counter() - t0
    th.join()
    t._touch()
    assert rc == lp.capi.MI_CANCELLED, (rc, k.value)
    assert k.value > 0 and dt < 10.0
    M, b = M0.copy(), b0.copy()
    st, npiv, _ = oracle.solve(M, b, max_pivots=int(k.value))
    assert npiv == k.value
    assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(t.basis_columns, b)
    tail = t.pivot_trace(cap=1 << 20)
    assert len(tail) == min(k.value, 1 << 20)
    # the handle carries on: 13 more pivots are the oracle's next 13
    k2 = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(h, 1, 1024.0, 13, ctypes.byref(k2))
    t._touch()
    assert (rc, k2.value) == (lp.capi.MI_MAX_PIVOTS, 13)
    st, npiv, _ = oracle.solve(M, b, max_pivots=13)
    assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(t.basis_columns, b)


def test_cancel_is_sticky_and_consumed_by_the_next_solve():
    L = lp.capi.lib()
    M0, b0 = lp.synth.tableau(60, 30, lp.synth.seed_for(2, 5))
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, 90, 30, {})
    assert L.mi355x_tab_cancel(t._h) == 0
    k = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(t._h, 1, 1024.0, 0, ctypes.byref(k))
    # an LP that finishes inside the first chunk finishes (nothing to cancel between chunks)
    assert rc in (lp.capi.MI_OPTIMAL, lp.capi.MI_CANCELLED)
    rc = L.mi355x_tab_solve(t._h, 1, 1024.0, 0, ctypes.byref(k))
    t._touch()
    assert rc == lp.capi.MI_OPTIMAL
    M, b = M0.copy(), b0.copy()
    oracle.solve(M, b)
    assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64))
    assert L.mi355x_tab_cancel(None) == lp.capi.MI_BAD_ARG


def test_python_mirror_raises_solve_cancelled():
    M0, b0 = beale()
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, M0.shape[1] - 1, M0.shape[0] - 1, {})
    t._h
    th = _cancel_after(lambda: lp.simplex.cancel_solve(t), 0.3)
    with pytest.raises(lp.simplex.SolveCancelled):
        lp.n_solve_tableau(t)
    th.join()
    assert t.n_pivots > 0


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_batch_cancel_reports_finished_and_unfinished_members(mode):
    """A batch with cycling members next to ordinary ones: the ordinary ones end OPTIMAL with the
    oracle's bits, the cycling ones are reported MI_RUNNING after whole pivots, the call returns
    MI_CANCELLED.  mode: the batch drivers (0 default = resident, 1 lockstep, 2 one workgroup per
    LP, 3 look-ahead per LP + sweeps over all LPs)."""
    L = lp.capi.lib()
    base, bb = embedded(beale(), 40, 17, 3)
    n, m = base.shape[1] - 1 - (base.shape[0] - 1), base.shape[0] - 1
    Ms, Bs = [], []
    for i in range(6):
        if i in (1, 4):
            Ms.append(base.copy()); Bs.append(bb.copy())
        else:
            M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(4, i))
            Ms.append(M0); Bs.append(b0)
    try:
        L.mi355x_tune_set_batch_mode(mode)
        batch = lp.TableauBatch.from_arrays(np.stack(Ms), np.stack(Bs))
    finally:
        L.mi355x_tune_set_batch_mode(0)
    st = np.zeros(6, dtype=np.int32)
    npv = np.zeros(6, dtype=np.int64)
    th = _cancel_after(lambda: L.mi355x_batch_cancel(batch._h), 0.4)
    rc = L.mi355x_batch_solve(batch._h, 1, 1024.0, 0, _ptr(st), _ptr(npv))
    th.join()
    assert rc == lp.capi.MI_CANCELLED
    for i in range(6):
        M, b = Ms[i].copy(), Bs[i].copy()
        G, gb = batch.download(i)
        if i in (1, 4):
            assert st[i] == lp.capi.MI_RUNNING and npv[i] > 0
            so, no, _ = oracle.solve(M, b, max_pivots=int(npv[i]))
            assert no == npv[i]
        else:
            so, no, _ = oracle.solve(M, b)
            assert (int(st[i]), int(npv[i])) == (so, no)
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(gb, b), i


def test_multibatch_cancel_through_the_worker_threads():
    L = lp.capi.lib()
    base, bb = embedded(chvatal(), 30, 11, 5)
    n, m = base.shape[1] - 1 - (base.shape[0] - 1), base.shape[0] - 1
    Ms, Bs = [], []
    for i in range(7):
        if i == 5:
            Ms.append(base.copy()); Bs.append(bb.copy())
        else:
            M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(4, 20 + i))
            Ms.append(M0); Bs.append(b0)
    mb = lp.MultiDeviceBatch.from_arrays(np.stack(Ms), np.stack(Bs), n_devices=3)
    st = np.zeros(7, dtype=np.int32)
    npv = np.zeros(7, dtype=np.int64)
    th = _cancel_after(lambda: L.mi355x_multibatch_cancel(mb._h), 0.4)
    rc = L.mi355x_multibatch_solve(mb._h, 1, 1024.0, 0, _ptr(st), _ptr(npv))
    th.join()
    assert rc == lp.capi.MI_CANCELLED
    for i in range(7):
        M, b = Ms[i].copy(), Bs[i].copy()
        G, gb = mb.download(i)
        if i == 5:
            assert st[i] == lp.capi.MI_RUNNING
            oracle.solve(M, b, max_pivots=int(npv[i]))
        else:
            so, no, _ = oracle.solve(M, b)
            assert (int(st[i]), int(npv[i])) == (so, no), i
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(gb, b), i
    # sub-batches whose members had all finished keep no stale request: the same call again ends
    # only by a second cancel (the cycling member is still there)
    th = _cancel_after(lambda: L.mi355x_multibatch_cancel(mb._h), 0.3)
    rc = L.mi355x_multibatch_solve(mb._h, 1, 1024.0, 0, _ptr(st), _ptr(npv))
    th.join()
    assert rc == lp.capi.MI_CANCELLED and st[5] == lp.capi.MI_RUNNING


@pytest.mark.parametrize("n_shards,exchange", [(1, 0), (3, 0), (2, 2)])
def test_colpart_cancel_leaves_every_shard_swept(n_shards, exchange):
    L = lp.capi.lib()
    M0, b0 = embedded(beale(), 90, 40, 9)
    h = ctypes.c_void_p()
    try:
        L.mi355x_tune_set_colpart_exchange(exchange)
        lp.capi.check(L.mi355x_colpart_create(ctypes.byref(h), M0.shape[0], M0.shape[1], _ptr(M0), _ptr(b0), n_shards),
                      "colpart_create")
    finally:
        L.mi355x_tune_set_colpart_exchange(0)
    try:
        k = ctypes.c_int64(0)
        th = _cancel_after(lambda: L.mi355x_colpart_cancel(h), 0.4)
        rc = L.mi355x_colpart_solve(h, 1, 1024.0, 0, ctypes.byref(k))
        th.join()
        assert rc == lp.capi.MI_CANCELLED and k.value > 0
        M, b = M0.copy(), b0.copy()
        st, npiv, _ = oracle.solve(M, b, max_pivots=int(k.value))
        G, gb = np.empty_like(M0), np.empty_like(b0)
        lp.capi.check(L.mi355x_colpart_download(h, _ptr(G), _ptr(gb), None, None), "download")
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(gb, b)
    finally:
        L.mi355x_colpart_destroy(h)


# ---- two-phase: a request on EITHER handle ends the call, whichever phase is running ------------
def _two_phase_cycling():
    """Beale's cycling LP plus one variable pinned by an equality row (y = 1): build-tableau returns
    (art main); phase 1 ends after the pivot that brings y in, phase 2 cycles for ever."""
    names = ["x1", "x2", "x3", "x4", "y"]
    p = lp.Problem(type="max", vars=names, objective_var="z",
                   objective_func=[("x1", 0.75), ("x2", -20.0), ("x3", 0.5), ("x4", -6.0)],
                   constraints=[("<=", [("x1", 0.25), ("x2", -8.0), ("x3", -1.0), ("x4", 9.0)], 0.0),
                                ("<=", [("x1", 0.5), ("x2", -12.0), ("x3", -0.5), ("x4", 3.0)], 0.0),
                                ("<=", [("x3", 1.0)], 1.0),
                                ("=", [("y", 1.0)], 1.0)])
    tabs = lp.build_tableau(p, p)
    assert isinstance(tabs, list)
    return tabs


@pytest.mark.parametrize("which", ["art", "main"])
def test_two_phase_cancel_on_either_handle_during_phase_2(which):
    """(round-4 advisor finding) phase 2 polled only the main handle's flag: a request on the
    artificial handle was never seen and a cycling phase 2 could not be ended through it."""
    L = lp.capi.lib()
    art, main = _two_phase_cycling()
    ha, hm = art._h, main._h
    npv = (ctypes.c_int64 * 2)()
    th = _cancel_after(lambda: L.mi355x_tab_cancel(ha if which == "art" else hm), 0.4)
    t0 = time.perf_counter()
    rc = L.mi355x_solve_two_phase(ha, hm, 1, 1024.0, npv)
    dt = time.perf_counter() - t0
    th.join()
    art._touch(); main._touch()
    assert rc == lp.capi.MI_CANCELLED and dt < 10.0, (rc, dt)
    assert npv[0] >= 1 and npv[1] > 0                       # phase 1 done, phase 2 was running
    # whole pivots: the main tableau is what the phases driven one by one (mi355x_tab_solve on the
    # artificial tableau, mi355x_two_phase_handover, mi355x_tab_solve with a cap on the main one -- each
    # pinned against the oracle elsewhere) leave after exactly that many phase-2 pivots
    a2, m2 = _two_phase_cycling()
    k = ctypes.c_int64(0)
    assert L.mi355x_tab_solve(a2._h, 0, 1024.0, 0, ctypes.byref(k)) == lp.capi.MI_OPTIMAL
    lp.capi.check(L.mi355x_two_phase_handover(a2._h, m2._h, 1024.0, ctypes.byref(k)), "handover")
    assert L.mi355x_tab_solve(m2._h, 1, 1024.0, int(npv[1]), ctypes.byref(k)) == lp.capi.MI_MAX_PIVOTS
    a2._touch(); m2._touch()
    assert np.array_equal(main.matrix.view(np.int64), m2.matrix.view(np.int64))
    assert np.array_equal(main.basis_columns, m2.basis_columns)
    # the request ended with the call: neither handle keeps a stale flag
    k = ctypes.c_int64(0)
    assert L.mi355x_tab_solve(hm, 1, 1024.0, 12, ctypes.byref(k)) == lp.capi.MI_MAX_PIVOTS and k.value == 12
    assert L.mi355x_tab_solve(ha, 0, 1024.0, 12, ctypes.byref(k)) == lp.capi.MI_OPTIMAL and k.value == 0


@pytest.mark.parametrize("which", ["art", "main", "both"])
def test_two_phase_cancel_requested_before_the_call(which):
    """A request with no solve in flight is aimed at the next one -- on either handle of the pair.
    The call ends with MI_CANCELLED before phase 2 makes a pivot (at the latest between the phases),
    and a second call runs to the oracle's optimum."""
    from tests.helpers import random_mixed_problem
    L = lp.capi.lib()
    p = random_mixed_problem(lp, 80, 30, 20, 10, 5)
    art, main = lp.build_tableau(p, p)
    if which in ("art", "both"):
        L.mi355x_tab_cancel(art._h)
    if which in ("main", "both"):
        L.mi355x_tab_cancel(main._h)
    npv = (ctypes.c_int64 * 2)()
    rc = L.mi355x_solve_two_phase(art._h, main._h, 1, 1024.0, npv)
    assert rc == lp.capi.MI_CANCELLED and npv[1] == 0, (rc, list(npv))
    # nothing is left over: fresh tableaux of the same problem solve to the end, bit for bit
    art2, main2 = lp.build_tableau(p, p)
    lp.n_solve_tableau([art2, main2])
    sol = lp.NativeProblem(p).solve()
    assert lp.solution_objective_value(main2) == sol.objective_value()
    # ... and the cancelled pair itself carries on: phase 1 is (or gets) finished, the hand-over and
    # phase 2 follow, ending in the same optimum
    k = ctypes.c_int64(0)
    rc1 = L.mi355x_tab_solve(art._h, 0, 1024.0, 0, ctypes.byref(k))
    assert rc1 == lp.capi.MI_OPTIMAL
    lp.capi.check(L.mi355x_two_phase_handover(art._h, main._h, 1024.0, ctypes.byref(k)), "handover")
    assert L.mi355x_tab_solve(main._h, 1, 1024.0, 0, ctypes.byref(k)) == lp.capi.MI_OPTIMAL
    art._touch(); main._touch()
    assert np.array_equal(main.matrix.view(np.int64), main2.matrix.view(np.int64))
    assert np.array_equal(main.basis_columns, main2.basis_columns)
