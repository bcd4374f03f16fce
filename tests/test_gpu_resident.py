"""The resident solve (k_resident: the stored tableau [non-basic columns | RHS] in registers, split
into column strips over up to 32 workgroups per LP, ONE exchange per pivot) against the oracle:
every strip shape (row slots x strip width), ragged last strips, both senses, tolerance factors,
pivot caps and resumption, asynchronous requests, batches, the two ways out (a non-finite column
-> dense path, workgroups that are not co-resident -> established paths) -- bit for bit."""
import ctypes

import numpy as np
import pytest

import oracle
from tests.helpers import lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _solve(M0, b0, is_max=True, factor=1024.0, cap=0, expect_resident=True):
    L = lp.capi.lib()
    n_m = M0.shape[1] - 1
    t = lp.Tableau(None, lp.Problem(type="max" if is_max else "min"), M0, b0, n_m, M0.shape[0] - 1, {})
    k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(t._h, int(is_max), factor, 0, 1), "prepare")   # -> compact representation
    L.mi355x_tab_sync(t._h, ctypes.byref(k))
    assert L.mi355x_tab_resident(t._h) == int(expect_resident)
    rc = L.mi355x_tab_solve(t._h, int(is_max), factor, cap, ctypes.byref(k))
    t._touch()
    return t, rc, int(k.value)


def _check(t, rc, k, M0, b0, is_max=True, factor=1024.0, cap=0):
    M, b = M0.copy(), b0.copy()
    with np.errstate(all="ignore"):
        so, no, trace = oracle.solve(M, b, is_max=is_max, factor=factor, max_pivots=cap, trace_cap=1 << 14)
    assert (rc, k) == (so, no)
    assert np.array_equal(t.pivot_trace()[:no], trace)
    G = t.matrix
    nan_o, nan_g = np.isnan(M), np.isnan(G)
    assert np.array_equal(nan_o, nan_g)
    assert np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64))
    assert np.array_equal(t.basis_columns, b)
    return so, no


# (n, m): row slots 1 / 2 / 4 (m <= 256 / 512 / 1024), strips 1 .. 32, ragged last strips
SHAPES = [(1, 1), (5, 3), (63, 10), (64, 30), (65, 30), (129, 255), (300, 256), (300, 257), (33, 500),
          (1024, 512), (500, 513), (511, 1000), (512, 1024), (2048, 200), (2047, 256)]


@pytest.fixture(params=[0, 1], ids=["strip-in-registers", "strip-part-in-lds"])
def strip_store(request):
    """Round 4: for shapes of <= 256 constraints the last 24 of a strip's 64 columns can live in LDS
    (three workgroups per CU instead of two: measured not faster, kept behind the tuning knob).
    1 = use it for every such shape."""
    L = lp.capi.lib()
    assert L.mi355x_tune_set_resident_lds(request.param) == request.param
    yield request.param
    L.mi355x_tune_set_resident_lds(0)


@pytest.mark.parametrize("n,m", SHAPES)
def test_resident_solve_bitwise(n, m, strip_store):
    if strip_store and m > 256:
        pytest.skip("one row per thread only")
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(2, 7 * n + m))
    t, rc, k = _solve(M0, b0)
    so, no = _check(t, rc, k, M0, b0)
    assert so == oracle.OPTIMAL


def test_resident_is_chosen_by_shape_and_by_knobs():
    """Resident when the stored tableau fits (constraints <= 1024, <= 32 strips) and every
    implementation knob is at its default; an explicit knob selects the path it names; mode 1
    switches it off, mode 2 keeps it on whatever the other knobs say."""
    L = lp.capi.lib()
    M0, b0 = lp.synth.tableau(300, 100, 5)

    def resident():
        t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, 400, 100, {})
        k = ctypes.c_int64(0)
        L.mi355x_tab_solve_async(t._h, 1, 1024.0, 0, 1)
        L.mi355x_tab_sync(t._h, ctypes.byref(k))
        return L.mi355x_tab_resident(t._h)
    try:
        assert resident() == 1
        for setter, value, default in ((L.mi355x_tune_set_block, 8, 16), (L.mi355x_tune_set_lookahead_mode, 1, 0),
                                       (L.mi355x_tune_set_select_mode, 2, 0), (L.mi355x_tune_set_resident, 1, 0)):
            setter(value)
            assert resident() == 0
            setter(default)
        L.mi355x_tune_set_resident(2)
        L.mi355x_tune_set_block(4)
        assert resident() == 1
    finally:
        L.mi355x_tune_set_resident(0)
        L.mi355x_tune_set_block(0)
    for n, m in ((8192, 4096), (100, 1025), (2100, 200)):       # too many constraints / strips
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, 1, 0, -1, 0), "create")
        k = ctypes.c_int64(0)
        L.mi355x_tab_solve_async(h, 1, 1024.0, 0, 1)
        L.mi355x_tab_sync(h, ctypes.byref(k))
        assert L.mi355x_tab_resident(h) == 0
        L.mi355x_tab_destroy(h)


@pytest.mark.parametrize("kind", ["max", "min"])
@pytest.mark.parametrize("factor", [16.0, 1024.0, float(2 ** 20)])
def test_resident_senses_tolerances_and_degeneracy(kind, factor, strip_store):
    rng = np.random.default_rng(17)
    for n, m, degenerate in ((90, 40, True), (300, 120, False), (700, 300, True)):
        if degenerate:
            A = rng.integers(0, 4, (m, n)).astype(np.float64)
            b = rng.integers(0, 5, m).astype(np.float64)
            c = rng.integers(-2, 5, n).astype(np.float64)
        else:
            A = rng.uniform(-0.5, 1.5, (m, n)); b = rng.uniform(0.5, 5.0, m); c = rng.uniform(-0.5, 2.0, n)
        M0 = np.zeros((m + 1, n + m + 1))
        M0[:m, :n] = A; M0[np.arange(m), n + np.arange(m)] = 1.0; M0[:m, -1] = b
        M0[m, :n] = -c if kind == "max" else c
        b0 = np.arange(n, n + m, dtype=np.int64)
        t, rc, k = _solve(M0, b0, is_max=(kind == "max"), factor=factor, cap=400)
        _check(t, rc, k, M0, b0, is_max=(kind == "max"), factor=factor, cap=400)


@pytest.mark.parametrize("n,m", [(1024, 512), (1500, 250)])
def test_resident_caps_resume_and_async_requests(n, m, strip_store):
    """A capped solve stops where the oracle stops; solving on continues from there; sequences of
    solve_async(n) requests (each ONE launch that keeps the tableau on chip) add up to the same."""
    L = lp.capi.lib()
    seed = lp.synth.seed_for(2, 3)
    M0, b0 = lp.synth.tableau(n, m, seed)
    t, rc, k = _solve(M0, b0, cap=37)
    _check(t, rc, k, M0, b0, cap=37)
    kk = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(t._h, 1, 1024.0, 0, ctypes.byref(kk))
    t._touch()
    M, b = M0.copy(), b0.copy()
    so, no, trace = oracle.solve(M, b, trace_cap=1 << 14)
    assert rc == so and kk.value == no - 37                     # pivots of THIS call
    assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(t.pivot_trace()[:no], trace)
    # asynchronous requests
    t2 = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    done, first = 0, 1
    for req in (1, 16, 5, 100, 3):
        lp.capi.check(L.mi355x_tab_solve_async(t2._h, 1, 1024.0, req, first), "solve_async")
        first = 0
        rc = L.mi355x_tab_sync(t2._h, ctypes.byref(kk))
        done += req
        assert rc == lp.capi.MI_RUNNING and kk.value == done
    t2._touch()
    M, b = M0.copy(), b0.copy()
    oracle.solve(M, b, max_pivots=done)
    assert np.array_equal(t2.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(t2.basis_columns, b)
    assert np.array_equal(t2.pivot_trace()[:done], trace[:done])
    lp.capi.check(L.mi355x_tab_solve_async(t2._h, 1, 1024.0, 5000, 0), "solve_async")   # far more than the LP needs
    rc = L.mi355x_tab_sync(t2._h, ctypes.byref(kk))
    assert rc == so and kk.value == no


def test_resident_hands_non_finite_columns_to_the_dense_path(strip_store):
    """Entries over hundreds of orders of magnitude: the resident solve stops at the first entering
    column it cannot follow on the compact representation (kNeedDense), writes the tableau back as
    it stands, and the dense per-pivot path takes over -- NaNs in the oracle's places."""
    found = 0
    for seed in range(30):
        rng = np.random.default_rng(seed)
        n, m = 48, 20
        mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(-300, 161, shape)   # noqa: E731
        M0 = np.zeros((m + 1, n + m + 1))
        M0[:m, :n] = mag((m, n)) * rng.choice([1.0, 1.0, -1.0], (m, n))
        M0[np.arange(m), n + np.arange(m)] = 1.0
        M0[:m, -1] = mag(m)
        M0[m, :n] = -mag(n)
        b0 = np.arange(n, n + m, dtype=np.int64)
        t, rc, k = _solve(M0, b0, cap=60)
        _check(t, rc, k, M0, b0, cap=60)
        found += bool(np.isnan(t.matrix).any())
    assert found >= 3, "no overflowing LP among the seeds"


def test_resident_workgroups_not_co_resident_fall_back(hooks_lib):
    """Test hook: the last workgroup of the LP never publishes its first record (what a workgroup
    that is not resident looks like).  Everybody gives up at the FIRST exchange, nothing has been
    modified, the handle continues (and stays) on the established paths: oracle's pivots and bits."""
    L = hooks_lib
    n, m = 700, 300
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(2, 99))
    try:
        L.mi355x_tune_set_la_max_spins(20000)
        L.mi355x_tune_set_resident_fault(1)
        t, rc, k = _solve(M0, b0)
    finally:
        L.mi355x_tune_set_la_max_spins(0)
        L.mi355x_tune_set_resident_fault(0)
    _check(t, rc, k, M0, b0)
    assert L.mi355x_tab_resident(t._h) == 0
    # ... also through the asynchronous entry points: the request is cut short, not lost
    t2 = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    kk = ctypes.c_int64(0)
    try:
        L.mi355x_tune_set_la_max_spins(20000)
        L.mi355x_tune_set_resident_fault(1)
        lp.capi.check(L.mi355x_tab_solve_async(t2._h, 1, 1024.0, 40, 1), "solve_async")
        rc = L.mi355x_tab_sync(t2._h, ctypes.byref(kk))
    finally:
        L.mi355x_tune_set_la_max_spins(0)
        L.mi355x_tune_set_resident_fault(0)
    assert rc == lp.capi.MI_RUNNING and kk.value == 0 and L.mi355x_tab_resident(t2._h) == 0
    lp.capi.check(L.mi355x_tab_solve_async(t2._h, 1, 1024.0, 40, 0), "solve_async")
    rc = L.mi355x_tab_sync(t2._h, ctypes.byref(kk))
    t2._touch()
    M, b = M0.copy(), b0.copy()
    oracle.solve(M, b, max_pivots=40)
    assert (rc, kk.value) == (lp.capi.MI_RUNNING, 40)
    assert np.array_equal(t2.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(t2.basis_columns, b)


@pytest.mark.parametrize("n,m,nl", [(512, 256, 40), (60, 30, 19), (300, 40, 9), (700, 300, 11), (130, 600, 5)])
def test_resident_batches_every_lp_vs_oracle(n, m, nl, strip_store):
    """Batches: every LP on chip with its own group of workgroups, all LPs in ONE launch, each
    progressing and finishing on its own (78 .. 199 pivots per LP at the config-4 shape)."""
    seeds = np.array([lp.synth.seed_for(4, 1000 + 3 * k) for k in range(nl)], dtype=np.uint64)
    batch = lp.TableauBatch.synthetic(nl, n, m, seeds)
    st, npv = batch.solve()
    pivots = set()
    for k in range(nl):
        M, b = lp.synth.tableau(n, m, int(seeds[k]))
        so, no, _ = oracle.solve(M, b)
        G, gb = batch.download(k)
        assert (int(st[k]), int(npv[k])) == (so, no), k
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(gb, b), k
        pivots.add(no)
    assert nl < 10 or len(pivots) > 3
    # capped: every LP stops at the cap (or earlier, optimal)
    batch = lp.TableauBatch.synthetic(nl, n, m, seeds)
    st, npv = batch.solve(max_pivots=9)
    for k in range(nl):
        M, b = lp.synth.tableau(n, m, int(seeds[k]))
        so, no, _ = oracle.solve(M, b, max_pivots=9)
        G, gb = batch.download(k)
        assert (int(st[k]), int(npv[k])) == (so, no), k
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(gb, b), k


def test_resident_batch_with_a_lost_member_and_an_overflowing_member(hooks_lib):
    """One launch, three kinds of LPs: ordinary ones (finished on chip), one whose entries overflow
    (kNeedDense: the batch continues on the dense tableaux) -- and, separately, the co-residency
    test hook (every LP's last workgroup mute): all end where the oracle ends."""
    L = hooks_lib
    n, m, nl = 48, 20, 6
    rng = np.random.default_rng(4)
    Ms, Bs = [], []
    for k in range(nl):
        if k == 3:
            mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(-300, 161, shape)   # noqa: E731
            M0 = np.zeros((m + 1, n + m + 1))
            M0[:m, :n] = mag((m, n)) * rng.choice([1.0, 1.0, -1.0], (m, n))
            M0[np.arange(m), n + np.arange(m)] = 1.0
            M0[:m, -1] = mag(m); M0[m, :n] = -mag(n)
            b0 = np.arange(n, n + m, dtype=np.int64)
        else:
            M0, b0 = lp.synth.tableau(n, m, 50 + k)
        Ms.append(M0); Bs.append(b0)
    for fault in (0, 1):
        try:
            L.mi355x_tune_set_la_max_spins(20000)
            L.mi355x_tune_set_resident_fault(fault)
            batch = lp.TableauBatch.from_arrays(np.stack(Ms), np.stack(Bs))
            st, npv = batch.solve(max_pivots=60)
        finally:
            L.mi355x_tune_set_la_max_spins(0)
            L.mi355x_tune_set_resident_fault(0)
        for k in range(nl):
            M, b = Ms[k].copy(), Bs[k].copy()
            with np.errstate(all="ignore"):
                so, no, _ = oracle.solve(M, b, max_pivots=60)
            G, gb = batch.download(k)
            assert (int(st[k]), int(npv[k])) == (so, no), (fault, k)
            nan_o, nan_g = np.isnan(M), np.isnan(G)
            assert np.array_equal(nan_o, nan_g), (fault, k)
            assert np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64)) and np.array_equal(gb, b), (fault, k)
