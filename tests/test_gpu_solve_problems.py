"""`mi355x-solve-problems` (lisp/mi355x-simplex.lisp), exercised through its Python mirror, which
makes exactly the glue's call sequence: build-tableau per member -> the single-phase members grouped
by shape and sense, every group of two or more as ONE multi-device batch (mi355x_multibatch_create /
_solve in bounded chunks / _download per member / _destroy) -> two-phase members
(src/simplex.lisp:402-452), integer problems and singletons through the one-problem hook
(src/solver.lisp:53-56).  Every member must end exactly where (solve-problem member) ends -- and
where the oracle ends."""
import numpy as np
import pytest

import oracle
from tests import goldens
from tests.helpers import lp_amd, random_mixed_problem

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _golden_problem(golden, name):
    return lp.Problem.from_dict(goldens.problem_dict(golden["cases"][name]))


def _random_le_problem(n, m, seed, kind="max"):
    rng = np.random.default_rng(seed)
    names = ["v%d" % i for i in range(n)]
    A = rng.uniform(0.1, 1.0, (m, n))
    cons = [("<=", list(zip(names, A[i].tolist())), float(rng.uniform(5, 9))) for i in range(m)]
    c = rng.uniform(0.5, 1.5, n) * (1.0 if kind == "max" else -1.0)
    return lp.Problem(type=kind, vars=names, objective_var="obj",
                      objective_func=list(zip(names, c.tolist())), constraints=cons)


def _oracle_outcome(p):
    """(status, final matrix, basis) of the oracle on the f64 tableau(x) build-tableau produces."""
    tabs = lp.build_tableau(p, p)
    if isinstance(tabs, list):
        art, main = tabs
        A, ab = art.matrix.copy(), art.basis_columns.copy()
        Mm, mb = main.matrix.copy(), main.basis_columns.copy()
        st, _ = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max, factor=main.fp_tolerance_factor)
        return st, Mm, mb
    M, b = tabs.matrix.copy(), tabs.basis_columns.copy()
    st, _, _ = oracle.solve(M, b, is_max=tabs.is_max, factor=tabs.fp_tolerance_factor)
    return st, M, b


def _mixed_list(golden):
    ps = [_golden_problem(golden, "basic")]                                   # README LP, t/simplex.lisp:170-194
    ps += [_random_le_problem(12, 7, s) for s in range(5)]                     # one shape, max: a batch of 5
    ps += [_golden_problem(golden, "equality"), _golden_problem(golden, "geq")]   # two-phase, t/simplex.lisp:196-275
    ps += [_random_le_problem(12, 7, 10 + s, kind="min") for s in range(3)]    # same shape, min: its own batch
    ps += [_random_le_problem(30, 11, 20 + s) for s in range(2)]               # another shape: a batch of 2
    ps += [random_mixed_problem(lp, 9, 4, 2, 1, 31)]                           # random two-phase
    ps += [_golden_problem(golden, "unbounded"), _golden_problem(golden, "infeasible")]
    ps += [_random_le_problem(17, 5, 40)]                                      # alone in its group
    ps += [_golden_problem(golden, "basic")]                                   # a second README LP: batch with the first
    return ps


@pytest.mark.parametrize("devices", [1, 3])
def test_mixed_list_every_member_vs_single_solves_and_the_oracle(golden, devices):
    ps = _mixed_list(golden)
    got = lp.solve_problems(ps, devices=devices, errorp=False)
    assert len(got) == len(ps)
    n_batched = 0
    for k, (p, r) in enumerate(zip(ps, got)):
        st, M, b = _oracle_outcome(p)
        if st == oracle.UNBOUNDED:
            assert isinstance(r, lp.UnboundedProblemError), k
            continue
        if st == oracle.INFEASIBLE:
            assert isinstance(r, lp.InfeasibleProblemError), k
            continue
        assert st == oracle.OPTIMAL and isinstance(r, lp.Tableau), (k, r)
        assert np.array_equal(r.matrix.view(np.int64), M.view(np.int64)), k
        assert np.array_equal(r.basis_columns, b), k
        one = lp.solve_problem(p, native=False)                                # the one-problem hook, build-tableau route
        assert np.array_equal(one.matrix.view(np.int64), r.matrix.view(np.int64)), k
        nat = lp.solve_problem(p, native=True)                                 # ... and its native route
        assert isinstance(nat, lp.NativeSolution) and lp.solution_objective_value(nat) == lp.solution_objective_value(one)
        assert lp.solution_objective_value(r) == lp.solution_objective_value(one)
        for v in p.vars:
            assert lp.solution_variable(r, v) == lp.solution_variable(one, v), (k, v)
        n_batched += 1
    assert n_batched >= 14
    # the goldens' known answers through the list entry (t/simplex.lisp:190, 233; README.md:58-62)
    assert lp.solution_objective_value(got[0]) == 28.5 and lp.solution_variable(got[0], "x") == 0.5
    assert lp.solution_objective_value(got[6]) == 28.5
    assert abs(lp.solution_objective_value(got[7]) - 85 / 3) <= 1e-10 * 85 / 3


def test_errorp_raises_the_first_failure_after_all_members_ran(golden):
    ps = [_golden_problem(golden, "basic"), _golden_problem(golden, "unbounded"), _golden_problem(golden, "basic")]
    with pytest.raises(lp.UnboundedProblemError):
        lp.solve_problems(ps)
    got = lp.solve_problems(ps, errorp=False)
    assert lp.solution_objective_value(got[0]) == 28.5 and lp.solution_objective_value(got[2]) == 28.5


def test_integer_members_are_declined_alone(golden):
    p = _golden_problem(golden, "basic")
    q = lp.Problem(type="max", vars=["x", "y"], objective_var="w", objective_func=[("x", 1), ("y", 1)],
                   integer_vars=["x"], constraints=[("<=", [("x", 1), ("y", 2)], 4)])
    got = lp.solve_problems([p, q, p], errorp=False)
    assert isinstance(got[1], lp.UnsupportedConstraintError)
    assert lp.solution_objective_value(got[0]) == 28.5 and lp.solution_objective_value(got[2]) == 28.5


def test_config4_shaped_list_in_bounded_chunks():
    """64 LPs of 48 x 24 through the list entry with a pivot cap that forces several chunks per
    member: the capped members stop where the oracle stops under the same cap."""
    ps = [_random_le_problem(48, 24, 100 + s) for s in range(64)]
    got = lp.solve_problems(ps, devices=8, errorp=False)
    for k, (p, r) in enumerate(zip(ps, got)):
        st, M, b = _oracle_outcome(p)
        assert st == oracle.OPTIMAL and np.array_equal(r.matrix.view(np.int64), M.view(np.int64)), k


def _drive_out_problem(seed):
    """Equality rows with right-hand side 0 whose artificial columns cancel in the phase-1 objective:
    phase 1 is optimal at once with artificials still basic -- drive-out pivots (src/simplex.lisp:
    419-434) are needed before the hand-over (seeds searched on the oracle, tests/test_gpu_parity.py)."""
    rng = np.random.default_rng(seed)
    n = 5
    names = ["x%d" % i for i in range(n)]
    rows = [rng.integers(-2, 3, n).astype(float) for _ in range(3)]
    cons = [("=", list(zip(names, a.tolist())), 0.0) for a in rows if a.any()]
    cons.append(("<=", list(zip(names, [1.0] * n)), 5.0))
    return lp.Problem(type="max", vars=names, objective_var="obj",
                      objective_func=list(zip(names, rng.integers(1, 4, n).astype(float).tolist())), constraints=cons)


@pytest.mark.parametrize("devices", [1, 3])
def test_two_phase_members_run_as_batches(golden, devices):
    """Two-phase members of one shape: phase 1, the per-member feasibility test and hand-over
    (src/simplex.lisp:402-452) and phase 2 as a pair of batches (mi355x_multibatch_solve_two_phase);
    members that need drive-out pivots get them inside the batch; infeasible and
    unbounded members keep their conditions.  Every member against the one-problem hook (pivot counts
    of both phases included) and the oracle."""
    ps = [random_mixed_problem(lp, 12, 5, 3, 2, 50 + s) for s in range(7)]            # one shape: a pair of batches of 7
    ps += [_golden_problem(golden, "equality"), _golden_problem(golden, "equality")]     # t/simplex.lisp:196-237, twice
    ps += [_golden_problem(golden, "geq"), _golden_problem(golden, "geq")]               # t/simplex.lisp:239-275
    ps += [_drive_out_problem(s) for s in (282, 957, 959, 1396, 1481, 1610)]             # drive-out pivots inside the batch
    ps += [random_mixed_problem(lp, 12, 5, 3, 2, 70 + s, kind="min") for s in range(3)]  # the same shape, min
    ps += [_golden_problem(golden, "infeasible"), _golden_problem(golden, "infeasible")]
    got = lp.solve_problems(ps, devices=devices, errorp=False)
    solved = 0
    for k, (p, r) in enumerate(zip(ps, got)):
        st, M, b = _oracle_outcome(p)
        if st == oracle.INFEASIBLE:
            assert isinstance(r, lp.InfeasibleProblemError), k
            continue
        if st == oracle.UNBOUNDED:
            assert isinstance(r, lp.UnboundedProblemError), k
            continue
        assert st == oracle.OPTIMAL and isinstance(r, lp.Tableau), (k, r)
        assert np.array_equal(r.matrix.view(np.int64), M.view(np.int64)) and np.array_equal(r.basis_columns, b), k
        one = lp.solve_problem(p, native=False)
        assert np.array_equal(one.matrix.view(np.int64), r.matrix.view(np.int64)), k
        assert tuple(one.n_pivots) == tuple(r.n_pivots), (k, one.n_pivots, r.n_pivots)
        nat = lp.solve_problem(p, native=True)
        assert nat.pivots() == tuple(one.n_pivots) and lp.solution_objective_value(nat) == lp.solution_objective_value(one)
        for v in p.vars:
            assert lp.solution_variable(nat, v) == lp.solution_variable(one, v), (k, v)
        solved += 1
    assert solved >= 18
    assert lp.solution_objective_value(got[7]) == 28.5
    assert abs(lp.solution_objective_value(got[9]) - 85 / 3) <= 1e-10 * 85 / 3


def test_multibatch_two_phase_entry_point_directly(golden):
    """mi355x_multibatch_solve_two_phase / _two_phase_handover through ctypes: statuses and pivot
    counts per member, members that need drive-out pivots (src/simplex.lisp:419-434) solved inside the
    batch, argument checks."""
    import ctypes
    L = lp.capi.lib()
    ps = [_drive_out_problem(s) for s in (282, 957, 959)]
    tabs = [lp.build_tableau(p, p) for p in ps]
    shapes = {(a.matrix.shape, t.matrix.shape) for a, t in tabs}
    assert len(shapes) == 1
    amb = lp.MultiDeviceBatch.from_arrays(np.stack([a.matrix for a, _ in tabs]), np.stack([a.basis_columns for a, _ in tabs]), n_devices=2)
    mmb = lp.MultiDeviceBatch.from_arrays(np.stack([t.matrix for _, t in tabs]), np.stack([t.basis_columns for _, t in tabs]), n_devices=2)
    st, npv = amb.solve_two_phase(mmb)
    # (round 4: such members were declined with MI_UNSUPPORTED) the drive-out pivots of
    # src/simplex.lisp:419-434 now run on the member inside the batch: outcome, pivot counts of both
    # phases and every bit of the solved main tableau as mi355x_solve_two_phase on that problem alone
    for q, p in enumerate(ps):
        one_tabs = lp.build_tableau(p, p)
        try:
            one = lp.n_solve_tableau(one_tabs)
            want = lp.capi.MI_OPTIMAL
        except lp.UnboundedProblemError:
            one, want = one_tabs[1], lp.capi.MI_UNBOUNDED
        assert int(st[q]) == want, (q, st)
        assert tuple(npv[q]) == tuple(one.n_pivots), (q, npv[q], one.n_pivots)
        assert npv[q, 0] >= 1                                   # phase 1 itself made no pivot: these ARE drive-out pivots
        if want == lp.capi.MI_OPTIMAL:
            G, gb = mmb.download(q)
            assert np.array_equal(G.view(np.int64), one.matrix.view(np.int64)) and np.array_equal(gb, one.basis_columns), q
    # the same through the step-by-step form (what the glue drives in bounded chunks)
    amb2 = lp.MultiDeviceBatch.from_arrays(np.stack([a.matrix for a, _ in tabs]), np.stack([a.basis_columns for a, _ in tabs]), n_devices=2)
    mmb2 = lp.MultiDeviceBatch.from_arrays(np.stack([t.matrix for _, t in tabs]), np.stack([t.basis_columns for _, t in tabs]), n_devices=2)
    st1, np1 = amb2.solve(is_max=False)
    between, nd = amb2.two_phase_handover(mmb2, phase1_status=st1)
    st2, np2 = mmb2.solve(is_max=True)
    assert (between == lp.capi.MI_OK).all() and (nd >= 1).all() and np.array_equal(np1 + nd, npv[:, 0]) and np.array_equal(np2, npv[:, 1])
    assert np.array_equal(st2, st)
    for q in range(len(ps)):
        A, _ = mmb.download(q)
        B, _ = mmb2.download(q)
        assert np.array_equal(A.view(np.int64), B.view(np.int64)), q
    # a member whose phase 1 did not end optimal keeps that status and its main tableau is passed over
    p1 = st1.copy(); p1[1] = lp.capi.MI_MAX_PIVOTS
    amb3 = lp.MultiDeviceBatch.from_arrays(np.stack([a.matrix for a, _ in tabs]), np.stack([a.basis_columns for a, _ in tabs]), n_devices=2)
    mmb3 = lp.MultiDeviceBatch.from_arrays(np.stack([t.matrix for _, t in tabs]), np.stack([t.basis_columns for _, t in tabs]), n_devices=2)
    amb3.solve(is_max=False)
    b3, _ = amb3.two_phase_handover(mmb3, phase1_status=p1)
    assert b3[1] == lp.capi.MI_MAX_PIVOTS and b3[0] == lp.capi.MI_OK and b3[2] == lp.capi.MI_OK
    s3, n3 = mmb3.solve(is_max=True)
    assert n3[1] == 0 and s3[1] == lp.capi.MI_OPTIMAL           # all-zero objective row: priced as optimal at once
    other = lp.MultiDeviceBatch.from_arrays(np.stack([t.matrix for _, t in tabs]), np.stack([t.basis_columns for _, t in tabs]), n_devices=1)
    s4 = np.zeros(3, dtype=np.int32)
    assert L.mi355x_multibatch_solve_two_phase(amb._h, other._h, 1, 1024.0, s4.ctypes.data_as(ctypes.c_void_p), None) == lp.capi.MI_BAD_ARG
    assert L.mi355x_multibatch_solve_two_phase(None, mmb._h, 1, 1024.0, s4.ctypes.data_as(ctypes.c_void_p), None) == lp.capi.MI_BAD_ARG


# ---- the whole list behind ONE job of the library (the glue's :native :many) ----------------------
def _same_outcome(a, b, p):
    """Two results of one problem: the same exception type, or solutions equal bit for bit."""
    if isinstance(a, Exception) or isinstance(b, Exception):
        return type(a) is type(b)
    if lp.solution_objective_value(a) != lp.solution_objective_value(b) or a.pivots() != b.pivots():
        return False
    for v in p.vars:
        if lp.solution_variable(a, v) != lp.solution_variable(b, v):
            return False
        ra, rb = [], []
        for s, out in ((a, ra), (b, rb)):
            try:
                out.append(lp.solution_reduced_cost(s, v))
            except ValueError as e:
                out.append(str(e))
        if ra != rb:
            return False
    return True


def _one_native(p):
    try:
        return lp.NativeProblem(p).solve_in_chunks()
    except lp.SolverError as e:
        return e


@pytest.mark.parametrize("chunk", [None, 3])
@pytest.mark.parametrize("devices", [1, 3])
def test_native_list_entry_equals_the_one_problem_job_member_by_member(golden, devices, chunk):
    """mi355x_simplex_solver_many_begin / _step / _finish on a mixed list -- single-phase groups, two-phase
    groups (incl. members that need drive-out pivots), members alone in their group, unbounded,
    infeasible and integer members: every member's outcome is the one-problem job's, bit for bit
    (objective, every variable, every reduced cost, pivot counts of both phases).  chunk = 3: every
    phase of every group is stepped three pivots per foreign call."""
    solve_many = lp.native.solve_many
    ps = _mixed_list(golden)
    ps += [_drive_out_problem(s) for s in (282, 957, 959)]
    ps += [random_mixed_problem(lp, 12, 5, 3, 2, 50 + s) for s in range(4)]
    q = lp.Problem(type="max", vars=["x", "y"], objective_var="w", objective_func=[("x", 1), ("y", 1)],
                   integer_vars=["x"], constraints=[("<=", [("x", 1), ("y", 2)], 4)])
    ps.append(q)
    ps.append(lp.Problem(type="max", vars=["x"], objective_var="w", objective_func=[("x", 1.0)]))      # no constraints: unbounded while building
    got = solve_many(ps, devices=devices, chunk=chunk)
    assert len(got) == len(ps)
    n_solved = 0
    for k, (p, r) in enumerate(zip(ps, got)):
        if p.integer_vars:
            assert isinstance(r, lp.UnsupportedConstraintError), k
            continue
        one = _one_native(p)
        assert _same_outcome(r, one, p), (k, r, one)
        n_solved += not isinstance(r, Exception)
    assert n_solved >= 20
    assert lp.solution_objective_value(got[0]) == 28.5 and lp.solution_variable(got[0], "x") == 0.5    # README.md:58-62
    # through the list entry of the mirror (the glue's (mi355x-solve-problems problems :native :many))
    again = lp.solve_problems(ps, devices=devices, native="many", errorp=False)
    assert all(_same_outcome(a, b, p) for a, b, p in zip(again, got, ps) if not p.integer_vars)
    with pytest.raises(lp.SolverError):
        lp.solve_problems(ps, native="many")                                                        # errorp: the first failure


def test_native_list_entry_config4_shaped():
    """64 LPs of 48 x 24 as ONE job over 8 (logical) sub-batches, stepped in chunks of 7 pivots, against the oracle."""
    solve_many = lp.native.solve_many
    ps = [_random_le_problem(48, 24, 100 + s) for s in range(64)]
    got = solve_many(ps, devices=8, chunk=7)
    for k, (p, r) in enumerate(zip(ps, got)):
        st, M, b = _oracle_outcome(p)
        assert st == oracle.OPTIMAL and isinstance(r, lp.NativeSolution), k
        assert np.float64(lp.solution_objective_value(r)).view(np.int64) == M[-1, -1:].view(np.int64)[0], k
    # a cap that ends the job early leaves the unfinished members at "pivot cap reached"
    capped = solve_many(ps[:8], devices=2, max_pivots=2)
    assert all(isinstance(r, lp.SolverError) and "pivot cap" in str(r) for r in capped)
    # argument checks
    import ctypes
    L = lp.capi.lib()
    job = ctypes.c_void_p()
    assert L.mi355x_simplex_solver_many_begin(None, 1, 1024.0, 1, None, ctypes.byref(job)) == lp.capi.MI_BAD_ARG
    assert L.mi355x_simplex_solver_many_step(None, 0, None) == lp.capi.MI_BAD_ARG
    L.mi355x_simplex_solver_many_abandon(None)
