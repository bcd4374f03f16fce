"""The NATIVE route of the hook: the Lisp glue's `solve-natively` --
marshal the parsed problem through mi355x_problem_*, mi355x_simplex_solver_begin, ..._step in
bounded chunks, ..._finish, then the four solution-* generics on the light solution object --
replayed call for call through ctypes (NativeProblem.solve_in_chunks is that sequence) on every
golden case of the reference, and compared with the BUILD-TABLEAU route bit for bit: objective,
every variable, every reduced cost, the errors of the read-back and the three conditions."""
import ctypes
import threading
import time

import numpy as np
import pytest

import oracle
from tests import goldens
from tests.helpers import lp_amd, random_mixed_problem

lp = lp_amd()
pytestmark = pytest.mark.gpu

ANSWER_CASES = ["basic", "free_x", "free_x_negative", "ub_only_x", "lb_x", "range_y",
                "free_z_reduced_cost", "widgets", "excessive_constraints", "numerical_issue",
                "variable_bounds_bug", "variable_bounds_only", "equality", "geq"]


@pytest.fixture(scope="module")
def golden():
    return goldens.load()


def _float_problem(case):
    """The case's problem as a double-float LP (what a Lisp caller writing 2d0 / 8d0 hands over)."""
    f32 = bool(case.get("float32_literals"))
    p = case["problem"]
    fl = lambda x: float(goldens.frac(x, f32))                                          # noqa: E731
    return lp.Problem(type=p["type"], vars=list(p["vars"]), objective_var=p.get("objective_var"),
                      objective_func=[(v, fl(c)) for v, c in p["objective"]],
                      var_bounds=[(b[0], (None if b[1] is None else fl(b[1]), None if b[2] is None else fl(b[2])))
                                  for b in p["bounds"]],
                      constraints=[(op, [(v, fl(c)) for v, c in e], fl(r)) for op, e, r in p["constraints"]])


def _outcome(fn):
    try:
        return ("value", fn())
    except (KeyError, ValueError) as e:
        return (type(e).__name__, str(e))


@pytest.mark.parametrize("chunk", [None, 1, 2])
@pytest.mark.parametrize("name", ANSWER_CASES)
def test_native_route_equals_the_build_tableau_route_bit_for_bit(golden, name, chunk):
    """Default dispatch: a double-float LP takes the native route and comes back as the glue's
    MI355X-SOLUTION; `native=False` is the build-tableau route.  chunk = 1 / 2: the job is stepped
    one / two pivots per foreign call, across the phases of the two-phase goldens."""
    case = golden["cases"][name]
    p = _float_problem(case)
    nat = lp.solve_problem(p, chunk=chunk)
    tab = lp.solve_problem(p, native=False, chunk=chunk)
    assert isinstance(nat, lp.NativeSolution) and isinstance(tab, lp.Tableau)
    assert lp.solution_problem(nat) is p and lp.solution_problem(tab) is p
    assert lp.solution_objective_value(nat) == lp.solution_objective_value(tab)
    assert np.float64(lp.solution_objective_value(nat)).view(np.int64) == np.float64(lp.solution_objective_value(tab)).view(np.int64)
    for v in [p.objective_var] + list(p.vars) + ["no-such-variable"]:
        a, b = _outcome(lambda: lp.solution_variable(nat, v)), _outcome(lambda: lp.solution_variable(tab, v))
        assert a == b, (v, a, b)
        if a[0] == "value":
            assert np.float64(a[1]).view(np.int64) == np.float64(b[1]).view(np.int64), v
    for v in list(p.vars) + ["no-such-variable"]:
        a, b = _outcome(lambda: lp.solution_reduced_cost(nat, v)), _outcome(lambda: lp.solution_reduced_cost(tab, v))
        assert a == b, (v, a, b)            # incl. "<v> has no lower bound" (src/simplex.lisp:117-118)
    n1, n2 = nat.pivots()
    assert (n1, n2) == (tuple(tab.n_pivots) if isinstance(tab.n_pivots, tuple) else (0, tab.n_pivots))
    # the reference's known answers (t/simplex.lisp:309-389, t/solver.lisp:20-32, t/integration.lisp)
    f32 = bool(case.get("float32_literals"))
    for v, e in case.get("variables", {}).items():
        assert abs(lp.solution_variable(nat, v) - float(goldens.frac(e, f32))) <= 1e-10 * max(1.0, abs(float(goldens.frac(e, f32)))), v
    values, reduced_cost = lp.with_solution_variables(p, nat)               # with-solution-variables, src/solver.lisp:96-115
    assert values[p.objective_var] == lp.solution_objective_value(nat)


def test_the_three_conditions_on_both_routes(golden):
    """unbounded-problem-error, infeasible-problem-error (src/conditions.lisp:43-60) and
    unsupported-constraint-error (69-77) come out of the native route as out of the other one."""
    for name, err in (("unbounded", lp.UnboundedProblemError), ("infeasible", lp.InfeasibleProblemError)):
        p = _float_problem(golden["cases"][name])
        for kw in ({}, {"native": False}, {"chunk": 1}):
            with pytest.raises(err):
                lp.solve_problem(p, **kw)
    p = _float_problem(golden["cases"]["basic"])
    p.integer_vars = ["x"]
    for kw in ({}, {"native": False}):
        with pytest.raises(lp.UnsupportedConstraintError) as e:
            lp.solve_problem(p, **kw)
        assert e.value.solver_name == "mi355x-simplex"
    # the library itself declines too (the glue's marshal-problem sends the integer variables)
    job = ctypes.c_void_p()
    npb = lp.NativeProblem(p)
    assert lp.capi.lib().mi355x_simplex_solver_begin(npb._h, 1024.0, 0, ctypes.byref(job)) == lp.capi.MI_UNSUPPORTED
    assert not job.value
    # the unbounded no-constraint special case is decided while building (src/simplex.lisp:170,174)
    q = lp.Problem(type="max", vars=["x"], objective_var="w", objective_func=[("x", 1.0)])
    for kw in ({}, {"native": False}):
        with pytest.raises(lp.UnboundedProblemError):
            lp.solve_problem(q, **kw)


def test_dispatch_rule_follows_the_numbers(golden):
    """:native :auto -- doubles and integers a double holds exactly go native; ratios (Fractions
    here) are combined exactly by build-tableau before they are rounded, so they keep that route;
    :full-tableau and :devices ask for the tableau itself."""
    from fractions import Fraction
    case = golden["cases"]["lb_x"]
    pf = _float_problem(case)
    pr = lp.Problem.from_dict(goldens.problem_dict(case))
    assert any(isinstance(c, Fraction) for _, e, _ in pr.constraints for _, c in e)
    assert isinstance(lp.solve_problem(pf), lp.NativeSolution)
    assert isinstance(lp.solve_problem(pr), lp.Tableau)
    assert isinstance(lp.solve_problem(pr, native=True), lp.NativeSolution)
    assert isinstance(lp.solve_problem(pf, full_tableau=True), lp.Tableau)
    assert isinstance(lp.solve_problem(pf, devices=2), lp.Tableau)
    pi = lp.Problem(type="max", vars=["x", "y", "z"], objective_var="w",          # README.md:43-47, integers
                    objective_func=[("x", 1), ("y", 4), ("z", 3)],
                    constraints=[("<=", [("x", 2), ("y", 1)], 8), ("<=", [("y", 1), ("z", 1)], 7)])
    s = lp.solve_problem(pi)
    assert isinstance(s, lp.NativeSolution)
    assert [lp.solution_variable(s, v) for v in ("w", "x", "y", "z")] == [28.5, 0.5, 7.0, 0.0]
    assert [lp.solution_reduced_cost(s, v) for v in ("x", "y", "z")] == [0.0, 0.0, 0.5]
    big = lp.Problem(type="max", vars=["x"], objective_var="w", objective_func=[("x", 2 ** 60 + 1)],
                     constraints=[("<=", [("x", 1)], 1)])
    assert isinstance(lp.solve_problem(big), lp.Tableau)


@pytest.mark.parametrize("n,mle,mge,meq,seed", [(12, 5, 3, 2, 77), (60, 20, 15, 8, 5), (80, 30, 20, 10, 5)])
def test_stepped_job_takes_the_pivots_of_one_long_call(n, mle, mge, meq, seed):
    """A two-phase job stepped with caps that fall inside phase 1, on the hand-over and inside phase
    2 ends in the bits of the one-call solver and of the oracle; the pivot counts add up."""
    L = lp.capi.lib()
    p = random_mixed_problem(lp, n, mle, mge, meq, seed)
    ref = lp.NativeProblem(p).solve()
    r1, r2 = ref.pivots()
    assert r1 > 0 and r2 > 0
    for caps in ([1] * 10000, [r1 - 1, 1, 1, 10 ** 6], [r1, 10 ** 6], [r1 + 1, 10 ** 6], [max(r1 // 2, 1)] * 10000):
        npb = lp.NativeProblem(p)
        job, s, k = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(0)
        lp.capi.check(L.mi355x_simplex_solver_begin(npb._h, 1024.0, 0, ctypes.byref(job)), "begin")
        total, calls = 0, 0
        for cap in caps:
            rc = L.mi355x_simplex_solver_step(job, cap, ctypes.byref(k))
            total += k.value
            calls += 1
            assert k.value <= cap
            if rc != lp.capi.MI_MAX_PIVOTS:
                break
        assert rc == lp.capi.MI_OPTIMAL and total == r1 + r2, (caps[:4], rc, total, r1, r2)
        assert L.mi355x_simplex_solver_step(job, 5, ctypes.byref(k)) == lp.capi.MI_OPTIMAL and k.value == 0   # done stays done
        lp.capi.check(L.mi355x_simplex_solver_finish(job, ctypes.byref(s)), "finish")
        sol = lp.NativeSolution(npb, s)
        assert sol.pivots() == (r1, r2)
        assert sol.objective_value() == ref.objective_value()
        for v in p.vars:
            assert sol.variable(v) == ref.variable(v)
    # finish before the end is refused and consumes the job; abandon releases one mid-way
    npb = lp.NativeProblem(p)
    job, s = ctypes.c_void_p(), ctypes.c_void_p()
    lp.capi.check(L.mi355x_simplex_solver_begin(npb._h, 1024.0, 0, ctypes.byref(job)), "begin")
    assert L.mi355x_simplex_solver_step(job, 1, None) == lp.capi.MI_MAX_PIVOTS
    assert L.mi355x_simplex_solver_finish(job, ctypes.byref(s)) == lp.capi.MI_BAD_ARG and not s.value
    lp.capi.check(L.mi355x_simplex_solver_begin(npb._h, 1024.0, 0, ctypes.byref(job)), "begin")
    L.mi355x_simplex_solver_abandon(job)
    L.mi355x_simplex_solver_abandon(None)
    assert L.mi355x_simplex_solver_step(None, 1, None) == lp.capi.MI_BAD_ARG


def test_max_pivots_is_honoured_on_the_native_route():
    p = random_mixed_problem(lp, 60, 20, 15, 8, 5)
    r1, r2 = lp.NativeProblem(p).solve().pivots()
    for cap in (1, r1, r1 + 1, r1 + r2 - 1):
        with pytest.raises(lp.SolverError, match="pivot cap reached"):
            lp.solve_problem(p, max_pivots=cap)
        with pytest.raises(lp.SolverError, match="pivot cap reached"):
            lp.solve_problem(p, max_pivots=cap, native=False)
    assert lp.solve_problem(p, max_pivots=r1 + r2 + 1).pivots() == (r1, r2)
    assert tuple(lp.solve_problem(p, max_pivots=r1 + r2 + 1, native=False).n_pivots) == (r1, r2)


def test_a_cycling_job_is_cancelled_from_another_thread():
    """Beale's LP cycles exactly in f64 (the reference has no anti-cycling rule, src/simplex.lisp:453-461):
    an uncapped step never returns by itself; mi355x_simplex_solver_cancel from a second thread ends
    it with MI_CANCELLED, the job can be stepped further (bounded) and abandoned."""
    L = lp.capi.lib()
    names = ["x1", "x2", "x3", "x4"]
    beale = lp.Problem(type="max", vars=names, objective_var="z",
                       objective_func=[("x1", 0.75), ("x2", -20.0), ("x3", 0.5), ("x4", -6.0)],     # dyadic: tests/test_gpu_cancel.py
                       constraints=[("<=", [("x1", 0.25), ("x2", -8.0), ("x3", -1.0), ("x4", 9.0)], 0.0),
                                    ("<=", [("x1", 0.5), ("x2", -12.0), ("x3", -0.5), ("x4", 3.0)], 0.0),
                                    ("<=", [("x3", 1.0)], 1.0)])
    npb = lp.NativeProblem(beale)
    job, k = ctypes.c_void_p(), ctypes.c_int64(0)
    lp.capi.check(L.mi355x_simplex_solver_begin(npb._h, 1024.0, 0, ctypes.byref(job)), "begin")
    assert L.mi355x_simplex_solver_step(job, 600, ctypes.byref(k)) == lp.capi.MI_MAX_PIVOTS and k.value == 600   # it does cycle
    out = {}

    def run():
        out["rc"] = L.mi355x_simplex_solver_step(job, 0, ctypes.byref(k))
    th = threading.Thread(target=run)
    th.start()
    time.sleep(0.3)
    assert th.is_alive()
    lp.capi.check(L.mi355x_simplex_solver_cancel(job), "cancel")
    th.join(timeout=20)
    assert not th.is_alive() and out["rc"] == lp.capi.MI_CANCELLED and k.value > 0
    # the request ended with that step: the next (bounded) step runs its full cap
    assert L.mi355x_simplex_solver_step(job, 64, ctypes.byref(k)) == lp.capi.MI_MAX_PIVOTS and k.value == 64
    L.mi355x_simplex_solver_abandon(job)


def test_config2_sized_problem_native_vs_oracle():
    """A dense 1024 x 512 LP as a `problem` (not a tableau) through the default hook: never a dense
    tableau on the host side, the answer in the bits of the oracle's solve of build-tableau's matrix."""
    n, m = 1024, 512
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(2, 7))
    names = ["x%d" % i for i in range(n)]
    cons = [("<=", list(zip(names, M0[i, :n].tolist())), float(M0[i, -1])) for i in range(m)]
    p = lp.Problem(type="max", vars=names, objective_var="obj",
                   objective_func=list(zip(names, (-M0[m, :n]).tolist())), constraints=cons)
    M, b = M0.copy(), b0.copy()
    st, npiv, _ = oracle.solve(M, b)
    assert st == oracle.OPTIMAL
    s = lp.solve_problem(p)
    assert isinstance(s, lp.NativeSolution) and s.pivots() == (0, npiv)
    assert np.float64(lp.solution_objective_value(s)).view(np.int64) == M[m, -1:].view(np.int64)[0]
    for j, v in enumerate(names):
        rows = np.nonzero(b == j)[0]
        want = M[rows[0], -1] if len(rows) else 0.0
        assert lp.solution_variable(s, v) == want, v
        assert lp.solution_reduced_cost(s, v) == M[m, j], v
