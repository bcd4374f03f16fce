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
        one = lp.solve_problem(p)                                              # the one-problem hook
        assert np.array_equal(one.matrix.view(np.int64), r.matrix.view(np.int64)), k
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
