"""CPU tests of the host-side logic above the C ABI (no GPU needed): build-tableau,
read-back, the synthetic generator, status mapping."""
from fractions import Fraction

import numpy as np
import pytest

from oracle import rational_ref as rr
from tests import goldens
from tests.goldens import fmat
from tests.helpers import lp_amd, random_mixed_problem

lp = lp_amd()


def _problem(case):
    return lp.Problem.from_dict(goldens.problem_dict(case))


@pytest.mark.parametrize("name", ["basic", "equality", "geq"])
def test_build_tableau_golden(golden, name):
    """t/simplex.lisp:60-133 through the product's host-side build_tableau (f64)."""
    case = golden["cases"][name]
    p = _problem(case)
    tabs = lp.build_tableau(p, p)
    main = tabs[1] if isinstance(tabs, list) else tabs
    exp = case["initial"]
    assert isinstance(main, lp.Tableau) and main.problem is p and main.instance_problem is p
    assert np.array_equal(main.matrix, np.array(fmat(exp["matrix"]), dtype=float))
    assert main.basis_columns.tolist() == exp["basis"]
    assert (main.var_count, main.constraint_count) == (exp["var_count"], exp["constraint_count"])
    assert lp.tableau_objective_value(main) == exp["objective"]
    if "initial_art" in case:
        art, exp = tabs[0], case["initial_art"]
        assert art.instance_problem.type == "min" and art.problem is p
        assert np.array_equal(art.matrix, np.array(fmat(exp["matrix"]), dtype=float))
        assert art.basis_columns.tolist() == exp["basis"]
        assert art.var_count == exp["var_count"]
        assert lp.tableau_objective_value(art) == exp["objective"]


def test_build_tableau_rejects_bad_operator(golden):
    """t/simplex.lisp:49-58: a /= constraint signals parsing-error."""
    p = lp.Problem(type="max", vars=["x", "y"], objective_func=[("x", 1), ("y", 2)],
                   constraints=[("<=", [("x", 5), ("y", 1)], 10), ("/=", [("x", 1), ("y", 1)], 5)])
    with pytest.raises(lp.ParsingError):
        lp.build_tableau(p, p)


@pytest.mark.parametrize("name", ["free_x", "free_x_negative", "ub_only_x", "lb_x", "range_y",
                                  "free_z_reduced_cost", "widgets", "excessive_constraints",
                                  "numerical_issue", "variable_bounds_bug", "variable_bounds_only"])
def test_build_tableau_matches_rational_restatement(golden, name):
    """Bounds handling (free / negative / shifted / ranged variables, the no-constraint special
    case): the product's f64 build_tableau equals the exact-rational restatement, entry by entry
    (the golden answers pin the latter in test_oracle_golden.py)."""
    case = golden["cases"][name]
    got = lp.build_tableau(_problem(case))
    exp = rr.build_tableau(goldens.problem(case))
    got = got if isinstance(got, list) else [got]
    exp = list(exp) if isinstance(exp, tuple) else [exp]
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        Me, be = goldens.to_f64(e)
        assert g.matrix.shape == Me.shape
        assert np.allclose(g.matrix, Me, rtol=1e-15, atol=0)
        assert g.basis_columns.tolist() == be.tolist()
        assert (g.var_count, g.constraint_count) == (e.var_count, e.constraint_count)
        assert g.is_max == e.is_max
        for var, mp in e.var_mapping.items():
            gm = g.var_mapping[var]
            assert gm[:2] == mp[:2]
            if len(mp) > 2:
                assert gm[2] == pytest.approx(float(mp[2]), rel=1e-15)


def test_random_mixed_problems_build_like_the_restatement():
    for seed in range(4):
        p = random_mixed_problem(lp, 12, 5, 4, 3, seed)
        d = {"type": p.type, "vars": p.vars, "objective_var": p.objective_var,
             "objective": [[v, Fraction(c)] for v, c in p.objective_func], "bounds": [],
             "constraints": [[op, [[v, Fraction(c)] for v, c in e], Fraction(r)]
                             for op, e, r in p.constraints]}
        art_e, main_e = rr.build_tableau(d)
        art, main = lp.build_tableau(p)
        assert np.allclose(art.matrix, goldens.to_f64(art_e)[0], rtol=1e-13, atol=0)
        assert np.array_equal(main.matrix, goldens.to_f64(main_e)[0])
        assert art.basis_columns.tolist() == art_e.basis
        assert main.basis_columns.tolist() == main_e.basis


def test_readback_on_host_arrays(golden):
    """tableau-variable / -reduced-cost / -objective-value (src/simplex.lisp:74-120) on a solved
    tableau supplied as host arrays (the golden final tableau), no device involved."""
    case = golden["cases"]["basic"]
    p = _problem(case)
    t0 = lp.build_tableau(p, p)
    t = lp.Tableau(p, p, np.array(fmat(case["final"]["matrix"]), dtype=float),
                   np.array(case["final"]["basis"]), t0.var_count, t0.constraint_count,
                   t0.var_mapping)
    assert lp.tableau_objective_value(t) == 28.5
    assert [lp.tableau_variable(t, v) for v in "wxyz"] == [28.5, 0.5, 7.0, 0.0]
    assert [lp.tableau_reduced_cost(t, v) for v in "xyz"] == [0.0, 0.0, 0.5]
    with pytest.raises(KeyError):
        lp.tableau_variable(t, "foo")
    with pytest.raises(KeyError):
        lp.tableau_reduced_cost(t, "w")
    assert lp.solution_problem(t) is p and lp.solution_objective_value(t) == 28.5


def test_synthetic_generator_is_the_documented_splitmix64():
    """First outputs of splitmix64 for seed 0 are the published test vector."""
    z = []
    s = 0
    for _ in range(3):
        s = (s + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        x = s
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        z.append(x ^ (x >> 31))
    assert z[0] == 0xE220A8397B1DCDAF and z[1] == 0x6E789E6AA1B965F4
    u = lp.synth.splitmix_u01(0, 0, 3)
    assert u.tolist() == [(v >> 11) * 2.0 ** -53 for v in z]
    assert lp.synth.splitmix_u01(0, 2, 1)[0] == u[2]            # random access == sequential


def test_synthetic_tableau_layout():
    n, m = 6, 4
    seed = lp.synth.seed_for(3)
    assert seed == (0x9E3779B97F4A7C15 ^ (20260928 + 3000))
    M, b = lp.synth.tableau(n, m, seed)
    A, rhs, c = lp.synth.lp_data(n, m, seed)
    assert M.shape == (m + 1, n + m + 1) and b.tolist() == [6, 7, 8, 9]
    assert np.array_equal(M[:m, :n], A) and np.array_equal(M[:m, n:n + m], np.eye(m))
    assert np.array_equal(M[:m, -1], rhs) and np.array_equal(M[m, :n], -c)
    assert not M[m, n:].any()
    assert (A >= 0.05).all() and (A < 1.05).all() and (rhs >= 0.25 * n).all() and (c >= 0.5).all()


def test_integer_problems_are_declined_before_any_device_work(golden):
    p = _problem(golden["cases"]["basic"])
    p.integer_vars = ["x", "y"]
    with pytest.raises(lp.UnsupportedConstraintError):
        lp.solve_problem(p)


# ------------------------------------------------------------------ native (C++) host side
ALL_LP = ["basic", "equality", "geq", "free_x", "free_x_negative", "ub_only_x", "lb_x", "range_y",
          "free_z_reduced_cost", "widgets", "excessive_constraints", "numerical_issue",
          "variable_bounds_bug", "variable_bounds_only"]


@pytest.mark.parametrize("name", ALL_LP)
def test_native_build_tableau_equals_python_mirror(golden, name):
    """csrc/host_problem.cpp's build-tableau (C++) and simplex.py's (Python) produce the same
    tableaux bit for bit, and both match the goldens / the rational restatement above."""
    p = _problem(golden["cases"][name])
    py = lp.build_tableau(p, p)
    py = py if isinstance(py, list) else [py]
    nat = lp.NativeProblem(p)
    got = nat.build_tableau()
    assert len(got) == len(py)
    for (M, b), t in zip(got, py):
        assert np.array_equal(M, t.matrix) and np.array_equal(b, t.basis_columns)
    for var in p.vars:
        exp = py[-1].var_mapping[var]
        assert nat.var_mapping(var) == tuple(exp)


def test_native_build_tableau_golden_matrices(golden):
    for name in ["basic", "equality", "geq"]:
        case = golden["cases"][name]
        got = lp.NativeProblem(_problem(case)).build_tableau()
        assert np.array_equal(got[-1][0], np.array(fmat(case["initial"]["matrix"]), dtype=float))
        assert got[-1][1].tolist() == case["initial"]["basis"]
        if "initial_art" in case:
            assert np.array_equal(got[0][0], np.array(fmat(case["initial_art"]["matrix"]), dtype=float))
            assert got[0][1].tolist() == case["initial_art"]["basis"]


def test_native_problem_validation():
    L = lp.capi.lib()
    import ctypes
    h = ctypes.c_void_p()
    assert L.mi355x_problem_create(ctypes.byref(h), 1, 0) == lp.capi.MI_BAD_ARG
    assert L.mi355x_problem_create(ctypes.byref(h), 1, 2) == 0
    v = np.array([0, 5], dtype=np.int64)
    c = np.array([1.0, 1.0])
    vp, cp = v.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p)
    assert L.mi355x_problem_set_objective(h, vp, cp, 2) == lp.capi.MI_BAD_ARG      # var 5 of 2
    assert L.mi355x_problem_add_constraint(h, 3, vp, cp, 1, 1.0) == lp.capi.MI_BAD_ARG  # bad op
    assert L.mi355x_problem_set_bounds(h, 0, 1, 2.0, 1, 1.0) == lp.capi.MI_BAD_ARG  # ub < lb
    assert L.mi355x_problem_set_bounds(h, 0, 1, 1.0, 1, 2.0) == 0
    L.mi355x_problem_destroy(h)
    p = lp.Problem(type="max", vars=["x"], objective_func=[("x", 1)])               # unbounded, no rows
    with pytest.raises(lp.UnboundedProblemError):
        lp.NativeProblem(p).build_tableau()
    p.integer_vars = ["x"]
    with pytest.raises(lp.UnsupportedConstraintError):
        lp.NativeProblem(p).solve()


# ------------------------------------------------------------------ property: build-tableau
from hypothesis import HealthCheck, given, settings, strategies as st   # noqa: E402


@settings(max_examples=150, deadline=None, derandomize=True, database=None,
          suppress_health_check=list(HealthCheck))
@given(n=st.integers(1, 7), seed=st.integers(0, 2 ** 31 - 1), kind=st.sampled_from(["max", "min"]),
       ncons=st.integers(0, 6))
def test_random_problems_native_equals_mirror_equals_rational(n, seed, kind, ncons):
    """Random small problems with every bound flavour (none / lb / ub / both / free), mixed
    <=, >=, = rows and negative right-hand sides: the C++ build-tableau, the Python mirror and
    the exact-rational restatement agree (bit for bit between the two float versions)."""
    rng = np.random.default_rng(seed)
    names = ["v%d" % i for i in range(n)]
    q = lambda: Fraction(int(rng.integers(-12, 13)), int(rng.integers(1, 5)))      # noqa: E731
    bounds = []
    for v in names:
        flavour = int(rng.integers(0, 5))
        lb, ub = q(), None
        ub = lb + abs(q())
        if flavour == 1:
            bounds.append([v, lb, None])
        elif flavour == 2:
            bounds.append([v, None, ub])
        elif flavour == 3:
            bounds.append([v, lb, ub])
        elif flavour == 4:
            bounds.append([v, None, None])
    cons = []
    for _ in range(ncons):
        expr = [[v, q()] for v in names if rng.uniform() < 0.7]
        expr = [e for e in expr if e[1] != 0] or [[names[0], Fraction(1)]]
        op = ["<=", ">=", "="][int(rng.integers(0, 3))]
        rhs = q() if op == "=" else abs(q())       # parsed <= / >= rows carry rhs >= 0
        cons.append([op, expr, rhs])
    obj = [[v, q()] for v in names]
    d = {"type": kind, "vars": names, "objective_var": "obj", "objective": obj, "bounds": bounds,
         "constraints": cons}
    p = lp.Problem.from_dict(d)
    try:
        exp = rr.build_tableau(d)
    except rr.Unbounded:
        with pytest.raises(lp.UnboundedProblemError):
            lp.build_tableau(p)
        with pytest.raises(lp.UnboundedProblemError):
            lp.NativeProblem(p).build_tableau()
        return
    exp = list(exp) if isinstance(exp, tuple) else [exp]
    py = lp.build_tableau(p)
    py = py if isinstance(py, list) else [py]
    nat = lp.NativeProblem(p).build_tableau()
    assert len(exp) == len(py) == len(nat)
    for e, t, (M, b) in zip(exp, py, nat):
        Me, be = goldens.to_f64(e)
        assert np.array_equal(M.view(np.int64), t.matrix.view(np.int64))          # C++ == Python
        assert b.tolist() == t.basis_columns.tolist() == be.tolist()
        assert np.allclose(M, Me, rtol=1e-13, atol=1e-13)                          # == exact / float
