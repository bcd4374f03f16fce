"""Pins the oracle (oracle/simplex_oracle.c and oracle/rational_ref.py) against
the reference's own known-answer tests, transcribed in
tests/golden/reference_cases.json.  CPU only."""
from fractions import Fraction

import numpy as np
import pytest

import oracle
from oracle import rational_ref as rr
from tests import goldens
from tests.goldens import fmat, frac

LP_CASES = ["basic", "equality", "geq", "free_x", "free_x_negative", "ub_only_x", "lb_x",
            "range_y", "free_z_reduced_cost", "widgets", "excessive_constraints",
            "numerical_issue", "variable_bounds_bug", "variable_bounds_only"]


def _build(case):
    return rr.build_tableau(goldens.problem(case))


# ---------------------------------------------------------------- build-tableau
@pytest.mark.parametrize("name", ["basic", "equality", "geq"])
def test_build_tableau_matches_reference(golden, name):
    """t/simplex.lisp:60-133."""
    case = golden["cases"][name]
    tabs = _build(case)
    main = tabs[1] if isinstance(tabs, tuple) else tabs
    exp = case["initial"]
    assert main.matrix == fmat(exp["matrix"])
    assert main.basis == exp["basis"]
    assert main.var_count == exp["var_count"]
    assert main.constraint_count == exp["constraint_count"]
    assert rr.objective_value(main) == exp["objective"]
    if "initial_art" in case:
        art = tabs[0]
        exp = case["initial_art"]
        assert art.matrix == fmat(exp["matrix"])
        assert art.basis == exp["basis"]
        assert art.var_count == exp["var_count"]
        assert not art.is_max          # artificial problem is a `min` problem
        assert rr.objective_value(art) == exp["objective"]


# ---------------------------------------------------------------------- one pivot
def test_one_pivot_rational_and_f64(golden):
    """t/simplex.lisp:135-159."""
    case = golden["cases"]["basic"]
    tab = _build(case)
    exp = case["one_pivot"]
    t2 = rr.pivot(tab.copy(), exp["entering_col"], exp["row"])
    assert t2.matrix == fmat(exp["matrix"]) and t2.basis == exp["basis"]
    assert tab.matrix == fmat(case["initial"]["matrix"])       # copy was pivoted, not the original
    M, basis = goldens.to_f64(tab)
    oracle.pivot(M, basis, exp["entering_col"], exp["row"])
    assert np.array_equal(M, np.array(fmat(exp["matrix"]), dtype=np.float64))   # all dyadic => exact
    assert basis.tolist() == exp["basis"]
    M2, basis2 = goldens.to_f64(tab)
    oracle.pivot(M2, basis2, exp["entering_col"], exp["row"], omp=True)
    assert np.array_equal(M, M2)


# -------------------------------------------------------------- full solves, exact
def _solve_rational(case):
    return rr.solve_any(_build(case))


def test_basic_final_tableau(golden):
    """t/simplex.lisp:170-194."""
    case = golden["cases"]["basic"]
    trace = []
    t = rr.solve_any(_build(case), trace)
    assert trace == [(1, 1), (0, 0)]
    assert t.matrix == fmat(case["final"]["matrix"]) and t.basis == case["final"]["basis"]


def test_equality_two_phase(golden):
    """t/simplex.lisp:196-237 -- also pins first-index-wins on ratio ties."""
    case = golden["cases"]["equality"]
    art, main = _build(case)
    art, main = art.copy(), main.copy()
    rr.solve_two_phase(art, main)
    assert any(art.matrix == fmat(a["matrix"]) and art.basis == a["basis"]
               for a in case["final_art"])
    assert rr.objective_value(art) == 0
    assert main.matrix == fmat(case["final"]["matrix"]) and main.basis == case["final"]["basis"]


def test_geq_two_phase(golden):
    """t/simplex.lisp:239-275."""
    case = golden["cases"]["geq"]
    art, main = _build(case)
    art, main = art.copy(), main.copy()
    rr.solve_two_phase(art, main)
    assert any(art.matrix == fmat(a["matrix"]) and art.basis == a["basis"]
               for a in case["final_art"])
    assert any(main.matrix == fmat(a["matrix"]) and main.basis == a["basis"]
               for a in case["final_alternatives"])
    assert rr.objective_value(main) == Fraction(85, 3)


def test_unsolvable(golden):
    """t/simplex.lisp:277-289."""
    with pytest.raises(rr.Infeasible):
        _solve_rational(golden["cases"]["infeasible"])
    with pytest.raises(rr.Unbounded):
        _solve_rational(golden["cases"]["unbounded"])


def _check_answers(case, objective, variable, reduced_cost, exact):
    f32 = bool(case.get("float32_literals"))

    def close(a, b):
        if exact:
            return a == b
        return abs(float(a) - float(b)) <= 1e-12 * max(1.0, abs(float(b)))
    if "objective" in case:
        assert close(objective(), frac(case["objective"], f32))
    for v, e in case.get("variables", {}).items():
        assert close(variable(v), frac(e, f32)), v
    for v, e in case.get("reduced_costs", {}).items():
        assert close(reduced_cost(v), frac(e, f32)), v
    for v, (lo, hi) in case.get("variable_ranges", {}).items():
        assert lo <= float(variable(v)) <= hi, v
    for v in case.get("reduced_cost_errors", []):
        with pytest.raises((KeyError, ValueError)):
            reduced_cost(v)
    for v in case.get("variable_errors", []):
        with pytest.raises(KeyError):
            variable(v)
    if "objective_fp_eq" in case:
        spec = case["objective_fp_eq"]
        tol = spec["factor"] * float(np.finfo(np.float32).eps) / 2 * (1 + 2.0 ** -23)
        assert abs(float(objective()) - float(np.float32(spec["value"]))) <= tol


@pytest.mark.parametrize("name", LP_CASES)
def test_answers_rational(golden, name):
    """Objective / variable / reduced-cost values of every LP case, exact arithmetic."""
    case = golden["cases"][name]
    t = _solve_rational(case)
    _check_answers(case, lambda: rr.objective_value(t), lambda v: rr.tableau_variable(t, v),
                   lambda v: rr.tableau_reduced_cost(t, v),
                   exact=not case.get("float32_literals"))


# ------------------------------------------------------ the C oracle (f64) itself
def _solve_f64(case):
    """build_tableau in rationals -> float64 -> C oracle.  Returns a Tableau of floats."""
    tabs = _build(case)
    if isinstance(tabs, tuple):
        art, main = tabs
        A, ab = goldens.to_f64(art)
        Mm, mb = goldens.to_f64(main)
        st, npv = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max)
        out = main.copy()
        arto = art.copy()
        arto.matrix, arto.basis = A.tolist(), ab.tolist()
    else:
        Mm, mb = goldens.to_f64(tabs)
        st, n, _ = oracle.solve(Mm, mb, is_max=tabs.is_max)
        out = tabs.copy()
        arto = None
    out.matrix = Mm.tolist()
    out.basis = mb.tolist()
    out.var_mapping = {k: tuple(float(x) if isinstance(x, Fraction) else x for x in v)
                       for k, v in out.var_mapping.items()}
    return st, out, arto


@pytest.mark.parametrize("name", LP_CASES)
def test_answers_c_oracle_f64(golden, name):
    case = golden["cases"][name]
    st, t, _ = _solve_f64(case)
    assert st == oracle.OPTIMAL
    _check_answers(case, lambda: rr.objective_value(t), lambda v: rr.tableau_variable(t, v),
                   lambda v: rr.tableau_reduced_cost(t, v), exact=False)


@pytest.mark.parametrize("name", ["basic", "equality", "geq"])
def test_final_tableaux_c_oracle_f64(golden, name):
    case = golden["cases"][name]
    st, t, art = _solve_f64(case)
    assert st == oracle.OPTIMAL
    alts = case.get("final_alternatives") or [case["final"]]
    got = np.array(t.matrix)

    def same(a):
        exp = np.array(fmat(a["matrix"]), dtype=np.float64)
        return t.basis == a["basis"] and np.allclose(got, exp, rtol=0, atol=4e-15)
    assert any(same(a) for a in alts)
    if art is not None:
        ga = np.array(art.matrix)
        assert any(art.basis == a["basis"]
                   and np.allclose(ga, np.array(fmat(a["matrix"]), dtype=np.float64), rtol=0,
                                   atol=4e-15) for a in case["final_art"])


def test_status_codes_c_oracle(golden):
    st, _, _ = _solve_f64(golden["cases"]["infeasible"])
    assert st == oracle.INFEASIBLE
    st, _, _ = _solve_f64(golden["cases"]["unbounded"])
    assert st == oracle.UNBOUNDED


def test_c_oracle_pivot_sequence_matches_rational(golden):
    """Where the arithmetic is dyadic the f64 path must take the same pivots as
    the rational path (pins strict-compare / first-index tie rules in C)."""
    for name in ["basic", "lb_x", "range_y", "free_x", "ub_only_x"]:
        case = golden["cases"][name]
        tab = _build(case)
        if isinstance(tab, tuple):
            continue
        tr = []
        rr.solve(tab.copy(), tr)
        M, b = goldens.to_f64(tab)
        st, n, trace = oracle.solve(M, b, is_max=tab.is_max, trace_cap=16)
        assert st == oracle.OPTIMAL and n == len(tr)
        assert [tuple(x) for x in trace.tolist()] == tr


def test_max_pivots_cap(golden):
    case = golden["cases"]["basic"]
    tab = _build(case)
    M, b = goldens.to_f64(tab)
    st, n, tr = oracle.solve(M, b, max_pivots=1, trace_cap=4)
    assert (st, n) == (oracle.MAX_PIVOTS, 1) and tr.tolist() == [[1, 1]]
    st, n, tr = oracle.solve(M, b, max_pivots=1, trace_cap=4)   # second pivot reaches the optimum
    assert (st, n) == (oracle.OPTIMAL, 1) and tr.tolist() == [[0, 0]]
    assert np.array_equal(M, np.array(fmat(case["final"]["matrix"]), dtype=np.float64))


# ------------------------------------------------------------- fp comparators
def test_fp_compare_tables(golden):
    """t/utils.lisp:90-93, 118-121, 147-150 (double-float rows)."""
    assert oracle.lib().orc_epsilon() == goldens.EPS == 2.0 ** -53 * (1 + 2.0 ** -52)
    for c in golden["fp_compare_double"]["cases"]:
        a = 4 * goldens.EPS if c["a"] == "4eps" else c["a"]
        b = 4 * goldens.EPS if c["b"] == "4eps" else c["b"]
        assert oracle.fp_compare(c["fn"], a, b, c["factor"]) is c["expect"], c


def test_thresholds():
    """128 eps / 512 eps / 1024 eps with the default factor 1024."""
    assert 128 * goldens.EPS == 1.4210854715202007e-14
    assert 512 * goldens.EPS == 5.684341886080803e-14
    assert 1024 * goldens.EPS == 1.1368683772161605e-13
    M = np.array([[1.0, 1.0, 1.0], [-1.4210854715202007e-14, 0.0, 0.0]])
    assert oracle.price(M) == -1                      # exactly -128 eps is NOT < -128 eps
    M[1, 0] = np.nextafter(M[1, 0], -1.0)
    assert oracle.price(M) == 0
    M = np.array([[5.684341886080803e-14, 0.0, 1.0], [-1.0, 0.0, 0.0]])
    assert oracle.ratio(M, 0) == -1                   # exactly 512 eps is not eligible
    M[0, 0] = np.nextafter(M[0, 0], 1.0)
    assert oracle.ratio(M, 0) == 0
