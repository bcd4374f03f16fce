"""Parity tests proper: the HIP path, called through the C ABI (via the host-side mirror of the
reference's interface), against (a) the reference's own known-answer vectors and (b) the C
oracle on the same inputs -- bit for bit.  Needs a real MI355X: `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest

import oracle
from oracle import rational_ref as rr
from tests import goldens
from tests.goldens import fmat, frac
from tests.helpers import lp_amd, random_mixed_problem

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _problem(case):
    return lp.Problem.from_dict(goldens.problem_dict(case))


def _f64(rows):
    return np.array(fmat(rows), dtype=np.float64)


def _oracle_solve(tabs):
    """Run the C oracle on the same initial f64 tableau(s).  Returns (status, M, basis, trace)."""
    if isinstance(tabs, list):
        art, main = tabs
        A, ab = art.matrix.copy(), art.basis_columns.copy()
        Mm, mb = main.matrix.copy(), main.basis_columns.copy()
        st, npv = oracle.solve_two_phase(A, ab, Mm, mb, main_is_max=main.is_max,
                                         factor=main.fp_tolerance_factor)
        return st, Mm, mb, (A, ab, npv)
    M, b = tabs.matrix.copy(), tabs.basis_columns.copy()
    st, n, trace = oracle.solve(M, b, is_max=tabs.is_max, factor=tabs.fp_tolerance_factor,
                                trace_cap=1 << 16)
    return st, M, b, trace


# =========================================================================== t/simplex.lisp
def test_pivot_row(golden):
    """t/simplex.lisp:135-159."""
    case = golden["cases"]["basic"]
    problem = _problem(case)
    tableau = lp.build_tableau(problem, problem)
    exp = case["one_pivot"]
    tableau2 = lp.pivot_row(tableau, exp["entering_col"], exp["row"])
    assert tableau is not tableau2
    assert lp.tableau_objective_value(tableau) == 0                 # original not mutated
    assert lp.n_pivot_row(tableau, exp["entering_col"], exp["row"]) is tableau
    assert np.array_equal(tableau.matrix, tableau2.matrix)
    assert np.array_equal(tableau.basis_columns, tableau2.basis_columns)
    assert tableau2.var_count == 5 and tableau2.constraint_count == 2
    assert np.array_equal(tableau.matrix, _f64(exp["matrix"]))
    assert tableau.basis_columns.tolist() == exp["basis"]
    assert lp.tableau_objective_value(tableau) == 4


def test_errors():
    """t/simplex.lisp:167-168."""
    with pytest.raises(TypeError):
        lp.n_solve_tableau("max x + y st 2x+y <= 5")


def test_basic_problem(golden):
    """t/simplex.lisp:170-194."""
    case = golden["cases"]["basic"]
    problem = _problem(case)
    tableau = lp.build_tableau(problem, problem)
    tableau2 = lp.solve_tableau(tableau)
    assert tableau is not tableau2
    assert lp.tableau_objective_value(tableau) == 0                 # original not solved
    assert lp.n_solve_tableau(tableau) is tableau
    assert np.array_equal(tableau.matrix, tableau2.matrix)
    assert np.array_equal(tableau.basis_columns, tableau2.basis_columns)
    assert np.array_equal(tableau.matrix, _f64(case["final"]["matrix"]))
    assert tableau.basis_columns.tolist() == case["final"]["basis"]
    assert lp.tableau_objective_value(tableau) == 28.5
    assert tableau.pivot_trace().tolist() == [[1, 1], [0, 0]]


def test_equality_constraint(golden):
    """t/simplex.lisp:196-237 (two-phase; pins first-index-wins on the ratio tie)."""
    case = golden["cases"]["equality"]
    problem = _problem(case)
    tableaus = lp.build_tableau(problem, problem)
    art_tab, main_tab = tableaus
    tab2 = lp.solve_tableau(tableaus)
    assert tab2 is not art_tab and tab2 is not main_tab
    assert np.array_equal(main_tab.matrix, _f64(case["initial"]["matrix"]))   # untouched
    assert lp.n_solve_tableau(tableaus) is main_tab
    assert np.array_equal(main_tab.matrix, tab2.matrix)
    assert np.array_equal(main_tab.basis_columns, tab2.basis_columns)
    assert any(np.array_equal(art_tab.matrix, _f64(a["matrix"]))
               and art_tab.basis_columns.tolist() == a["basis"] for a in case["final_art"])
    assert lp.tableau_objective_value(art_tab) == 0
    assert np.array_equal(main_tab.matrix, _f64(case["final"]["matrix"]))
    assert main_tab.basis_columns.tolist() == case["final"]["basis"]
    assert lp.tableau_objective_value(main_tab) == 28.5


def test_leq_constraint(golden):
    """t/simplex.lisp:239-275 (values are thirds: f64 agrees to a few ulp)."""
    case = golden["cases"]["geq"]
    problem = _problem(case)
    tableaus = lp.build_tableau(problem, problem)
    art_tab, main_tab = tableaus
    assert lp.n_solve_tableau(tableaus) is main_tab
    assert any(np.allclose(art_tab.matrix, _f64(a["matrix"]), rtol=0, atol=4e-15)
               and art_tab.basis_columns.tolist() == a["basis"] for a in case["final_art"])
    assert abs(lp.tableau_objective_value(art_tab)) == 0
    assert any(np.allclose(main_tab.matrix, _f64(a["matrix"]), rtol=0, atol=4e-15)
               and main_tab.basis_columns.tolist() == a["basis"]
               for a in case["final_alternatives"])
    assert abs(lp.tableau_objective_value(main_tab) - 85 / 3) <= 1e-10 * 85 / 3


def test_unsolvable_problems(golden):
    """t/simplex.lisp:277-289."""
    p = _problem(golden["cases"]["infeasible"])
    with pytest.raises(lp.InfeasibleProblemError):
        lp.solve_tableau(lp.build_tableau(p, p))
    p = _problem(golden["cases"]["unbounded"])
    with pytest.raises(lp.UnboundedProblemError):
        lp.solve_tableau(lp.build_tableau(p, p))


def test_copy_tableau(golden):
    """t/simplex.lisp:293-307."""
    p = _problem(golden["cases"]["basic"])
    t1 = lp.build_tableau(p, p)
    t2 = lp.copy_tableau(t1)
    assert t1 is not t2 and t1.problem is t2.problem
    assert t1.matrix is not t2.matrix and np.array_equal(t1.matrix, t2.matrix)
    assert t1.basis_columns is not t2.basis_columns
    assert np.array_equal(t1.basis_columns, t2.basis_columns)
    assert (t1.var_count, t1.constraint_count) == (t2.var_count, t2.constraint_count)
    lp.n_pivot_row(t2, 0, 0)
    assert not np.array_equal(t1.matrix, t2.matrix)                 # deep copy


def test_with_tableau_variables_and_with_solution_variables(golden):
    """t/simplex.lisp:391-405 and t/solver.lisp:117-127."""
    problem = _problem(golden["cases"]["basic"])
    tableau = lp.n_solve_tableau(lp.build_tableau(problem, problem))
    assert lp.with_tableau_variables(["x", "y", "z", "w"], tableau) == {"x": 0.5, "y": 7.0, "z": 0.0, "w": 28.5}
    assert lp.with_tableau_variables(problem, tableau) == {"w": 28.5, "x": 0.5, "y": 7.0, "z": 0.0}
    solution = lp.solve_problem(problem)
    values, reduced_cost = lp.with_solution_variables(["w", "x", "z"], solution)
    assert values == {"w": 28.5, "x": 0.5, "z": 0.0}
    assert reduced_cost("x") == 0.0 and reduced_cost("z") == 0.5


ANSWER_CASES = ["basic", "free_x", "free_x_negative", "ub_only_x", "lb_x", "range_y",
                "free_z_reduced_cost", "widgets", "excessive_constraints", "numerical_issue",
                "variable_bounds_bug", "variable_bounds_only", "equality", "geq"]


@pytest.mark.parametrize("name", ANSWER_CASES)
def test_golden_answers_and_oracle_bits(golden, name):
    """tableau-variable / reduced-cost / objective of every LP the reference tests
    (t/simplex.lisp:309-389, t/solver.lisp:20-32,70-83, t/integration.lisp:18-124), solved
    through the hook, plus bitwise equality with the C oracle on the same input."""
    case = golden["cases"][name]
    problem = _problem(case)
    f32 = bool(case.get("float32_literals"))
    st, M_or, b_or, _ = _oracle_solve(lp.build_tableau(problem, problem))
    assert st == oracle.OPTIMAL
    solution = lp.solve_problem(problem)
    assert lp.solution_problem(solution) is problem
    assert np.array_equal(solution.matrix, M_or), "HIP path differs from the oracle"
    assert np.array_equal(solution.basis_columns, b_or)

    def close(a, b):
        return abs(a - float(b)) <= 1e-10 * max(1.0, abs(float(b)))
    if "objective" in case:
        assert close(lp.solution_objective_value(solution), frac(case["objective"], f32))
    for v, e in case.get("variables", {}).items():
        assert close(lp.solution_variable(solution, v), frac(e, f32)), v
    for v, e in case.get("reduced_costs", {}).items():
        assert close(lp.solution_reduced_cost(solution, v), frac(e, f32)), v
    for v, (lo, hi) in case.get("variable_ranges", {}).items():
        assert lo <= lp.solution_variable(solution, v) <= hi, v
    for v in case.get("reduced_cost_errors", []):
        with pytest.raises((KeyError, ValueError)):
            lp.solution_reduced_cost(solution, v)
    for v in case.get("variable_errors", []):
        with pytest.raises(KeyError):
            lp.solution_variable(solution, v)
    if "objective_fp_eq" in case:
        spec = case["objective_fp_eq"]
        tol = spec["factor"] * 5.960464477539063e-08 * (1 + 2.0 ** -23)
        assert abs(lp.solution_objective_value(solution) - float(np.float32(spec["value"]))) <= tol


@pytest.mark.parametrize("name", ANSWER_CASES)
def test_native_solver_golden_answers(golden, name):
    """The all-C++ path (mi355x_simplex_solver + light solution object) on the same goldens."""
    case = golden["cases"][name]
    problem = _problem(case)
    f32 = bool(case.get("float32_literals"))
    sol = lp.NativeProblem(problem).solve()
    ref = lp.solve_problem(problem)
    assert sol.objective_value() == lp.solution_objective_value(ref)
    for v in problem.vars:
        assert sol.variable(v) == lp.solution_variable(ref, v)

    def close(a, b):
        return abs(a - float(b)) <= 1e-10 * max(1.0, abs(float(b)))
    if "objective" in case:
        assert close(sol.objective_value(), frac(case["objective"], f32))
    for v, e in case.get("variables", {}).items():
        assert close(sol.variable(v), frac(e, f32)), v
    for v, e in case.get("reduced_costs", {}).items():
        assert close(sol.reduced_cost(v), frac(e, f32)), v
    for v in case.get("reduced_cost_errors", []):
        with pytest.raises((KeyError, ValueError)):
            sol.reduced_cost(v)


def test_native_solver_errors(golden):
    with pytest.raises(lp.InfeasibleProblemError):
        lp.NativeProblem(_problem(golden["cases"]["infeasible"])).solve()
    with pytest.raises(lp.UnboundedProblemError):
        lp.NativeProblem(_problem(golden["cases"]["unbounded"])).solve()


def test_integer_problems_are_declined(golden):
    """A backend must signal unsupported-constraint-error for what it does not handle
    (src/conditions.lisp:69-77); B&B stays with the reference's own solver."""
    p = _problem(golden["cases"]["basic"])
    p.integer_vars = ["x"]
    with pytest.raises(lp.UnsupportedConstraintError) as e:
        lp.solve_problem(p)
    assert e.value.solver_name == "mi355x-simplex"


# =========================================================================== vs the C oracle
@pytest.mark.parametrize("n,m,seed", [(5, 3, 1), (33, 17, 2), (64, 32, 3), (200, 100, 4),
                                      (257, 511, 5), (1000, 7, 6), (7, 300, 7)])
def test_full_solve_bitwise_vs_oracle(n, m, seed):
    """Same pivot sequence and bit-identical final tableau on random dense LPs of odd shapes
    (row lengths that are not multiples of the vector width, single tiles, ragged tiles)."""
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, seed))
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, trace_cap=1 << 16)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    lp.n_solve_tableau(t)
    assert st == oracle.OPTIMAL and t.n_pivots == npiv
    assert np.array_equal(t.pivot_trace(), trace)
    assert np.array_equal(t.matrix, M)
    assert np.array_equal(t.basis_columns, b)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("n,m,seed", [(33, 17, 2), (257, 511, 5), (1500, 300, 8)])
def test_both_select_paths_bitwise_vs_oracle(n, m, seed, mode):
    """The single-workgroup select and the split (multi-workgroup) select are forced in turn on
    shapes either side of the automatic switch; both must reproduce the oracle exactly."""
    L = lp.capi.lib()
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, seed))
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, trace_cap=1 << 16)
    try:
        L.mi355x_tune_set_select_mode(mode)
        t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
        lp.n_solve_tableau(t)
    finally:
        L.mi355x_tune_set_select_mode(0)
    assert t.n_pivots == npiv and np.array_equal(t.pivot_trace(), trace)
    assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)


@pytest.fixture
def block_size(request):
    L = lp.capi.lib()
    L.mi355x_tune_set_block(request.param)
    yield request.param
    L.mi355x_tune_set_block(0)                         # the library default


@pytest.fixture(params=[2, 1], ids=["persistent-lookahead", "two-launches-per-step"])
def lookahead_mode(request):
    L = lp.capi.lib()
    L.mi355x_tune_set_lookahead_mode(request.param)
    yield request.param
    L.mi355x_tune_set_lookahead_mode(0)


@pytest.mark.parametrize("block_size", [1, 2, 3, 5, 8, 13, 16, 24, 28], indirect=True)
@pytest.mark.parametrize("n,m,seed", [(5, 3, 1), (33, 17, 2), (257, 511, 5), (700, 333, 6),
                                      (2000, 1100, 9)])
def test_blocked_pivoting_bitwise_vs_oracle(n, m, seed, block_size, lookahead_mode):
    """Blocked pivoting (k pivots selected ahead on the objective row / one column / the RHS /
    one row as they WOULD be, then applied in one sweep) for every block size, including sizes
    that do not divide the pivot count (the terminating step sits in the middle of a block): same
    pivot sequence, bit-identical tableau, and a capped solve stops on exactly the same pivot.
    Both forms of the look-ahead: two launches per step, and the whole block as one launch of
    persistent workgroups that exchange their reduction candidates through memory."""
    L = lp.capi.lib()
    L.mi355x_tune_set_select_mode(2)                    # small shapes too: the blocked path
    try:
        M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, seed))
        M, b = M0.copy(), b0.copy()
        st, npiv, trace = oracle.solve(M, b, trace_cap=1 << 16)
        t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
        lp.n_solve_tableau(t)
        assert st == oracle.OPTIMAL and t.n_pivots == npiv
        assert np.array_equal(t.pivot_trace(), trace)
        assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)
        cap = max(1, npiv // 2 + 1)                     # a cap in the middle of a block
        M, b = M0.copy(), b0.copy()
        st, _, _ = oracle.solve(M, b, max_pivots=cap)
        t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
        rc = L.mi355x_tab_solve(t._h, 1, 1024.0, cap, None)
        t._touch()
        assert rc == st == oracle.MAX_PIVOTS
        assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)
        lp.n_solve_tableau(t)                           # and on to optimality from there
        st, _, _ = oracle.solve(M, b)
        assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)
    finally:
        L.mi355x_tune_set_select_mode(0)


@pytest.mark.parametrize("n,m,cap", [(2600, 2300, 200), (4000, 4000, 100)])
def test_blocked_pivoting_many_lookahead_workgroups_bitwise(n, m, cap):
    """The persistent look-ahead with 9 and 16 workgroups exchanging their candidates (the small
    parity shapes need 1-5): the first pivots (ending inside a block), bit for bit against the
    OpenMP oracle."""
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, 77))
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, max_pivots=cap, trace_cap=1 << 16, omp=True)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    k = ctypes.c_int64(0)
    rc = lp.capi.lib().mi355x_tab_solve(t._h, 1, 1024.0, cap, ctypes.byref(k))
    t._touch()
    assert (rc, k.value) == (st, npiv)
    assert np.array_equal(t.pivot_trace(), trace)
    assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)


def _layout(t):
    c, cols, ld = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
    lp.capi.check(lp.capi.lib().mi355x_tab_layout(t._h, ctypes.byref(c), ctypes.byref(cols),
                                                  ctypes.byref(ld)), "layout")
    return c.value, cols.value, ld.value


@pytest.mark.parametrize("compact", [0, 1])
@pytest.mark.parametrize("n,m,seed", [(33, 17, 2), (257, 511, 5), (1500, 300, 8), (2000, 1100, 9)])
def test_dense_and_compact_representations_bitwise(n, m, seed, compact):
    """The solve loop on the dense logical tableau and on the compact [non-basic | RHS]
    representation: same pivots, same bits, and the representation is invisible to every
    other entry point (download / price / ratio / pivot / copy in between)."""
    L = lp.capi.lib()
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, seed))
    M, b = M0.copy(), b0.copy()
    try:
        L.mi355x_tune_set_compact(compact)
        t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
        rc = L.mi355x_tab_solve(t._h, 1, 1024.0, 3, None)
        t._touch()
        assert _layout(t) == ((1, n + 1, (n + 1 + 15) // 16 * 16) if compact
                              else (0, n + m + 1, (n + m + 1 + 15) // 16 * 16))
        st, _, _ = oracle.solve(M, b, max_pivots=3)
        assert rc == st == oracle.MAX_PIVOTS
        ec = lp.find_entering_column(t)                  # dense entry points in the middle
        assert _layout(t)[0] == 0
        assert ec == oracle.price(M) and lp.find_pivoting_row(t, ec) == oracle.ratio(M, ec)
        assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)
        t2 = lp.copy_tableau(t)
        cr = oracle.ratio(M, ec)
        lp.n_pivot_row(t, ec, cr)
        oracle.pivot(M, b, ec, cr)
        lp.n_solve_tableau(t)                            # back to compact, to optimality
        st, npiv, _ = oracle.solve(M, b)
        assert st == oracle.OPTIMAL and t.n_pivots == npiv
        assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)
        lp.n_solve_tableau(t2)                           # the copy, solved independently
        assert np.array_equal(t2.matrix, M)
    finally:
        L.mi355x_tune_set_compact(1)


@pytest.mark.parametrize("n,m,seed", [(40, 25, 1), (700, 333, 2), (1500, 300, 3)])
def test_compact_upload_equals_dense_upload(n, m, seed):
    """mi355x_tab_create_compact (only [A | b ; -c | 0] crosses the boundary, no dense buffer is
    ever allocated for solve + light read-back) against the ordinary dense upload of the same
    tableau and against the oracle; and the dense logical form still materialises on demand."""
    L = lp.capi.lib()
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(11, seed))
    stored = np.ascontiguousarray(np.concatenate([M0[:, :n], M0[:, -1:]], axis=1))
    cols = np.arange(n, dtype=np.int64)
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_compact(ctypes.byref(h), m + 1, n + m, n,
                                              stored.ctypes.data_as(ctypes.c_void_p),
                                              cols.ctypes.data_as(ctypes.c_void_p),
                                              b0.ctypes.data_as(ctypes.c_void_p), 0), "create_compact")
    t = lp.Tableau(None, lp.Problem(type="max"), None, None, n + m, m, {}, _handle=h)
    assert _layout(t) == (1, n + 1, (n + 1 + 15) // 16 * 16)
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, trace_cap=1 << 15)
    k = ctypes.c_int64(0)
    assert L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(k)) == st and k.value == npiv
    assert np.array_equal(t.pivot_trace(), trace)
    last_row = np.empty(n + m + 1); last_col = np.empty(m + 1); basis = np.empty(m, dtype=np.int64)
    lp.capi.check(L.mi355x_tab_download(h, None, basis.ctypes.data_as(ctypes.c_void_p),
                                        last_row.ctypes.data_as(ctypes.c_void_p),
                                        last_col.ctypes.data_as(ctypes.c_void_p)), "light download")
    assert _layout(t)[0] == 1                                  # still compact: nothing was expanded
    assert np.array_equal(last_row.view(np.int64), M[m].view(np.int64))
    assert np.array_equal(last_col.view(np.int64), M[:, -1].view(np.int64)) and np.array_equal(basis, b)
    t._touch()
    assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64))      # full download expands
    assert _layout(t)[0] == 0
    # bad inputs are refused
    bad = b0.copy(); bad[0] = 0                                 # a stored column cannot be basic too
    h2 = ctypes.c_void_p()
    assert L.mi355x_tab_create_compact(ctypes.byref(h2), m + 1, n + m, n, stored.ctypes.data_as(ctypes.c_void_p),
                                       cols.ctypes.data_as(ctypes.c_void_p),
                                       bad.ctypes.data_as(ctypes.c_void_p), 0) == lp.capi.MI_BAD_ARG


def test_inconsistent_basis_falls_back_to_dense():
    """A caller-supplied basis whose columns are NOT unit vectors (nothing in the reference
    forbids it): the compact representation is refused and the dense path reproduces the
    oracle; same for repeated and out-of-range basis entries."""
    rng = np.random.default_rng(3)
    n, m = 50, 20
    for kind in ["dense_columns", "repeated", "out_of_range", "minus_zero"]:
        M0, b0 = lp.synth.tableau(n, m, 77)
        if kind == "dense_columns":
            M0[:m, n:n + m] += rng.uniform(0.0, 0.1, (m, m))
        elif kind == "repeated":
            b0[3] = b0[2]
        elif kind == "out_of_range":
            b0[5] = n + m + 3                       # like build-tableau's marker for art rows
        else:
            M0[4, n + 2] = -0.0                     # a -0.0 inside a basic column
        M, b = M0.copy(), b0.copy()
        st, npiv, trace = oracle.solve(M, b, max_pivots=60, trace_cap=64)
        t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
        rc = lp.capi.lib().mi355x_tab_solve(t._h, 1, 1024.0, 60, None)
        assert _layout(t)[0] == 0, kind
        t._touch()
        assert rc == st and np.array_equal(t.pivot_trace(), trace), kind
        assert np.array_equal(t.matrix.view(np.int64), M.view(np.int64)), kind
        assert np.array_equal(t.basis_columns, b), kind


def test_every_update_variant_bitwise_vs_oracle():
    """All compiled tilings of the rank-1 update kernel give the same bits."""
    L = lp.capi.lib()
    n, m = 700, 333
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, 70))
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, max_pivots=40, trace_cap=64)
    try:
        for v in range(L.mi355x_tune_variant_count()):
            L.mi355x_tune_set_variant(v)
            t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
            with pytest.raises(lp.SolverError):
                lp.n_solve_tableau(t, max_pivots=40)
            assert np.array_equal(t.pivot_trace(), trace), v
            assert np.array_equal(t.matrix, M), v
    finally:
        L.mi355x_tune_set_variant(0)


def test_config2_full_solve_bitwise_vs_oracle():
    """BASELINE config 2: dense random LP 1024 vars x 512 constraints, solved to optimality."""
    n, m = 1024, 512
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(2))
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, trace_cap=1 << 16, omp=True)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    lp.n_solve_tableau(t)
    assert st == oracle.OPTIMAL and t.n_pivots == npiv and npiv > m // 4
    assert np.array_equal(t.pivot_trace(), trace)
    assert np.array_equal(t.matrix, M)
    assert np.array_equal(t.basis_columns, b)


@pytest.mark.parametrize("kind", ["max", "min"])
def test_min_and_max_problems_bitwise(kind):
    """min problems price with arg-max / fp> (src/simplex.lisp:373-379)."""
    rng = np.random.default_rng(11)
    n, m = 40, 25
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = rng.uniform(0.1, 1.0, (m, n))
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = rng.uniform(5, 10, m)
    c = rng.uniform(0.5, 1.5, n)
    M0[m, :n] = -c if kind == "max" else c       # a min problem improves on positive entries
    b0 = np.arange(n, n + m, dtype=np.int64)
    M, b = M0.copy(), b0.copy()
    st, npiv, trace = oracle.solve(M, b, is_max=(kind == "max"), trace_cap=4096)
    t = lp.Tableau(None, lp.Problem(type=kind), M0, b0, n + m, m, {})
    lp.n_solve_tableau(t)
    assert st == oracle.OPTIMAL and npiv > 0
    assert np.array_equal(t.pivot_trace(), trace) and np.array_equal(t.matrix, M)


@pytest.mark.parametrize("n,mle,mge,meq,seed", [(6, 3, 2, 1, 1), (30, 10, 8, 4, 2),
                                                (80, 30, 20, 10, 3), (40, 0, 25, 0, 4),
                                                (300, 100, 60, 30, 5), (12, 4, 4, 4, 6)])
@pytest.mark.parametrize("handover", [0, 1], ids=["column-parallel", "sequential"])
def test_two_phase_bitwise_vs_oracle(n, mle, mge, meq, seed, handover):
    """Two-phase branch (src/simplex.lisp:402-452) on random LPs with >= and = rows.  Whatever
    the reference's algorithm does on the input -- including declaring a feasible problem
    infeasible because phase 1 ends a few ulp above its 1024-eps test -- the HIP path does
    the same, bit for bit."""
    problem = random_mixed_problem(lp, n, mle, mge, meq, seed)
    tabs = lp.build_tableau(problem, problem)
    assert isinstance(tabs, list)
    st, M_or, b_or, (A_or, ab_or, npv) = _oracle_solve(tabs)
    art, main = tabs
    lp.capi.lib().mi355x_tune_set_handover_mode(handover)
    try:
        if st == oracle.OPTIMAL:
            lp.n_solve_tableau(tabs)
            assert main.n_pivots == (int(npv[0]), int(npv[1]))
        else:
            assert st == oracle.INFEASIBLE
            with pytest.raises(lp.InfeasibleProblemError):
                lp.n_solve_tableau(tabs)
    finally:
        lp.capi.lib().mi355x_tune_set_handover_mode(0)
    assert np.array_equal(art.matrix, A_or) and np.array_equal(art.basis_columns, ab_or)
    assert np.array_equal(main.matrix, M_or) and np.array_equal(main.basis_columns, b_or)


def _colpart_two_phase(tabs, shards):
    """mi355x_colpart_create on the artificial tableau + mi355x_colpart_solve_two_phase with the main
    tableau's objective row -> (status, pivots, artificial tableau, its basis, main tableau, its basis)."""
    import importlib
    cp = importlib.import_module("linear-programming_amd.colpart")
    art, main = tabs
    tab = cp.NativeColumnPartition.from_arrays(art.matrix.copy(), art.basis_columns.copy(), shards)
    try:
        rc, npv, mt = tab.solve_two_phase(main.matrix[-1].copy(), main.is_max, main.fp_tolerance_factor)
        A, ab, _, _ = tab.download()
        Mm = mb = None
        if mt is not None:
            Mm, mb, last_row, last_col = mt.download()
            assert np.array_equal(last_row.view(np.int64), Mm[-1].view(np.int64))
            assert np.array_equal(last_col.view(np.int64), Mm[:, -1].view(np.int64))
            mt.close()
    finally:
        tab.close()
    return rc, npv, A, ab, Mm, mb


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
@pytest.mark.parametrize("name", ["equality", "geq"])
def test_two_phase_goldens_on_the_column_partition(golden, name, shards):
    """The reference's two-phase goldens (t/simplex.lisp:196-275) with the artificial tableau
    column-partitioned over 1 / 2 / 3 / 8 logical shards: phase 1, the hand-over and phase 2 all
    on the partition -- same bits as the oracle (and therefore as the single-device path)."""
    case = golden["cases"][name]
    problem = _problem(case)
    tabs = lp.build_tableau(problem, problem)
    st, M_or, b_or, (A_or, ab_or, npv) = _oracle_solve(tabs)
    rc, got_npv, A, ab, Mm, mb = _colpart_two_phase(tabs, shards)
    assert rc == st == oracle.OPTIMAL and got_npv == (int(npv[0]), int(npv[1]))
    assert np.array_equal(A.view(np.int64), A_or.view(np.int64)) and np.array_equal(ab, ab_or)
    assert np.array_equal(Mm.view(np.int64), M_or.view(np.int64)) and np.array_equal(mb, b_or)
    # and through the hook: (solve-problem problem :devices n)
    sol = lp.solve_problem(problem, devices=max(shards, 2))
    assert np.array_equal(sol.matrix.view(np.int64), M_or.view(np.int64)) and np.array_equal(sol.basis_columns, b_or)
    if name == "equality":
        assert lp.solution_objective_value(sol) == 28.5


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
@pytest.mark.parametrize("n,mle,mge,meq,seed", [(6, 3, 2, 1, 1), (30, 10, 8, 4, 2), (80, 30, 20, 10, 3),
                                                (40, 0, 25, 0, 4), (300, 100, 60, 30, 5), (12, 4, 4, 4, 6),
                                                (9, 2, 7, 5, 7)])
def test_two_phase_on_the_column_partition_bitwise_vs_oracle(n, mle, mge, meq, seed, shards):
    """Random LPs with >= and = rows through mi355x_colpart_solve_two_phase on 1 / 2 / 3 / 8 logical
    shards (the last shape leaves shards with artificial columns only: dead slots)."""
    problem = random_mixed_problem(lp, n, mle, mge, meq, seed)
    tabs = lp.build_tableau(problem, problem)
    st, M_or, b_or, (A_or, ab_or, npv) = _oracle_solve(tabs)
    rc, got_npv, A, ab, Mm, mb = _colpart_two_phase(tabs, shards)
    assert rc == st
    assert np.array_equal(A.view(np.int64), A_or.view(np.int64)) and np.array_equal(ab, ab_or)
    if st == oracle.OPTIMAL:
        assert got_npv == (int(npv[0]), int(npv[1]))
        assert np.array_equal(Mm.view(np.int64), M_or.view(np.int64)) and np.array_equal(mb, b_or)
    else:
        assert st == oracle.INFEASIBLE and Mm is None


def test_two_phase_column_partition_drives_degenerate_artificials_out():
    """Artificial variables still basic at level zero after phase 1 (src/simplex.lisp:419-434): the
    partition pivots them out with a caller-chosen column AND row (the owner contributes the
    column, every shard pivots on the given row).  Equality rows through a degenerate vertex (a
    point with zero components, integer data) make that happen in about half the draws; the rest
    are ordinary two-phase solves or end as the reference's "cannot be replaced" error."""
    hit = 0
    for seed in range(40):
        rng = np.random.default_rng(seed)
        n = 6
        names = ["x%d" % i for i in range(n)]
        x0 = rng.integers(0, 3, n).astype(float) * (rng.uniform(size=n) < 0.5)
        rows = [rng.integers(0, 3, n).astype(float) for _ in range(4)]
        cons = [("=", list(zip(names, a.tolist())), float(a @ x0)) for a in rows if a.any()]
        cons.append(("<=", list(zip(names, [1.0] * n)), float(x0.sum() + 3)))
        problem = lp.Problem(type="max", vars=names, objective_var="obj",
                             objective_func=list(zip(names, rng.integers(1, 4, n).astype(float).tolist())),
                             constraints=cons)
        tabs = lp.build_tableau(problem, problem)
        if not isinstance(tabs, list):
            continue
        st, M_or, b_or, (A_or, ab_or, npv) = _oracle_solve(tabs)
        # drive-out pivots happened iff phase 1 of the oracle counted more pivots than its plain solve
        A1, b1 = tabs[0].matrix.copy(), tabs[0].basis_columns.copy()
        _, n_plain, _ = oracle.solve(A1, b1, is_max=False)
        drove = int(npv[0]) > n_plain
        for shards in (1, 2, 3, 8):
            # (a drive-out pivot on a NEGATIVE element leaves -0.0 in basic columns, which compact shards do
            # not store: declined with MI_UNSUPPORTED until round 5; since round 6 the partition moves the
            # tableau to dense shards at that pivot and follows the reference bit for bit, -0.0s included)
            rc, got_npv, A, ab, Mm, mb = _colpart_two_phase(tabs, shards)
            assert rc == st, (seed, shards, rc, st)
            if st in (oracle.OPTIMAL, oracle.UNBOUNDED):
                assert got_npv[0] == int(npv[0])
                assert np.array_equal(A.view(np.int64), A_or.view(np.int64)) and np.array_equal(ab, ab_or)
                assert np.array_equal(Mm.view(np.int64), M_or.view(np.int64)) and np.array_equal(mb, b_or)
        # and the hook (:devices 3) ends with the oracle's bits
        if st == oracle.OPTIMAL:
            sol = lp.solve_problem(problem, devices=3)
            assert np.array_equal(sol.matrix.view(np.int64), M_or.view(np.int64)) and np.array_equal(sol.basis_columns, b_or)
        hit += drove and st == oracle.OPTIMAL
    assert hit >= 10, "the seeds produced too few drive-out pivots (%d)" % hit
    # Drive-out pivots on POSITIVE elements (equality rows with right-hand side 0 whose artificial
    # columns cancel in the phase-1 objective: phase 1 is optimal at once with the artificials still
    # basic; seeds searched on the oracle): the partition follows, bit for bit.
    followed = 0
    for seed in (282, 957, 959, 1396, 1481, 1610, 1832, 2136, 2288, 2340, 2737):
        rng = np.random.default_rng(seed)
        n = 5
        names = ["x%d" % i for i in range(n)]
        rows = [rng.integers(-2, 3, n).astype(float) for _ in range(3)]
        cons = [("=", list(zip(names, a.tolist())), 0.0) for a in rows if a.any()]
        cons.append(("<=", list(zip(names, [1.0] * n)), 5.0))
        problem = lp.Problem(type="max", vars=names, objective_var="obj",
                             objective_func=list(zip(names, rng.integers(1, 4, n).astype(float).tolist())),
                             constraints=cons)
        tabs = lp.build_tableau(problem, problem)
        st, M_or, b_or, (A_or, ab_or, npv) = _oracle_solve(tabs)
        A1, b1 = tabs[0].matrix.copy(), tabs[0].basis_columns.copy()
        _, n_plain, _ = oracle.solve(A1, b1, is_max=False)
        assert st == oracle.OPTIMAL and int(npv[0]) > n_plain            # drive-out pivots happened
        for shards in (1, 2, 3):
            rc, got_npv, A, ab, Mm, mb = _colpart_two_phase(tabs, shards)
            assert rc == st and got_npv == (int(npv[0]), int(npv[1])), (seed, shards)
            assert np.array_equal(A.view(np.int64), A_or.view(np.int64)) and np.array_equal(ab, ab_or), (seed, shards)
            assert np.array_equal(Mm.view(np.int64), M_or.view(np.int64)) and np.array_equal(mb, b_or), (seed, shards)
            followed += 1
    assert followed == 33


def test_two_phase_column_partition_nonfinite_objective_coefficient_on_a_basic_column():
    """src/simplex.lisp:444-451 reads every scale of the re-elimination from the objective row AS REDUCED SO FAR.
    An infinite objective coefficient on a variable that is basic after phase 1 makes the first product
    inf * 0 a NaN in the objective entries of the OTHER basic columns -- later scales are then NaNs, and the
    main tableau holds NaNs in columns a compact shard does not store.  Declined (MI_UNSUPPORTED) until round
    5; since round 6 the partition moves to dense shards, computes the scales with the sequential loop on the
    basic block (k_handover_scales_seq) and follows the oracle bit for bit -- whatever that outcome is."""
    seen = 0
    for seed in range(12):
        rng = np.random.default_rng(1000 + seed)
        n = 5
        names = ["x%d" % i for i in range(n)]
        x0 = rng.integers(1, 4, n).astype(float)
        rows = [rng.integers(0, 3, n).astype(float) for _ in range(2)]
        cons = [("=", list(zip(names, a.tolist())), float(a @ x0)) for a in rows if a.any()]
        cons.append(("<=", list(zip(names, [1.0] * n)), float(x0.sum() + 2)))
        cons.append((">=", [(names[0], 1.0)], 1.0))
        c = rng.integers(1, 4, n).astype(float)
        c[int(rng.integers(0, n))] = np.inf if seed % 2 else -np.inf
        problem = lp.Problem(type="max", vars=names, objective_var="obj",
                             objective_func=list(zip(names, c.tolist())), constraints=cons)
        tabs = lp.build_tableau(problem, problem)
        if not isinstance(tabs, list):
            continue
        with np.errstate(all="ignore"):
            st, M_or, b_or, (A_or, ab_or, npv) = _oracle_solve(tabs)
        if st == oracle.INFEASIBLE:
            continue
        nonfinite_basic = not np.all(np.isfinite(tabs[1].matrix[-1][ab_or]))
        for shards in (1, 2, 3):
            rc, got_npv, A, ab, Mm, mb = _colpart_two_phase(tabs, shards)
            assert rc == st, (seed, shards, rc, st)
            assert got_npv == (int(npv[0]), int(npv[1])), (seed, shards, got_npv, npv)
            assert _same_bits_nan_aware(A, A_or) and np.array_equal(ab, ab_or), (seed, shards)
            assert _same_bits_nan_aware(Mm, M_or) and np.array_equal(mb, b_or), (seed, shards)
        seen += nonfinite_basic
    assert seen >= 3, "too few draws put the infinite coefficient on a basic column (%d)" % seen


def test_degenerate_shapes():
    """Edge shapes: no constraint rows at all, a single column, one row x many columns, many
    rows x few columns, a 1 x 1 tableau -- same outcome and bits as the oracle."""
    def both(M0, b0, is_max=True):
        M, b = M0.copy(), b0.copy()
        st, n, tr = oracle.solve(M, b, is_max=is_max, trace_cap=4096)
        t = lp.Tableau(None, lp.Problem(type="max" if is_max else "min"), M0, b0,
                       M0.shape[1] - 1, M0.shape[0] - 1, {})
        rc = lp.capi.lib().mi355x_tab_solve(t._h, int(is_max), 1024.0, 0, None)
        t._touch()
        assert rc == st
        assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)
        assert np.array_equal(t.pivot_trace(), tr)
        return st
    e = np.zeros(0, dtype=np.int64)
    assert both(np.array([[3.0]]), e) == oracle.OPTIMAL                       # 1 x 1: nothing to price
    assert both(np.array([[0.5, -1.0, 2.0, 7.0]]), e) == oracle.UNBOUNDED     # objective row only
    assert both(np.array([[0.5, 1.0, 2.0, 7.0]]), e) == oracle.OPTIMAL
    assert both(np.array([[0.5, -1.0, 2.0, 7.0]]), e, is_max=False) == oracle.UNBOUNDED
    M0, b0 = lp.synth.tableau(20000, 1, 3)                                     # one constraint
    assert both(M0, b0) == oracle.OPTIMAL
    M0, b0 = lp.synth.tableau(3, 5000, 4)                                      # tall and thin
    assert both(M0, b0) == oracle.OPTIMAL
    M0 = np.array([[1.0, 4.0], [-1.0, 0.0]])                                   # one variable column
    assert both(M0, np.array([0], dtype=np.int64)) == oracle.OPTIMAL


def test_forced_pivot_sequence_bitwise():
    """n-pivot-row with caller-chosen pivots (not the Dantzig choice), many in a row."""
    rng = np.random.default_rng(5)
    R, C = 37, 101
    M0 = rng.uniform(0.5, 1.5, (R, C))
    b0 = np.arange(R - 1, dtype=np.int64)
    M, b = M0.copy(), b0.copy()
    t = lp.Tableau(None, lp.Problem(), M0, b0, C - 1, R - 1, {})
    for k in range(60):
        ec, cr = (7 * k + 3) % C, (5 * k + 1) % (R - 1)
        oracle.pivot(M, b, ec, cr)
        lp.n_pivot_row(t, ec, cr)
    assert np.array_equal(t.matrix, M)
    assert np.array_equal(t.basis_columns, b)
    assert np.array_equal(t.pivot_trace(), [[(7 * k + 3) % C, (5 * k + 1) % (R - 1)] for k in range(60)])


def test_price_and_ratio_entry_points():
    """find-entering-column / find-pivoting-row on their own, incl. the exact thresholds."""
    eps = lp.capi.lib().mi355x_epsilon()
    assert eps == oracle.EPSILON
    M0, b0 = lp.synth.tableau(50, 20, 77)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, 70, 20, {})
    ec = lp.find_entering_column(t)
    assert ec == oracle.price(M0)
    assert lp.find_pivoting_row(t, ec) == oracle.ratio(M0, ec)
    # exactly -128 eps is NOT below the pricing threshold, the next double is
    M = np.array([[1.0, 1.0, 1.0], [-128 * eps, 0.0, 0.0]])
    t = lp.Tableau(None, lp.Problem(type="max"), M, np.array([1], dtype=np.int64), 2, 1, {})
    assert lp.find_entering_column(t) is None
    M[1, 0] = np.nextafter(M[1, 0], -1.0)
    t = lp.Tableau(None, lp.Problem(type="max"), M, np.array([1], dtype=np.int64), 2, 1, {})
    assert lp.find_entering_column(t) == 0
    # exactly 512 eps is not an eligible pivot element, the next double is
    M = np.array([[512 * eps, 0.0, 1.0], [-1.0, 0.0, 0.0]])
    t = lp.Tableau(None, lp.Problem(type="max"), M, np.array([1], dtype=np.int64), 2, 1, {})
    assert lp.find_pivoting_row(t, 0) is None
    M[0, 0] = np.nextafter(M[0, 0], 1.0)
    t = lp.Tableau(None, lp.Problem(type="max"), M, np.array([1], dtype=np.int64), 2, 1, {})
    assert lp.find_pivoting_row(t, 0) == 0


def test_ties_take_the_lowest_index():
    """Exact ties in pricing and in the ratio test: first index wins, wherever the tied
    entries fall relative to wave / thread boundaries."""
    n, m = 3000, 2500
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = 1.0
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = 4.0
    M0[m, :n] = -1.0
    for cols, rows in [((5, 70, 1029, 2999), (3, 64, 1024, 2047)), ((2998, 2999), (2498, 2499))]:
        M = M0.copy()
        M[m, list(cols)] = -2.0                  # tied most-negative reduced costs
        M[list(rows), -1] = 2.0                  # tied minimum ratios
        t = lp.Tableau(None, lp.Problem(type="max"), M, np.arange(n, n + m, dtype=np.int64),
                       n + m, m, {})
        ec = lp.find_entering_column(t)
        assert ec == cols[0] == oracle.price(M)
        assert lp.find_pivoting_row(t, ec) == rows[0] == oracle.ratio(M, ec)


def test_max_pivots_and_resume():
    M0, b0 = lp.synth.tableau(120, 60, 5)
    M, b = M0.copy(), b0.copy()
    st, total, trace = oracle.solve(M, b, trace_cap=4096)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, 180, 60, {})
    with pytest.raises(lp.SolverError):
        lp.n_solve_tableau(t, max_pivots=7)
    assert t.n_pivots == 7
    Mo, bo = M0.copy(), b0.copy()
    oracle.solve(Mo, bo, max_pivots=7)
    assert np.array_equal(t.matrix, Mo)
    lp.n_solve_tableau(t)                        # resume to optimality
    assert t.n_pivots == total - 7
    assert np.array_equal(t.matrix, M) and np.array_equal(t.pivot_trace(), trace)


def test_handles_are_independent_across_host_threads():
    """The ABI promises thread-safety per handle: several host threads solve different LPs on
    their own handles at the same time (ctypes drops the GIL during the calls) and every result
    still matches the oracle bit for bit."""
    import threading
    jobs = [(120 + 10 * k, 60 + 5 * k, lp.synth.seed_for(7, k)) for k in range(6)]
    results, errors = {}, []

    def work(k, n, m, seed):
        try:
            M0, b0 = lp.synth.tableau(n, m, seed)
            t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
            for _ in range(3):                       # solve, download, re-upload, solve again
                lp.n_solve_tableau(t)
                results[k] = (t.matrix.copy(), t.basis_columns.copy(), t.pivot_trace())
                lp.capi.check(lp.capi.lib().mi355x_tab_upload(
                    t._h, M0.ctypes.data_as(ctypes.c_void_p), b0.ctypes.data_as(ctypes.c_void_p)), "upload")
                t._touch()
        except Exception as e:                       # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,) + j) for k, j in enumerate(jobs)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for k, (n, m, seed) in enumerate(jobs):
        M, b = lp.synth.tableau(n, m, seed)
        st, npiv, trace = oracle.solve(M, b, trace_cap=1 << 14)
        Mg, bg, tg = results[k]
        assert np.array_equal(Mg, M) and np.array_equal(bg, b) and np.array_equal(tg, trace)


def test_async_enqueue_matches_blocking_solve():
    """mi355x_tab_solve_async + mi355x_tab_sync (what bench.py times)."""
    n, m = 300, 150
    M0, b0 = lp.synth.tableau(n, m, 21)
    t = lp.Tableau(None, lp.Problem(type="max"), M0, b0, n + m, m, {})
    L = lp.capi.lib()
    lp.capi.check(L.mi355x_tab_solve_async(t._h, 1, 1024.0, 25, 1), "solve_async")
    npv = ctypes.c_int64(0)
    rc = L.mi355x_tab_sync(t._h, ctypes.byref(npv))
    assert rc == lp.capi.MI_RUNNING and npv.value == 25
    t._touch()
    Mo, bo = M0.copy(), b0.copy()
    oracle.solve(Mo, bo, max_pivots=25)
    assert np.array_equal(t.matrix, Mo)
    # enqueue far more iterations than the LP needs: the surplus must be no-ops
    lp.capi.check(L.mi355x_tab_solve_async(t._h, 1, 1024.0, 3000, 0), "solve_async")
    rc = L.mi355x_tab_sync(t._h, ctypes.byref(npv))
    assert rc == lp.capi.MI_OPTIMAL
    t._touch()
    st, total, _ = oracle.solve(Mo, bo)
    assert npv.value == 25 + total and np.array_equal(t.matrix, Mo)


# =========================================================================== batches (config 4)
@pytest.fixture(params=[3, 2, 1], ids=["lookahead-per-LP+sweep-over-all-LPs", "one-workgroup-per-LP",
                                        "lockstep-launch-pairs"])
def batch_mode(request):
    L = lp.capi.lib()
    L.mi355x_tune_set_batch_mode(request.param)
    yield request.param
    L.mi355x_tune_set_batch_mode(0)


@pytest.fixture(params=[1, 0], ids=["compact", "dense"])
def representation(request):
    L = lp.capi.lib()
    L.mi355x_tune_set_compact(request.param)
    yield request.param
    L.mi355x_tune_set_compact(1)


@pytest.mark.parametrize("block", [1, 4, 8, 16])
@pytest.mark.parametrize("n,m,nl", [(60, 30, 21), (7, 3, 5), (300, 40, 7), (33, 200, 6)])
def test_batch_blocked_kernel_bitwise_vs_oracle(block, n, m, nl):
    """The one-workgroup-per-LP kernel with blocked pivoting (look-ahead state in LDS, the tableau
    read and written once per block) for every block size it is instantiated for, and a pivot cap
    that falls inside a block: every LP bit-identical to the oracle run on it alone."""
    L = lp.capi.lib()
    seeds = [lp.synth.seed_for(4, 100 + k) for k in range(nl)]
    tabs = [lp.synth.tableau(n, m, s) for s in seeds]
    Ms = np.stack([t[0] for t in tabs])
    Bs = np.stack([t[1] for t in tabs])
    L.mi355x_tune_set_batch_block(block)
    try:
        for cap in (0, 5):
            batch = lp.TableauBatch.from_arrays(Ms, Bs)
            st, npv = batch.solve(max_pivots=cap)
            for k in range(nl):
                M, b = Ms[k].copy(), Bs[k].copy()
                so, no, _ = oracle.solve(M, b, max_pivots=cap)
                Mg, bg = batch.download(k)
                assert (st[k], npv[k]) == (so, no), (k, cap)
                assert np.array_equal(Mg, M) and np.array_equal(bg, b), (k, cap)
    finally:
        L.mi355x_tune_set_batch_block(0)


@pytest.mark.parametrize("n_sub", [1, 3, 8])
def test_multi_device_batch_logical_sub_batches_bitwise(n_sub):
    """mi355x_multibatch_*: one batch handle over n sub-batches (one per device; this box has one
    GPU, so they are logical sub-batches on it, each driven by its own worker thread of the
    library and its own stream, all in flight at once).  Status, pivot count and final tableau of
    every LP against the oracle, by GLOBAL LP index; ragged split (37 LPs over 8)."""
    n, m, nl = 60, 35, 37
    tabs = [lp.synth.tableau(n, m, lp.synth.seed_for(4, 100 + k)) for k in range(nl)]
    Ms = np.stack([x[0] for x in tabs]); Bs = np.stack([x[1] for x in tabs])
    mb = lp.MultiDeviceBatch.from_arrays(Ms, Bs, n_sub)
    assert mb.info() == {"n_sub_batches": n_sub, "n_devices_used": 1}
    st, npv = mb.solve()
    for k in range(nl):
        M, b = Ms[k].copy(), Bs[k].copy()
        so, no, _ = oracle.solve(M, b)
        G, bg = mb.download(k)
        assert (int(st[k]), int(npv[k])) == (so, no), k
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b), k
    # synthetic form (generated in HBM), more sub-batches than LPs asked for: capped
    seeds = np.array([lp.synth.seed_for(4, 300 + k) for k in range(5)], dtype=np.uint64)
    mb2 = lp.MultiDeviceBatch.synthetic(5, 40, 20, seeds, 8)
    assert mb2.info()["n_sub_batches"] == 5
    st, npv = mb2.solve(max_pivots=7)
    for k in range(5):
        M, b = lp.synth.tableau(40, 20, int(seeds[k]))
        so, no, _ = oracle.solve(M, b, max_pivots=7)
        G, bg = mb2.download(k)
        assert (int(st[k]), int(npv[k])) == (so, no)
        assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, b)


def test_batch_solve_async_overlaps_two_batches():
    """mi355x_batch_solve_async / _sync: two batches started back to back from ONE host thread and
    collected afterwards -- both equal the blocking solve's results; a second async call on a
    batch in flight and a sync without a solve are refused."""
    L = lp.capi.lib()
    n, m = 80, 45
    sets = []
    for base in (0, 50):
        tabs = [lp.synth.tableau(n, m, lp.synth.seed_for(4, 500 + base + k)) for k in range(9)]
        sets.append((np.stack([x[0] for x in tabs]), np.stack([x[1] for x in tabs])))
    batches = [lp.TableauBatch.from_arrays(Ms, Bs) for Ms, Bs in sets]
    assert L.mi355x_batch_sync(batches[0]._h, None, None) == lp.capi.MI_BAD_ARG
    for b in batches:
        b.solve_async()
    assert L.mi355x_batch_solve_async(batches[0]._h, 1, 1024.0, 0) == lp.capi.MI_BAD_ARG
    for b, (Ms, Bs) in zip(batches, sets):
        st, npv = b.sync()
        for k in range(Ms.shape[0]):
            M, bb = Ms[k].copy(), Bs[k].copy()
            so, no, _ = oracle.solve(M, bb)
            G, bg = b.download(k)
            assert (int(st[k]), int(npv[k])) == (so, no)
            assert np.array_equal(G.view(np.int64), M.view(np.int64)) and np.array_equal(bg, bb)


@pytest.mark.parametrize("n,m,nl", [(60, 30, 37), (7, 3, 5), (300, 40, 9), (33, 200, 6)])
def test_batch_bitwise_vs_oracle(batch_mode, representation, n, m, nl):
    """Every LP of a batch ends bit-identical to the oracle run on it alone (LPs of different
    pivot counts, so finished LPs idle while others continue), with both batch drivers and both
    tableau representations."""
    seeds = [lp.synth.seed_for(4, k) for k in range(nl)]
    tabs = [lp.synth.tableau(n, m, s) for s in seeds]
    Ms = np.stack([t[0] for t in tabs])
    Bs = np.stack([t[1] for t in tabs])
    batch = lp.TableauBatch.from_arrays(Ms, Bs)
    st, npv = batch.solve()
    pivots = []
    for k in range(nl):
        M, b = Ms[k].copy(), Bs[k].copy()
        so, no, _ = oracle.solve(M, b)
        pivots.append(no)
        Mg, bg = batch.download(k)
        assert st[k] == so == oracle.OPTIMAL and npv[k] == no, k
        assert np.array_equal(Mg, M) and np.array_equal(bg, b), k
    assert nl < 10 or len(set(pivots)) > 3


def test_batch_synthetic_config4_shape_and_statuses(batch_mode):
    """BASELINE config 4 shape (512 vars x 256 constraints) on a sub-batch: device generator ==
    numpy generator per LP, mixed outcomes (an unbounded LP in the batch), pivot cap."""
    n, m, nl = 512, 256, 8
    seeds = np.array([lp.synth.seed_for(4, k) for k in range(nl)], dtype=np.uint64)
    batch = lp.TableauBatch.synthetic(nl, n, m, seeds)
    for k in (0, nl - 1):
        M, b = lp.synth.tableau(n, m, int(seeds[k]))
        Mg, bg = batch.download(k)
        assert np.array_equal(Mg, M) and np.array_equal(bg, b)
    st, npv = batch.solve(max_pivots=50)
    assert (st == lp.capi.MI_MAX_PIVOTS).all() and (npv == 50).all()
    M, b = lp.synth.tableau(n, m, int(seeds[3]))
    oracle.solve(M, b, max_pivots=50)
    assert np.array_equal(batch.download(3)[0], M)
    st, npv = batch.solve()                       # resume to optimality
    M, b = lp.synth.tableau(n, m, int(seeds[3]))
    so, no, _ = oracle.solve(M, b, omp=True)
    assert st[3] == so == oracle.OPTIMAL and npv[3] == no - 50
    assert np.array_equal(batch.download(3)[0], M)
    # an unbounded member does not disturb the others
    tabs = [lp.synth.tableau(20, 10, 100 + k) for k in range(5)]
    Ms = np.stack([t[0] for t in tabs]); Bs = np.stack([t[1] for t in tabs])
    Ms[2, :10, 4] = -1.0                          # column 4 of LP 2 can grow without bound
    batch = lp.TableauBatch.from_arrays(Ms, Bs)
    st, npv = batch.solve()
    for k in range(5):
        M, b = Ms[k].copy(), Bs[k].copy()
        so, no, _ = oracle.solve(M, b)
        assert st[k] == so and npv[k] == no
        assert np.array_equal(batch.download(k)[0], M)
    assert st[2] == lp.capi.MI_UNBOUNDED


def _same_bits_nan_aware(G, M):
    nan_g, nan_m = np.isnan(G), np.isnan(M)
    return np.array_equal(nan_g, nan_m) and np.array_equal(G[~nan_g].view(np.int64), M[~nan_m].view(np.int64))


def test_batch_with_nonfinite_member_falls_back_to_dense(batch_mode):
    """One LP of a batch overflows to inf during its pivots: the reference would spread NaNs
    over basic columns (x - inf*0), which the compact representation does not store, so the whole
    batch finishes on the dense tableaux -- every member still equals the oracle."""
    n, m = 12, 8
    tabs = [lp.synth.tableau(n, m, 300 + k) for k in range(4)]
    Ms = np.stack([t[0] for t in tabs]); Bs = np.stack([t[1] for t in tabs])
    Ms[1, :m, :n] *= 1e200                         # products overflow
    Ms[1, :m, -1] *= 1e150
    Ms[1, 2, 3] = 1e308
    batch = lp.TableauBatch.from_arrays(Ms, Bs)
    st, npv = batch.solve(max_pivots=30)
    for k in range(4):
        M, b = Ms[k].copy(), Bs[k].copy()
        with np.errstate(all="ignore"):
            so, no, _ = oracle.solve(M, b, max_pivots=30)
        Mg, bg = batch.download(k)
        assert (st[k], npv[k]) == (so, no), k
        assert _same_bits_nan_aware(Mg, M) and np.array_equal(bg, b), k


def test_compact_shards_report_nonfinite_columns():
    """A compact column shard cannot reproduce NaNs in columns nobody stores: MI_NONFINITE."""
    import importlib
    import torch
    cp = importlib.import_module("linear-programming_amd.colpart")
    n, m = 40, 20
    M0, b0 = lp.synth.tableau(n, m, 5)
    M0[3, 7] = np.inf
    M0[m, 7] = -1e9                                # make column 7 the first entering column
    shards = []
    for r, (b, e) in enumerate(cp.partition(n, 2)):
        local = np.ascontiguousarray(np.concatenate([M0[:, b:e], M0[:, -1:]], axis=1))
        h = ctypes.c_void_p()
        lp.capi.check(lp.capi.lib().mi355x_tab_create(ctypes.byref(h), m + 1, local.shape[1],
                                                      local.ctypes.data_as(ctypes.c_void_p),
                                                      b0.ctypes.data_as(ctypes.c_void_p), 0), "create")
        lp.capi.check(lp.capi.lib().mi355x_tab_set_stream(
            h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "stream")
        cols = np.arange(b, e, dtype=np.int64)
        lp.capi.check(lp.capi.lib().mi355x_shard_set_compact(h, n + m, cols.ctypes.data_as(ctypes.c_void_p)),
                      "compact")
        shards.append(cp.Shard(torch, h, b, e, m + 1, 2, torch.device("cuda", 0)))
    tab = cp.ColumnPartitionedTableau(shards, cp.LocalComm(torch), cp.HipBackend(), block=8)
    st, npiv = tab.solve(check_every=4)
    assert (st, npiv) == (lp.capi.MI_NONFINITE, 0)
    cp.destroy_shards(shards)


# =========================================================================== column partition (config 5)
@pytest.mark.parametrize("compact", [False, True], ids=["dense-shards", "compact-shards"])
@pytest.mark.parametrize("n_shards,n,m", [(1, 60, 40), (2, 96, 64), (3, 100, 50), (8, 512, 256)])
@pytest.mark.parametrize("block", [1, 7, 16], ids=["per-pivot", "blocks-of-7", "blocks-of-16"])
def test_column_partition_logical_shards_bitwise(n_shards, n, m, compact, block):
    """One tableau as N column shards on ONE device (exchanges = local tensor ops with the
    collectives' semantics): same pivot sequence and same bits as the unpartitioned solve, with
    shards that hold fixed blocks of all columns and with compact shards (non-basic columns
    only, slots handed over at every pivot); updated after every pivot, or blocked (the same
    exchanges, every shard swept once per block, the solve ending inside a block)."""
    import importlib
    import torch
    cp = importlib.import_module("linear-programming_amd.colpart")
    seed = lp.synth.seed_for(5, n_shards)
    shards = cp.synthetic_shards(torch, n, m, seed, list(range(n_shards)), n_shards, 0, compact=compact)
    tab = cp.ColumnPartitionedTableau(shards, cp.LocalComm(torch), cp.HipBackend(), block=block)
    st, npiv = tab.solve(check_every=32)
    M, b = lp.synth.tableau(n, m, seed)
    so, no, trace = oracle.solve(M, b, trace_cap=1 << 16)
    assert (st, npiv) == (so, no) == (oracle.OPTIMAL, no)
    parts = [cp.download_shard(sh) for sh in shards]
    if compact:
        got, bs = cp.assemble_compact(shards, n + m)
        assert np.array_equal(got.view(np.int64), M.view(np.int64)) and np.array_equal(bs, b)
        assert sum(p[0].shape[1] - 1 for p in parts) == n     # only non-basic columns are stored
    else:
        got = np.concatenate([p[0][:, :-1] for p in parts], axis=1)
        assert np.array_equal(got, M[:, :-1])
    for Ms, bs in parts:
        assert np.array_equal(Ms[:, -1], M[:, -1])            # every RHS copy
        assert np.array_equal(bs, b)                          # global column indices
    ec = np.empty(no, dtype=np.int64); cr = np.empty(no, dtype=np.int64); k = ctypes.c_int64(0)
    lp.capi.check(lp.capi.lib().mi355x_tab_trace(shards[-1].handle, ec.ctypes.data_as(ctypes.c_void_p),
                                                 cr.ctypes.data_as(ctypes.c_void_p), no, ctypes.byref(k)),
                  "trace")
    assert k.value == no and np.array_equal(np.stack([ec, cr], axis=1), trace)
    cp.destroy_shards(shards)


@pytest.mark.parametrize("mode,split,la_block", [(2, 0, 1), (2, 2, 1), (3, 2, 1), (2, 0, 0)],
                         ids=["by-size", "two-launch-step", "four-launch-step", "persistent-block"])
@pytest.mark.parametrize("world,n,m,cap", [(2, 300, 120, 0), (3, 700, 300, 0), (4, 1500, 700, 90)])
def test_column_partition_p2p_exchange_between_processes(world, n, m, cap, mode, split, la_block, tmp_path):
    """Exchange mode 2 between real OS processes: every rank owns one shard behind mi355x_colpart_*,
    there is NO communicator (RCCL is not touched), the ranks map each other's fine-grained exchange
    buffers through IPC handles, and the per-pivot loop -- blind enqueue, no host in it -- runs in
    every process at its own pace: the processes' kernels execute concurrently on the (one) GPU and
    meet only through the self-validating granules they write into each other's buffers.  Every
    rank must end with the oracle's status, pivot sequence, basis and RHS column.  (What this box
    cannot show is the visibility of such stores across xGMI; the protocol, its parities and tags
    and its freedom from deadlock under real asynchrony are what runs here.)
    two-launch-step: the multi-workgroup look-ahead step of large shards forced at these sizes, in
    the form a shard with its device to itself uses -- pricing pair, pairs, column and ratio
    partials as ONE kernel (k_shard_p2p_step), so that a consumer and the producer it waits for are
    the same kernel in different processes; four-launch-step: the same step as the separate kernels
    shards on one stream use; persistent-block (round 6): every rank's look-ahead of a whole block is ONE
    persistent launch (k_shard_la_block) -- the ranks' launches are co-resident on the GPU, wait for each
    other's pairs and column granules INSIDE the kernels, and must have run (la_blocks > 0, nothing lost)."""
    import os
    import socket
    import subprocess
    import sys
    from tests.helpers import ROOT
    seed = lp.synth.seed_for(5, 70 + world)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_colpart_p2p_worker.py"),
                                       str(tmp_path), str(n), str(m), str(seed), str(cap), str(mode), str(split), str(la_block)],
                                      env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=600) == 0
    M, b = lp.synth.tableau(n, m, seed)
    so, no, trace = oracle.solve(M, b, max_pivots=cap, trace_cap=1 << 14)
    for r in range(world):
        res = np.load(os.path.join(tmp_path, "rank%d.npz" % r))
        assert (int(res["status"]), int(res["npiv"])) == (so, no), r
        assert np.array_equal(res["trace"], trace), r
        assert np.array_equal(res["basis"], b), r
        assert np.array_equal(res["last_col"].view(np.int64), M[:, -1].view(np.int64)), r
        if la_block == 0:
            assert int(res["la_blocks"]) > 0 and int(res["la_losses"]) == 0, (r, int(res["la_blocks"]), int(res["la_losses"]))
        else:
            assert int(res["la_blocks"]) == 0, r


@pytest.mark.parametrize("world,n,m,kind", [(2, 96, 64, "dense"), (3, 200, 90, "compact"),
                                            (2, 300, 120, "compact")])
def test_column_partition_multi_process(world, n, m, kind, tmp_path):
    """One OS process per shard, as in production (torch.distributed rendezvous, one handle per
    rank, real kernels); the ranks have to share the single GPU of the test box, so the two
    exchanges are staged through the host with gloo instead of RCCL."""
    import os
    import socket
    import subprocess
    import sys
    from tests.helpers import ROOT
    seed = lp.synth.seed_for(5, 40 + world)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_colpart_gpu_worker.py"),
                                       str(tmp_path), str(n), str(m), str(seed), "0", kind], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=600) == 0
    M, b = lp.synth.tableau(n, m, seed)
    so, no, _ = oracle.solve(M, b)
    res = [np.load(os.path.join(tmp_path, "rank%d.npz" % r)) for r in range(world)]
    got = np.zeros_like(M)
    for i, bc in enumerate(b):
        got[i, bc] = 1.0                                      # basic columns (stored nowhere if compact)
    for r in res:
        assert (int(r["status"]), int(r["npiv"])) == (so, no)
        assert np.array_equal(r["basis"], b) and np.array_equal(r["M"][:, -1], M[:, -1])
        got[:, r["cols"]] = r["M"][:, :-1]
    got[:, -1] = M[:, -1]
    assert np.array_equal(got, M)


def test_config5_full_size_dense_and_compact_shards_agree():
    """BASELINE config 5 at full size (65536 vars x 32768 constraints: 3.2e9 tableau entries,
    past 2^31, 25.8 GB dense / 17.2 GB compact) on one GPU: no CPU oracle can follow here, but the
    dense-shard and the compact-shard implementations are independent code paths over different
    layouts -- the first pivots must be the same pivots and leave the same RHS column, objective
    value and basis."""
    import importlib
    import torch
    cp = importlib.import_module("linear-programming_amd.colpart")
    n, m, K = 65536, 32768, 6
    seed = lp.synth.seed_for(5)
    out = []
    for compact in (False, True):
        sh = cp.synthetic_shards(torch, n, m, seed, [0], 1, 0, compact=compact)
        tab = cp.ColumnPartitionedTableau(sh, cp.LocalComm(torch), cp.HipBackend())
        st, npiv = tab.solve(max_pivots=K, check_every=K)
        assert (st, npiv) == (lp.capi.MI_MAX_PIVOTS, K)
        ec = np.empty(K, dtype=np.int64); cr = np.empty(K, dtype=np.int64); k = ctypes.c_int64(0)
        L = lp.capi.lib()
        lp.capi.check(L.mi355x_tab_trace(sh[0].handle, ec.ctypes.data_as(ctypes.c_void_p),
                                         cr.ctypes.data_as(ctypes.c_void_p), K, ctypes.byref(k)), "trace")
        rows, cols = ctypes.c_int64(0), ctypes.c_int64(0)
        L.mi355x_tab_shape(sh[0].handle, ctypes.byref(rows), ctypes.byref(cols), None)
        last_col = np.empty(rows.value); basis = np.empty(rows.value - 1, dtype=np.int64)
        lp.capi.check(L.mi355x_tab_download(sh[0].handle, None, basis.ctypes.data_as(ctypes.c_void_p), None,
                                            last_col.ctypes.data_as(ctypes.c_void_p)), "download")
        out.append((ec.copy(), cr.copy(), last_col, basis, cols.value))
        cp.destroy_shards(sh)
        torch.cuda.empty_cache()
    # third, independent path: the unpartitioned solver (split select, compact representation)
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, 0), "create")
    assert L.mi355x_tab_solve(h, 1, 1024.0, K, None) == lp.capi.MI_MAX_PIVOTS
    ec = np.empty(K, dtype=np.int64); cr = np.empty(K, dtype=np.int64)
    lp.capi.check(L.mi355x_tab_trace(h, ec.ctypes.data_as(ctypes.c_void_p), cr.ctypes.data_as(ctypes.c_void_p),
                                     K, ctypes.byref(k)), "trace")
    last_col = np.empty(m + 1); basis = np.empty(m, dtype=np.int64)
    lp.capi.check(L.mi355x_tab_download(h, None, basis.ctypes.data_as(ctypes.c_void_p), None,
                                        last_col.ctypes.data_as(ctypes.c_void_p)), "download")
    L.mi355x_tab_destroy(h)
    (e0, c0, r0, b0, w0), (e1, c1, r1, b1, w1) = out
    assert np.array_equal(ec, e0) and np.array_equal(cr, c0)
    assert np.array_equal(last_col.view(np.int64), r0.view(np.int64)) and np.array_equal(basis, b0)
    assert (w0, w1) == (n + m + 1, n + 1)
    assert np.array_equal(e0, e1) and np.array_equal(c0, c1)
    assert np.array_equal(r0.view(np.int64), r1.view(np.int64)) and np.array_equal(b0, b1)
    assert len(set(e0.tolist())) == K and (e0 < n).all()      # structural columns entered


@pytest.mark.parametrize("block", [1, 4, 16])
def test_column_partition_pivot_cap_and_unbounded(block):
    import importlib
    import torch
    cp = importlib.import_module("linear-programming_amd.colpart")
    n, m, seed = 80, 40, lp.synth.seed_for(5, 99)
    shards = cp.synthetic_shards(torch, n, m, seed, [0, 1, 2], 3, 0)
    tab = cp.ColumnPartitionedTableau(shards, cp.LocalComm(torch), cp.HipBackend(), block=block)
    st, npiv = tab.solve(max_pivots=11, check_every=4)
    assert (st, npiv) == (lp.capi.MI_MAX_PIVOTS, 11)
    M, b = lp.synth.tableau(n, m, seed)
    oracle.solve(M, b, max_pivots=11)
    got = np.concatenate([cp.download_shard(sh)[0][:, :-1] for sh in shards], axis=1)
    assert np.array_equal(got, M[:, :-1])
    cp.destroy_shards(shards)


# =========================================================================== synthetic inputs
@pytest.mark.parametrize("n,m", [(8, 4), (130, 70), (1024, 512)])
def test_device_generator_matches_numpy(n, m):
    """k_synth_fill writes the same doubles as linear-programming_amd/synth.py."""
    h = ctypes.c_void_p()
    seed = lp.synth.seed_for(3, n)
    lp.capi.check(lp.capi.lib().mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, 0),
                  "create_synthetic")
    t = lp.Tableau(None, lp.Problem(type="max"), None, None, n + m, m, {}, _handle=h)
    M, b = lp.synth.tableau(n, m, seed)
    assert np.array_equal(t.matrix, M) and np.array_equal(t.basis_columns, b)


# =========================================================================== full size
def test_config3_first_pivots_bitwise_and_full_solve_properties():
    """BASELINE config 3 (8192 vars x 4096 constraints, 4097 x 12289 f64 tableau).
    (a) the first 24 pivots are bit-identical to the oracle; (b) the full solve terminates
    optimal and satisfies size-independent properties: dual feasibility of the objective row,
    primal feasibility, exactly-unit basic columns, objective == c'x from the original data."""
    n, m = 8192, 4096
    seed = lp.synth.seed_for(3)
    h = ctypes.c_void_p()
    lp.capi.check(lp.capi.lib().mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, 0),
                  "create_synthetic")
    t = lp.Tableau(None, lp.Problem(type="max"), None, None, n + m, m, {}, _handle=h)
    K = 24
    M, b = lp.synth.tableau(n, m, seed)
    A, rhs, c = M[:m, :n].copy(), M[:m, -1].copy(), -M[m, :n].copy()
    st, npiv, trace = oracle.solve(M, b, max_pivots=K, trace_cap=K, omp=True)
    assert (st, npiv) == (oracle.MAX_PIVOTS, K)
    with pytest.raises(lp.SolverError):
        lp.n_solve_tableau(t, max_pivots=K)
    assert np.array_equal(t.pivot_trace(), trace)
    assert np.array_equal(t.matrix, M)
    del M
    lp.n_solve_tableau(t)
    Mf, bf = t.matrix, t.basis_columns
    eps = oracle.EPSILON
    assert Mf[m, :n + m].min() >= -128 * eps                    # nothing left to price
    assert Mf[:m, -1].min() >= -1e-9                            # primal feasible
    assert len(set(bf.tolist())) == m
    cols = Mf[:, bf]                                            # basic columns are unit vectors
    assert np.array_equal(cols[:m], np.eye(m)) and not cols[m].any()
    x = np.zeros(n + m)
    x[bf] = Mf[:m, -1]
    obj = Mf[m, -1]
    assert abs(c @ x[:n] - obj) <= 1e-10 * abs(obj)
    assert (A @ x[:n] - rhs).max() <= 1e-8 * np.abs(rhs).max()


# =========================================================================== :devices through the solver hook
@pytest.mark.parametrize("devices", [2, 3, 8])
def test_solver_hook_devices_keyword(devices):
    """(solve-problem problem :devices n): the call sequence of the Lisp glue's
    solve-column-partitioned (create -> solve -> download -> destroy) through the Python mirror.
    Single-phase AND two-phase problems go through the column partition (logical shards on this
    one GPU) and end with the same solution object as on one device."""
    p = lp.Problem(type="max", vars=["x", "y", "z"], objective_var="w",          # README.md:43-47
                   objective_func=[("x", 1), ("y", 4), ("z", 3)],
                   constraints=[("<=", [("x", 2), ("y", 1)], 8), ("<=", [("y", 1), ("z", 1)], 7)])
    one = lp.solve_problem(p, native=False)
    many = lp.solve_problem(p, devices=devices)
    assert lp.solution_variable(many, "w") == 28.5 and lp.solution_variable(many, "x") == 0.5
    assert np.array_equal(many.matrix.view(np.int64), one.matrix.view(np.int64))
    assert np.array_equal(many.basis_columns, one.basis_columns)
    for v in ("x", "y", "z"):
        assert lp.solution_reduced_cost(many, v) == lp.solution_reduced_cost(one, v)
    # a larger single-phase LP: same bits as the single-device solve
    rng = np.random.default_rng(devices)
    n, m = 60, 35
    names = ["v%d" % i for i in range(n)]
    A = rng.uniform(0.1, 1.0, (m, n))
    cons = [("<=", list(zip(names, A[i].tolist())), float(rng.uniform(5, 9))) for i in range(m)]
    q = lp.Problem(type="max", vars=names, objective_var="obj",
                   objective_func=list(zip(names, rng.uniform(0.5, 1.5, n).tolist())), constraints=cons)
    one, many = lp.solve_problem(q, native=False), lp.solve_problem(q, devices=devices)
    assert np.array_equal(many.matrix.view(np.int64), one.matrix.view(np.int64))
    assert lp.solution_objective_value(many) == lp.solution_objective_value(one)
    # two-phase problem (>= and = rows): phase 1, hand-over and phase 2 on the partition, same bits
    from tests.helpers import random_mixed_problem
    r = random_mixed_problem(lp, 12, 5, 3, 2, 77)
    one, many = lp.solve_problem(r, native=False), lp.solve_problem(r, devices=devices)
    assert np.array_equal(many.matrix.view(np.int64), one.matrix.view(np.int64))
    assert np.array_equal(many.basis_columns, one.basis_columns) and many.n_pivots == one.n_pivots
    assert lp.solution_objective_value(many) == lp.solution_objective_value(one)


def test_solver_hook_devices_falls_back_when_the_tableau_overflows():
    """Entries over hundreds of orders of magnitude: the partitioned solve meets a non-finite entering column
    (round 6: it then moves to dense shards and goes on; a NaN quotient still ends it with MI_NONFINITE, and the
    hook solves the untouched tableau on one device) -- same result as without the keyword, whatever the
    reference does with the infinities."""
    import ctypes
    found = 0
    for seed in range(40):
        rng = np.random.default_rng(seed)
        n, m = 20, 8
        names = ["v%d" % i for i in range(n)]
        mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(-300, 161, shape)   # noqa: E731
        A = mag((m, n))
        cons = [("<=", list(zip(names, A[i].tolist())), float(mag(1)[0])) for i in range(m)]
        q = lp.Problem(type="max", vars=names, objective_var="obj",
                       objective_func=list(zip(names, mag(n).tolist())), constraints=cons)
        tab = lp.build_tableau(q, q)
        L = lp.capi.lib()
        h = ctypes.c_void_p()
        M, b = tab.matrix, tab.basis_columns
        lp.capi.check(L.mi355x_colpart_create(ctypes.byref(h), M.shape[0], M.shape[1], M.ctypes.data_as(ctypes.c_void_p),
                                              b.ctypes.data_as(ctypes.c_void_p), 3), "create")
        rc = L.mi355x_colpart_solve(h, 1, 1024.0, 0, None)
        went_dense = not L.mi355x_colpart_is_compact(h)         # (round 6: a non-finite entering column moves the
        L.mi355x_colpart_destroy(h)                               # tableau to dense shards instead of ending the solve)
        if rc != lp.capi.MI_NONFINITE and not went_dense:
            continue
        found += 1
        outcomes = []
        for kw in ({"native": False}, {"devices": 3}):
            try:
                s = lp.solve_problem(q, **kw)
                G = s.matrix
                outcomes.append(("solved", G[~np.isnan(G)].view(np.int64).tolist(), np.isnan(G).tolist(), s.basis_columns.tolist()))
            except lp.SolverError as e:
                outcomes.append((type(e).__name__,))
        assert outcomes[0] == outcomes[1], seed
        if found >= 3:
            break
    assert found >= 1, "no overflowing problem among the seeds: the fallback was not exercised"
