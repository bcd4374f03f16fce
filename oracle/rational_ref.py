"""ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

Exact-rational (``fractions.Fraction``) pure-Python restatement of the parts
of the reference that surround the hot path, used to pin the C oracle and the
HIP path against the reference's own known-answer tests (which are written in
rationals):

* ``build_tableau``    -- src/simplex.lisp:142-328 (problem -> tableau(s))
* ``pivot`` / ``price`` / ``ratio`` / ``solve`` / ``solve_two_phase``
                       -- src/simplex.lisp:337-461 in exact arithmetic
                          (the reference's `rational` dispatch of fp=, fp<, fp>,
                          src/utils.lisp:88,102)
* ``tableau_variable`` / ``tableau_reduced_cost`` / ``objective_value``
                       -- src/simplex.lisp:74-120

Small cases only (pure-Python loops).  Nothing in the product package imports
this module; only tests/ do.

A *problem* here is the reference's already-parsed ``problem`` struct
(src/problem.lisp:45-53) as a dict::

    {"type": "max" | "min",
     "vars": ["x", "y", ...],                     # column order
     "objective_var": "w",
     "objective": [["x", 1], ["y", 4], ...],      # alist (var . coef)
     "bounds": [["x", lb_or_None, ub_or_None], ...],   # (var . (lb . ub))
     "constraints": [["<=", [["x", 2], ["y", 1]], 8], ...]}   # rhs >= 0 for <=, >=

The DSL parser (src/problem.lisp:73-205) is upstream of the hot path and is
not restated.
"""
from fractions import Fraction


class Unbounded(Exception):
    """unbounded-problem-error, src/conditions.lisp:47."""


class Infeasible(Exception):
    """infeasible-problem-error, src/conditions.lisp:55."""


class Tableau:
    """The `tableau` struct, src/simplex.lisp:48-58."""

    def __init__(self, matrix, basis, var_count, constraint_count, var_mapping, is_max,
                 objective_var=None):
        self.matrix = matrix            # list of rows, rows = constraint_count + 1
        self.basis = basis              # list, length constraint_count
        self.var_count = var_count
        self.constraint_count = constraint_count
        self.var_mapping = var_mapping  # var -> (kind, col[, offset])
        self.is_max = is_max            # problem-type of instance-problem
        self.objective_var = objective_var

    def copy(self):
        return Tableau([row[:] for row in self.matrix], self.basis[:], self.var_count,
                       self.constraint_count, self.var_mapping, self.is_max, self.objective_var)


def _F(x):
    return x if isinstance(x, Fraction) else Fraction(x)


def build_tableau(problem):
    """src/simplex.lisp:142-328.  Returns a Tableau, or (art_tableau, main_tableau)."""
    pvars = list(problem["vars"])
    n = len(pvars)
    is_max = problem["type"] == "max"
    objective = [(v, _F(c)) for v, c in problem["objective"]]
    bounds = {b[0]: (None if b[1] is None else _F(b[1]), None if b[2] is None else _F(b[2]))
              for b in problem.get("bounds", [])}
    constraints = [(op, [(v, _F(c)) for v, c in expr], _F(rhs))
                   for op, expr, rhs in problem.get("constraints", [])]
    objvar = problem.get("objective_var")
    mappings = {}

    if not constraints:                                                  # :153-186
        mat = [[Fraction(0)] * (n + 1) for _ in range(n + 1)]
        basis = [0] * n
        objective_value = Fraction(0)
        objd = dict(objective)
        for i, var in enumerate(pvars):
            coef = objd[var]
            lb, ub = bounds.get(var, (None, None)) if var in bounds else (None, None)
            basis[i] = i
            mat[i][i] = Fraction(1)
            if (0 <= coef) == is_max:
                if ub is None:
                    raise Unbounded()
                mappings[var] = ("positive", i, ub)
                objective_value += coef * ub
            else:
                # NB: a variable absent from var-bounds has (cadr nil) = NIL here too
                if var not in bounds or lb is None:
                    raise Unbounded()
                mappings[var] = ("positive", i, lb)
                objective_value += coef * lb
        mat[n][n] = objective_value
        return Tableau(mat, basis, n, n, mappings, is_max, objvar)

    ncols_vars = n                                                       # :189-212
    column = 0
    for var in pvars:
        if var not in bounds:
            mappings[var] = ("positive", column, Fraction(0))
        else:
            lb, ub = bounds[var]
            if lb is not None and ub is not None:
                if 0 <= ub:
                    constraints.insert(0, ("<=", [(var, Fraction(1))], ub))
                else:
                    constraints.insert(0, (">=", [(var, Fraction(1))], -ub))
                mappings[var] = ("positive", column, lb)
            elif lb is not None:
                mappings[var] = ("positive", column, lb)
            elif ub is not None:
                mappings[var] = ("negative", column, ub)
            else:
                mappings[var] = ("signed", column)
                column += 1
                ncols_vars += 1
        column += 1

    m = len(constraints)                                                 # :214-221
    num_slack = sum(1 for c in constraints if c[0] != "=")
    num_cols = ncols_vars + num_slack + 1
    mat = [[Fraction(0)] * num_cols for _ in range(m + 1)]
    basis = [0] * m
    art_rows = []                                   # pushed => most recent first
    col_offset = 0
    for row, (op, expr, rhs) in enumerate(constraints):                  # :223-268
        mat[row][num_cols - 1] = rhs
        for var, coef in expr:
            mp = mappings[var]
            if mp[0] == "positive":
                mat[row][mp[1]] = coef
                mat[row][num_cols - 1] -= coef * mp[2]
            elif mp[0] == "negative":
                mat[row][mp[1]] = -coef
                mat[row][num_cols - 1] -= coef * mp[2]
            else:
                mat[row][mp[1]] = coef
                mat[row][mp[1] + 1] = -coef
        if mat[row][num_cols - 1] < 0:                                   # :243-252
            mat[row] = [-x for x in mat[row]]
            op = {"<=": ">=", ">=": "<=", "=": "="}[op]
        if op == "<=":                                                   # :254-265
            mat[row][ncols_vars + col_offset] = Fraction(1)
            basis[row] = ncols_vars + col_offset
            col_offset += 1
        elif op == ">=":
            art_rows.insert(0, row)
            mat[row][ncols_vars + col_offset] = Fraction(-1)
            basis[row] = num_cols
            col_offset += 1
        elif op == "=":
            art_rows.insert(0, row)
            basis[row] = num_cols
        else:
            raise ValueError("not a valid constraint equation: %r" % (op,))
    for var, coef in objective:                                          # :270-283
        mp = mappings[var]
        if mp[0] == "positive":
            mat[m][mp[1]] = -coef
            mat[m][num_cols - 1] += coef * mp[2]
        elif mp[0] == "negative":
            mat[m][mp[1]] = coef
            mat[m][num_cols - 1] += coef * mp[2]
        else:
            mat[m][mp[1]] = -coef
            mat[m][mp[1] + 1] = coef
    main = Tableau(mat, basis, num_cols - 1, m, mappings, is_max, objvar)
    if not art_rows:
        return main
    num_art = len(art_rows)                                              # :292-325
    nac = num_cols + num_art
    amat = [[Fraction(0)] * nac for _ in range(m + 1)]
    abasis = basis[:]
    for i, row in enumerate(art_rows):
        abasis[row] = num_cols - 1 + i
        amat[row][num_cols - 1 + i] = Fraction(1)
    for c in range(num_cols - 1):
        s = Fraction(0)
        for r in range(m):
            amat[r][c] = mat[r][c]
            if r in art_rows:
                s += amat[r][c]
        amat[m][c] = s
    s = Fraction(0)
    for r in range(m):
        amat[r][nac - 1] = mat[r][num_cols - 1]
        if r in art_rows:
            s += amat[r][nac - 1]
    amat[m][nac - 1] = s
    art = Tableau(amat, abasis, num_cols - 1 + num_art, m, mappings, False, objvar)
    return art, main


def price(t):
    """find-entering-column, :362-379 (rational dispatch: exact compare with 0)."""
    obj = t.matrix[t.constraint_count]
    if t.var_count == 0:
        return None
    best = 0
    for i in range(1, t.var_count):
        if (obj[i] < obj[best]) if t.is_max else (obj[i] > obj[best]):
            best = i
    ok = obj[best] < 0 if t.is_max else obj[best] > 0
    return best if ok else None


def ratio(t, ec):
    """find-pivoting-row, :382-389."""
    row, bestq = None, None
    for i in range(t.constraint_count):
        a = t.matrix[i][ec]
        if 0 < a:
            q = t.matrix[i][t.var_count] / a
            if row is None or q < bestq:
                row, bestq = i, q
    return row


def pivot(t, ec, cr):
    """n-pivot-row, :337-359."""
    M = t.matrix
    rs = M[cr][ec]
    M[cr] = [x / rs for x in M[cr]]
    for r in range(len(M)):
        if r != cr:
            s = M[r][ec]
            M[r] = [x - s * p for x, p in zip(M[r], M[cr])]
    t.basis[cr] = ec
    return t


def solve(t, trace=None):
    """n-solve-tableau single phase, :453-461."""
    while True:
        ec = price(t)
        if ec is None:
            return t
        cr = ratio(t, ec)
        if cr is None:
            raise Unbounded()
        if trace is not None:
            trace.append((ec, cr))
        pivot(t, ec, cr)


def solve_two_phase(art, main, trace=None):
    """n-solve-tableau two-phase branch, :402-452."""
    solve(art, trace)
    if objective_value(art) != 0:
        raise Infeasible()
    nv, nav, m = main.var_count, art.var_count, main.constraint_count
    for i in range(m):
        if art.basis[i] >= nv:
            if art.matrix[i][nav] != 0:
                raise RuntimeError("Artificial variable still non-zero")
            new_col = None
            for j in range(nv):
                if art.matrix[i][j] != 0 and all(b != j for b in art.basis):
                    new_col = j
                    break
            if new_col is None:
                raise RuntimeError("Artificial variable still in basis and cannot be replaced")
            pivot(art, new_col, i)
    for r in range(m):
        for c in range(nv):
            main.matrix[r][c] = art.matrix[r][c]
        main.matrix[r][nv] = art.matrix[r][nav]
    for i, bc in enumerate(art.basis):
        main.basis[i] = bc
        scale = main.matrix[m][bc]
        if scale != 0:
            main.matrix[m] = [x - scale * y for x, y in zip(main.matrix[m], main.matrix[i])]
    return solve(main, trace)


def solve_any(tabs, trace=None):
    """solve-tableau on whatever build_tableau returned (:391-397)."""
    if isinstance(tabs, tuple):
        return solve_two_phase(tabs[0].copy(), tabs[1].copy(), trace)
    return solve(tabs.copy(), trace)


def objective_value(t):
    """tableau-objective-value, :74-78."""
    return t.matrix[t.constraint_count][t.var_count]


def _basic_value(t, col):
    if col in t.basis:
        return t.matrix[t.basis.index(col)][t.var_count]   # `position` = first match
    return 0


def tableau_variable(t, var):
    """tableau-variable, :81-107."""
    if t.objective_var is not None and var == t.objective_var:
        return objective_value(t)
    mp = t.var_mapping.get(var)
    if mp is None:
        raise KeyError("%s is not a variable in the tableau" % var)
    if mp[0] == "positive":
        return mp[2] + _basic_value(t, mp[1])
    if mp[0] == "negative":
        return mp[2] + (-_basic_value(t, mp[1]))
    return _basic_value(t, mp[1]) - _basic_value(t, mp[1] + 1)


def tableau_reduced_cost(t, var):
    """tableau-reduced-cost, :111-120."""
    mp = t.var_mapping.get(var)
    if mp is None:
        raise KeyError("%s is not a variable in the tableau" % var)
    if mp[0] != "positive":
        raise ValueError("%s has no lower bound" % var)
    return t.matrix[t.constraint_count][mp[1]]
