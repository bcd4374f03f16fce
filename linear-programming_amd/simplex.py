"""Host-side mirror of the reference's `linear-programming/simplex` package
(src/simplex.lisp) over the MI355X C ABI.

Same names (Lisp `n-pivot-row` -> `n_pivot_row`), same argument meaning, same error
behaviour, so the parity tests read like t/simplex.lisp.  Every numeric step of the hot path
-- pricing, ratio test, rank-1 update, the solve loops, the two-phase hand-over -- runs in
``libmi355x_simplex.so`` on the GPU; this module only moves arrays across the boundary and
does the O(n) bookkeeping that stays on the host in the Lisp glue as well
(`build-tableau` before the boundary, `tableau-variable` read-back after it).

The tableau lives in HBM behind ``Tableau._h``; ``Tableau.matrix`` / ``basis_columns`` are
host copies refreshed lazily after every device-side mutation.
"""
import ctypes

import numpy as np

from . import capi
from .conditions import (InfeasibleProblemError, ParsingError, SolverError,
                         UnboundedProblemError, UnsupportedConstraintError)
from .problem import Problem

_i64p = ctypes.POINTER(ctypes.c_int64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Tableau:
    """The `tableau` struct (src/simplex.lisp:48-58): problem, instance-problem, matrix,
    basis-columns, var-count, constraint-count, var-mapping, fp-tolerance-factor."""

    def __init__(self, problem, instance_problem, matrix, basis_columns, var_count,
                 constraint_count, var_mapping, fp_tolerance_factor=1024, device=0, _handle=None):
        self.problem = problem
        self.instance_problem = instance_problem
        self.var_count = int(var_count)
        self.constraint_count = int(constraint_count)
        self.var_mapping = var_mapping
        self.fp_tolerance_factor = fp_tolerance_factor
        self.device = device
        self._matrix = None
        self._basis = None
        self._stale = True
        self._light = None
        self._handle = None
        if _handle is not None:
            self._handle = _handle
        else:
            matrix = np.ascontiguousarray(matrix, dtype=np.float64)
            basis = np.ascontiguousarray(basis_columns, dtype=np.int64)
            if matrix.shape != (self.constraint_count + 1, self.var_count + 1):
                raise ValueError("matrix shape %r does not match counts" % (matrix.shape,))
            if basis.shape != (self.constraint_count,):
                raise ValueError("basis length %r does not match constraint count" % (basis.shape,))
            # host arrays only; the upload happens at the first device operation
            self._matrix, self._basis, self._stale = matrix.copy(), basis.copy(), False

    # -- device <-> host
    @property
    def _h(self):
        """The device handle (uploads the host arrays on first use; no CPU fallback)."""
        if self._handle is None:
            h = ctypes.c_void_p()
            capi.check(capi.lib().mi355x_tab_create(
                ctypes.byref(h), self._matrix.shape[0], self._matrix.shape[1], _ptr(self._matrix),
                _ptr(self._basis) if self._basis.size else None, self.device), "mi355x_tab_create")
            self._handle = h
        return self._handle

    def _refresh(self):
        if self._stale:
            R, C = self.constraint_count + 1, self.var_count + 1
            M = np.empty((R, C), dtype=np.float64)
            b = np.empty(max(R - 1, 0), dtype=np.int64)
            capi.check(capi.lib().mi355x_tab_download(self._h, _ptr(M), _ptr(b) if b.size else None,
                                                      None, None), "mi355x_tab_download")
            self._matrix, self._basis, self._stale = M, b, False

    def _touch(self):
        self._stale = True
        self._light = None

    def _readback(self):
        """(objective row, RHS column, basis) -- all the read-back functions need
        (src/simplex.lisp:74-120).  Served from the host copy when it is current, otherwise
        fetched alone (mi355x_tab_download without the matrix), never the whole tableau."""
        if not self._stale:
            return self._matrix[-1], self._matrix[:, -1], self._basis
        if getattr(self, "_light", None) is None:
            R, C = self.constraint_count + 1, self.var_count + 1
            row, col = np.empty(C), np.empty(R)
            b = np.empty(max(R - 1, 0), dtype=np.int64)
            capi.check(capi.lib().mi355x_tab_download(self._h, None, _ptr(b) if b.size else None,
                                                      _ptr(row), _ptr(col)), "mi355x_tab_download")
            self._light = (row, col, b)
        return self._light

    @property
    def matrix(self):
        """tableau-matrix (host copy of the HBM-resident matrix)."""
        self._refresh()
        return self._matrix

    @property
    def basis_columns(self):
        """tableau-basis-columns."""
        self._refresh()
        return self._basis

    @property
    def is_max(self):
        return self.instance_problem.type == "max"

    def pivot_trace(self, cap=1 << 20):
        """(entering column, row) of every pivot made on this tableau since it was uploaded."""
        n = ctypes.c_int64(0)
        ec = np.empty(cap, dtype=np.int64)
        cr = np.empty(cap, dtype=np.int64)
        capi.check(capi.lib().mi355x_tab_trace(self._h, _ptr(ec), _ptr(cr), cap, ctypes.byref(n)),
                   "mi355x_tab_trace")
        k = min(n.value, cap)
        return np.stack([ec[:k], cr[:k]], axis=1)

    def __del__(self):
        h = getattr(self, "_handle", None)
        self._handle = None
        if h:
            try:
                capi.lib().mi355x_tab_destroy(h)
            except Exception:
                pass


def copy_tableau(tableau):
    """copy-tableau (src/simplex.lisp:61-71): deep-copies matrix and basis (device to device),
    shares everything else."""
    h = ctypes.c_void_p()
    capi.check(capi.lib().mi355x_tab_copy(ctypes.byref(h), tableau._h), "mi355x_tab_copy")
    return Tableau(tableau.problem, tableau.instance_problem, None, None, tableau.var_count,
                   tableau.constraint_count, tableau.var_mapping, tableau.fp_tolerance_factor,
                   tableau.device, _handle=h)


# ------------------------------------------------------------------ read-back (host, O(n))
def tableau_objective_value(tableau):
    """tableau-objective-value (src/simplex.lisp:74-78)."""
    return float(tableau._readback()[0][tableau.var_count])


def _basic_value(tableau, col):
    _, rhs, basis = tableau._readback()
    pos = np.nonzero(basis == col)[0]                       # `position`: first match
    return float(rhs[pos[0]]) if pos.size else 0.0


def tableau_variable(tableau, var):
    """tableau-variable (src/simplex.lisp:81-107)."""
    if var == tableau.instance_problem.objective_var:
        return tableau_objective_value(tableau)
    mapping = tableau.var_mapping.get(var)
    if mapping is None:
        raise KeyError("%s is not a variable in the tableau" % (var,))
    kind = mapping[0]
    if kind == "positive":
        return mapping[2] + _basic_value(tableau, mapping[1])
    if kind == "negative":
        return mapping[2] + (-_basic_value(tableau, mapping[1]))
    return _basic_value(tableau, mapping[1]) - _basic_value(tableau, mapping[1] + 1)


def tableau_reduced_cost(tableau, var):
    """tableau-reduced-cost (src/simplex.lisp:111-120)."""
    mapping = tableau.var_mapping.get(var)
    if mapping is None:
        raise KeyError("%s is not a variable in the tableau" % (var,))
    if mapping[0] != "positive":
        raise ValueError("%s has no lower bound" % (var,))
    return float(tableau._readback()[0][mapping[1]])


def with_tableau_variables(var_list, tableau):
    """with-tableau-variables (src/simplex.lisp:125-139) as a function: the values of the given
    variables (a sequence of names, or a Problem: its objective variable and all its variables)
    read from the tableau, as a dict."""
    if isinstance(var_list, Problem):
        names = [var_list.objective_var] + list(var_list.vars)
    else:
        names = list(var_list)
    return {v: tableau_variable(tableau, v) for v in names}


# ------------------------------------------------------------------ build-tableau (host)
def build_tableau(problem, instance_problem=None, fp_tolerance_factor=1024, device=0):
    """build-tableau (src/simplex.lisp:142-328) in double-float: returns a Tableau, or
    [art_tableau, main_tableau] when the trivial basis is infeasible.

    This is the producer of what crosses the boundary; in the Lisp deployment the reference's
    own build-tableau does this job and the glue converts the result to double-float."""
    if instance_problem is None:
        instance_problem = problem
    f = float
    constraints = [(op, list(expr), rhs) for op, expr, rhs in instance_problem.constraints]
    pvars = list(problem.vars)
    n = len(pvars)
    bounds = dict(problem.var_bounds)
    mappings = {}

    def mk(matrix, basis, var_count, ccount, inst):
        return Tableau(problem, inst, matrix, basis, var_count, ccount, mappings,
                       fp_tolerance_factor, device)

    if not constraints:                                                  # :153-186
        M = np.zeros((n + 1, n + 1))
        basis = np.arange(n, dtype=np.int64)
        objd = dict(problem.objective_func)
        is_max = problem.type == "max"
        objective_value = 0.0
        for i, var in enumerate(pvars):
            coef = objd[var]
            lb, ub = bounds.get(var, (None, None))
            M[i, i] = 1.0
            if (0 <= coef) == is_max:
                if ub is None:
                    raise UnboundedProblemError()
                mappings[var] = ("positive", i, f(ub))
                objective_value = objective_value + f(coef) * f(ub)
            else:
                if lb is None:
                    raise UnboundedProblemError()
                mappings[var] = ("positive", i, f(lb))
                objective_value = objective_value + f(coef) * f(lb)
        M[n, n] = objective_value
        return mk(M, basis, n, n, problem)

    ncv = n                                                              # :189-212
    column = 0
    for var in pvars:
        if var not in bounds:
            mappings[var] = ("positive", column, 0.0)
        else:
            lb, ub = bounds[var]
            if lb is not None and ub is not None:
                if 0 <= ub:
                    constraints.insert(0, ("<=", [(var, 1)], ub))
                else:
                    constraints.insert(0, (">=", [(var, 1)], -ub))
                mappings[var] = ("positive", column, f(lb))
            elif lb is not None:
                mappings[var] = ("positive", column, f(lb))
            elif ub is not None:
                mappings[var] = ("negative", column, f(ub))
            else:
                mappings[var] = ("signed", column)
                column += 1
                ncv += 1
        column += 1

    m = len(constraints)                                                 # :214-221
    num_slack = sum(1 for c in constraints if c[0] != "=")
    num_cols = ncv + num_slack + 1
    M = np.zeros((m + 1, num_cols))
    basis = np.zeros(m, dtype=np.int64)
    art_rows = []
    col_offset = 0
    for row, (op, expr, rhs) in enumerate(constraints):                  # :223-268
        M[row, num_cols - 1] = f(rhs)
        for var, coef in expr:
            mp = mappings[var]
            if mp[0] == "positive":
                M[row, mp[1]] = f(coef)
                M[row, num_cols - 1] -= f(coef) * mp[2]
            elif mp[0] == "negative":
                M[row, mp[1]] = -f(coef)
                M[row, num_cols - 1] -= f(coef) * mp[2]
            else:
                M[row, mp[1]] = f(coef)
                M[row, mp[1] + 1] = -f(coef)
        if M[row, num_cols - 1] < 0:                                     # :243-252
            M[row, :] = -M[row, :]
            op = {"<=": ">=", ">=": "<=", "=": "="}.get(op, op)
        if op == "<=":                                                   # :254-265
            M[row, ncv + col_offset] = 1.0
            basis[row] = ncv + col_offset
            col_offset += 1
        elif op == ">=":
            art_rows.insert(0, row)
            M[row, ncv + col_offset] = -1.0
            basis[row] = num_cols
            col_offset += 1
        elif op == "=":
            art_rows.insert(0, row)
            basis[row] = num_cols
        else:
            raise ParsingError("%r is not a valid constraint equation" % ((op, expr, rhs),))
    for var, coef in problem.objective_func:                             # :270-283
        mp = mappings[var]
        if mp[0] == "positive":
            M[m, mp[1]] = -f(coef)
            M[m, num_cols - 1] += f(coef) * mp[2]
        elif mp[0] == "negative":
            M[m, mp[1]] = f(coef)
            M[m, num_cols - 1] += f(coef) * mp[2]
        else:
            M[m, mp[1]] = -f(coef)
            M[m, mp[1] + 1] = f(coef)
    main = mk(M, basis, num_cols - 1, m, instance_problem)
    if not art_rows:
        return main
    num_art = len(art_rows)                                              # :292-325
    nac = num_cols + num_art
    A = np.zeros((m + 1, nac))
    abasis = basis.copy()
    for i, row in enumerate(art_rows):
        abasis[row] = num_cols - 1 + i
        A[row, num_cols - 1 + i] = 1.0
    A[:m, :num_cols - 1] = M[:m, :num_cols - 1]
    A[:m, nac - 1] = M[:m, num_cols - 1]
    for c in list(range(num_cols - 1)) + [nac - 1]:
        s = 0.0
        for r in range(m):                                               # same summation order
            if r in art_rows:
                s = s + A[r, c]
        A[m, c] = s
    art_problem = Problem(type="min", vars=list(problem.vars))           # artificial problem
    art = mk(A, abasis, num_cols - 1 + num_art, m, art_problem)
    return [art, main]


# ------------------------------------------------------------------ the hot path (device)
def find_entering_column(tableau):
    """find-entering-column (src/simplex.lisp:362-379): column index or None."""
    col = ctypes.c_int64(-1)
    capi.check(capi.lib().mi355x_tab_price(tableau._h, int(tableau.is_max),
                                           float(tableau.fp_tolerance_factor), ctypes.byref(col)),
               "mi355x_tab_price")
    return None if col.value < 0 else int(col.value)


def find_pivoting_row(tableau, entering_col):
    """find-pivoting-row (src/simplex.lisp:382-389): row index or None."""
    row = ctypes.c_int64(-1)
    capi.check(capi.lib().mi355x_tab_ratio(tableau._h, int(entering_col),
                                           float(tableau.fp_tolerance_factor), ctypes.byref(row)),
               "mi355x_tab_ratio")
    return None if row.value < 0 else int(row.value)


def n_pivot_row(tableau, entering_col, changing_row):
    """n-pivot-row (src/simplex.lisp:337-359): destructively applies a single pivot."""
    capi.check(capi.lib().mi355x_tab_pivot(tableau._h, int(entering_col), int(changing_row)),
               "mi355x_tab_pivot")
    tableau._touch()
    return tableau


def pivot_row(tableau, entering_col, changing_row):
    """pivot-row (src/simplex.lisp:333-335): non-destructive."""
    return n_pivot_row(copy_tableau(tableau), entering_col, changing_row)


def _raise_for(rc):
    if rc == capi.MI_UNBOUNDED:
        raise UnboundedProblemError()
    if rc == capi.MI_INFEASIBLE:
        raise InfeasibleProblemError()
    if rc == capi.MI_ART_NONZERO:
        raise SolverError("Artificial variable still non-zero")
    if rc == capi.MI_ART_STUCK:
        raise SolverError("Artificial variable still in basis and cannot be replaced")
    if rc == capi.MI_MAX_PIVOTS:
        raise SolverError("pivot cap reached")
    if rc == capi.MI_CANCELLED:
        raise SolveCancelled("solve cancelled (mi355x_tab_cancel)")


class SolveCancelled(SolverError):
    """cancel_solve() was called from another thread: the solve stopped between two chunks of
    launches.  The tableau holds whole pivots only and can be solved further.  (In Lisp the same
    situation is an interrupt honoured between two bounded foreign calls.)"""


def cancel_solve(tableau):
    """Ask the solve running on `tableau` (in another thread) -- or the next one -- to stop:
    mi355x_tab_cancel.  The reference's loop has no cap and no anti-cycling rule
    (src/simplex.lisp:453-461); in Lisp a cycling LP is interruptible, a foreign call is not."""
    for t in (tableau if isinstance(tableau, (list, tuple)) else [tableau]):
        capi.check(capi.lib().mi355x_tab_cancel(t._h), "mi355x_tab_cancel")


def chunk_pivots(rows, cols):
    """The glue's `chunk-pivots`: pivots per foreign call, about a tenth to half a second of GPU
    time at every size, so that a single-threaded host is back in its own code (where interrupts
    are served) a few times per second."""
    return max(1024, min(65536, (1 << 36) // max(1, rows * cols)))


def _solve_in_chunks(call, rows, cols, max_pivots, chunk=None):
    """The glue's `solve-in-chunks`: call(cap) -> (status, pivots of that call) until the status is
    something else than MI_MAX_PIVOTS or max_pivots (0 = no cap) are used up.  A solve continued
    call by call takes exactly the pivots of one long call.  (chunk: tests force a small one.)"""
    chunk, total = chunk or chunk_pivots(rows, cols), 0
    while True:
        cap = min(chunk, max_pivots - total) if max_pivots > 0 else chunk
        rc, k = call(cap)
        total += k
        if rc != capi.MI_MAX_PIVOTS or (max_pivots > 0 and total >= max_pivots):
            return rc, total


def _solve_two_phase_in_chunks(art, main, max_pivots=0, chunk=None):
    """The glue's `solve-two-phase-in-chunks`: phase 1 in chunks on the artificial tableau,
    mi355x_two_phase_handover, phase 2 in chunks on the main tableau -- the pivots and bits of
    mi355x_solve_two_phase without an unbounded foreign call; max_pivots caps the phases together."""
    L = capi.lib()
    n = ctypes.c_int64(0)
    f = float(main.fp_tolerance_factor)
    rows, cols = art.constraint_count + 1, art.var_count + 1

    def phase(t, is_max):
        def call(cap):
            rc = capi.check(L.mi355x_tab_solve(t._h, int(is_max), f, int(cap), ctypes.byref(n)), "mi355x_tab_solve")
            return rc, int(n.value)
        return call
    try:
        rc, n1 = _solve_in_chunks(phase(art, 0), rows, cols, int(max_pivots), chunk=chunk)
        main.n_pivots = (n1, 0)
        if rc != capi.MI_OPTIMAL:
            return rc
        rc = capi.check(L.mi355x_two_phase_handover(art._h, main._h, f, ctypes.byref(n)), "mi355x_two_phase_handover")
        n1 += int(n.value)
        main.n_pivots = (n1, 0)
        if rc != capi.MI_OK:
            return rc
        if max_pivots > 0 and n1 >= max_pivots:
            return capi.MI_MAX_PIVOTS
        rc, n2 = _solve_in_chunks(phase(main, main.is_max), rows, cols, max_pivots - n1 if max_pivots > 0 else 0, chunk=chunk)
        main.n_pivots = (n1, n2)
        return rc
    finally:
        art._touch()
        main._touch()


def n_solve_tableau(tableau, max_pivots=0, chunked=False, chunk=None):
    """n-solve-tableau (src/simplex.lisp:399-461): a Tableau (single phase) or a list
    [art, main] (two-phase).  Returns the solved (main) tableau.  chunked: bounded foreign calls
    (what the Lisp glue does, see chunk_pivots)."""
    if isinstance(tableau, (list, tuple)):
        art, main = tableau
        if not isinstance(art, Tableau) or not isinstance(main, Tableau):
            raise TypeError("expected tableaus")
        if chunked:
            rc = _solve_two_phase_in_chunks(art, main, int(max_pivots), chunk=chunk)
            _raise_for(rc)
            return main
        npv = (ctypes.c_int64 * 2)()
        rc = capi.check(capi.lib().mi355x_solve_two_phase(
            art._h, main._h, int(main.is_max), float(main.fp_tolerance_factor), npv),
            "mi355x_solve_two_phase")
        art._touch()
        main._touch()
        main.n_pivots = (int(npv[0]), int(npv[1]))
        _raise_for(rc)
        return main
    if not isinstance(tableau, Tableau):                    # (check-type tableau tableau) :454
        raise TypeError("%r is not a tableau" % (tableau,))
    n = ctypes.c_int64(0)

    def call(cap):
        rc = capi.check(capi.lib().mi355x_tab_solve(tableau._h, int(tableau.is_max),
                                                    float(tableau.fp_tolerance_factor),
                                                    int(cap), ctypes.byref(n)),
                        "mi355x_tab_solve")
        return rc, int(n.value)
    if chunked:
        rc, total = _solve_in_chunks(call, tableau.constraint_count + 1, tableau.var_count + 1, int(max_pivots), chunk=chunk)
    else:
        rc, total = call(max_pivots)
    tableau._touch()
    tableau.n_pivots = total
    _raise_for(rc)
    return tableau


def solve_tableau(tableau):
    """solve-tableau (src/simplex.lisp:391-397): leaves the argument(s) untouched."""
    if isinstance(tableau, (list, tuple)):
        return n_solve_tableau([copy_tableau(t) for t in tableau])
    return n_solve_tableau(copy_tableau(tableau))


# ------------------------------------------------------------------ the *solver* hook value
def _solve_column_partitioned(tableau, devices, max_pivots=0):
    """The glue's `solve-column-partitioned` (lisp/mi355x-simplex.lisp): the tableau of a
    single-phase problem split over `devices` GPUs behind mi355x_colpart_create / _solve /
    _download / _destroy.  Returns False when the library stopped with MI_NONFINITE (the tableau
    overflowed; compact shards cannot reproduce that case): the caller's tableau is untouched."""
    L = capi.lib()
    M, b = tableau.matrix, tableau.basis_columns
    h = ctypes.c_void_p()
    capi.check(L.mi355x_colpart_create(ctypes.byref(h), M.shape[0], M.shape[1], _ptr(M), _ptr(b),
                                       int(devices)), "mi355x_colpart_create")
    try:
        n = ctypes.c_int64(0)

        def call(cap):
            rc = capi.check(L.mi355x_colpart_solve(h, int(tableau.is_max), float(tableau.fp_tolerance_factor),
                                                   int(cap), ctypes.byref(n)), "mi355x_colpart_solve")
            return rc, int(n.value)
        rc, total = _solve_in_chunks(call, M.shape[0], M.shape[1], int(max_pivots))
        if rc == capi.MI_NONFINITE:
            return False
        tableau.n_pivots = total
        _raise_for(rc)
        G, bg = np.empty_like(M), np.empty_like(b)
        capi.check(L.mi355x_colpart_download(h, _ptr(G), _ptr(bg), None, None), "mi355x_colpart_download")
    finally:
        L.mi355x_colpart_destroy(h)
    # the solved arrays become the tableau's host copy; its device handle (if any) is stale
    old, tableau._handle = tableau._handle, None
    if old:
        L.mi355x_tab_destroy(old)
    tableau._matrix, tableau._basis, tableau._stale, tableau._light = G, bg, False, None
    return True


def _solve_two_phase_column_partitioned(art, main, devices):
    """The two-phase branch with the artificial tableau column-partitioned over `devices` GPUs
    (mi355x_colpart_create -> mi355x_colpart_solve_two_phase -> download of both tableaux).  The
    main tableau is never uploaded: the library takes its objective row only.  Returns False when
    the partitioned path does not apply (MI_UNSUPPORTED: the artificial basis is not a set of unit
    columns; MI_NONFINITE: the tableau overflowed) -- the caller's tableaux are untouched."""
    L = capi.lib()
    A, ab = art.matrix, art.basis_columns
    Mm = main.matrix
    h = ctypes.c_void_p()
    capi.check(L.mi355x_colpart_create(ctypes.byref(h), A.shape[0], A.shape[1], _ptr(A), _ptr(ab),
                                       int(devices)), "mi355x_colpart_create")
    hm = ctypes.c_void_p()
    try:
        obj = np.ascontiguousarray(Mm[-1])
        npv = (ctypes.c_int64 * 2)()
        rc = L.mi355x_colpart_solve_two_phase(h, int(Mm.shape[1]), _ptr(obj), int(main.is_max),
                                              float(main.fp_tolerance_factor), npv, ctypes.byref(hm))
        if rc in (capi.MI_UNSUPPORTED, capi.MI_NONFINITE):
            return False
        capi.check(rc, "mi355x_colpart_solve_two_phase")
        GA, ga = np.empty_like(A), np.empty_like(ab)
        capi.check(L.mi355x_colpart_download(h, _ptr(GA), _ptr(ga), None, None), "mi355x_colpart_download")
        GM, gm = None, None
        if hm:
            GM, gm = np.empty_like(Mm), np.empty_like(main.basis_columns)
            capi.check(L.mi355x_colpart_download(hm, _ptr(GM), _ptr(gm), None, None), "mi355x_colpart_download")
    finally:
        if hm:
            L.mi355x_colpart_destroy(hm)
        L.mi355x_colpart_destroy(h)
    for tab, G, g in ((art, GA, ga), (main, GM, gm)):
        if G is None:
            continue
        old, tab._handle = tab._handle, None
        if old:
            L.mi355x_tab_destroy(old)
        tab._matrix, tab._basis, tab._stale, tab._light = G, g, False, None
    main.n_pivots = (int(npv[0]), int(npv[1]))
    _raise_for(rc)
    return True


def _native_number(x):
    """The glue's `native-number-p`: numbers the library's double-float build-tableau treats exactly
    as generic arithmetic followed by a coerce would -- doubles and integers a double holds exactly
    (a Fraction, like a Lisp ratio, is combined exactly first and rounded afterwards)."""
    if isinstance(x, bool):
        return False
    if isinstance(x, int):
        return abs(x) <= 2 ** 53
    return isinstance(x, float)


def _native_numbers(problem):
    ok = all(_native_number(c) for _, c in problem.objective_func)
    ok = ok and all((lb is None or _native_number(lb)) and (ub is None or _native_number(ub))
                    for _, (lb, ub) in problem.var_bounds)
    return ok and all(all(_native_number(c) for _, c in e) and _native_number(rhs)
                      for _, e, rhs in problem.constraints)


def mi355x_simplex_solver(problem, fp_tolerance=1024, device=0, devices=1, max_pivots=0,
                          full_tableau=False, native="auto", chunk=None, **_ignored):
    """What the Lisp glue installs as `*solver*` (src/solver.lisp:39-56): takes a problem and
    backend keyword arguments, returns a solution object answering the four solution-*
    generics -- on the NATIVE route (default whenever the problem's numbers are floats / integers
    and neither full_tableau nor devices > 1 asks for the tableau itself; native=True forces it,
    native=False never takes it) a NativeSolution (the glue's MI355X-SOLUTION: problem marshalled
    through mi355x_problem_*, mi355x_simplex_solver_begin / _step in bounded chunks / _finish), on
    the build-tableau route a solved Tableau.  LP only: integer/binary variables are declined the way a backend must
    (unsupported-constraint-error, src/conditions.lisp:69-77); branch-and-bound
    (src/simplex.lisp:506-542) stays with the reference's own solver.  devices > 1: the tableau
    (single-phase problems) or the artificial tableau (two-phase problems: phase 1, the hand-over
    and phase 2 all stay partitioned) is column-partitioned over that many GPUs (logical shards of
    one GPU when fewer are visible); tableaux that overflow run on `device`."""
    if problem.integer_vars:
        raise UnsupportedConstraintError(("integer",) + tuple(problem.integer_vars),
                                         "mi355x-simplex")
    if native and not full_tableau and devices <= 1 and len(problem.vars) > 0 and \
            (native is True or _native_numbers(problem)):
        from .native import NativeProblem
        return NativeProblem(problem).solve_in_chunks(fp_tolerance=fp_tolerance, device=device,
                                                      max_pivots=max_pivots, chunk=chunk)
    tabs = build_tableau(problem, problem, fp_tolerance_factor=fp_tolerance, device=device)
    if devices > 1 and isinstance(tabs, Tableau) and _solve_column_partitioned(tabs, devices, max_pivots):
        return tabs
    if devices > 1 and isinstance(tabs, list) and _solve_two_phase_column_partitioned(tabs[0], tabs[1], devices):
        return tabs[1]
    return n_solve_tableau(tabs, max_pivots=max_pivots, chunked=True, chunk=chunk)


simplex_solver = mi355x_simplex_solver


def mi355x_solve_problems(problems, fp_tolerance=1024, device=0, devices=1, max_pivots=0, errorp=True, native=False):
    """The glue's `mi355x-solve-problems`: a LIST of problems -> the list of their solved tableaus,
    what [solve_problem(p) for p in problems] returns, with the independent LPs side by side on the
    GPU(s).  Single-phase problems are grouped by tableau shape and sense; a group of two or more
    is ONE multi-device batch (mi355x_multibatch_create / _solve in bounded chunks / _download per
    member / _destroy: `devices` sub-batches, no communication).  Two-phase problems
    (src/simplex.lisp:402-452) are grouped by the shapes of their two tableaux; a group of two or more
    is a pair of batches: phase 1 and phase 2 through mi355x_multibatch_solve in bounded chunks, the
    per-member feasibility test, drive-out pivots and hand-over through
    mi355x_multibatch_two_phase_handover.  Integer problems (declined) and problems alone in their
    group go through mi355x_simplex_solver one by one; max_pivots caps every phase.  A member without a solution does not abort the
    others: errorp False leaves the exception object in its place, errorp True raises the first
    one after every member has been attempted."""
    from .batch import MultiDeviceBatch
    if native == "many":
        # the whole list behind ONE job of the library (mi355x_simplex_solver_many_*): members come
        # back as NativeSolution objects (the glue's :native :many)
        from .native import solve_many
        results = solve_many(problems, fp_tolerance=fp_tolerance, devices=devices, max_pivots=max_pivots)
        if errorp:
            for r in results:
                if isinstance(r, Exception):
                    raise r
        return results
    results = [None] * len(problems)
    groups, groups2 = {}, {}

    def alone(k):
        try:
            # (native: what the one-problem hook may return for a member solved alone -- False keeps the
            # list homogeneous, every member a Tableau; "auto" lets lone members take the native route)
            results[k] = mi355x_simplex_solver(problems[k], fp_tolerance=fp_tolerance, device=device,
                                               max_pivots=max_pivots, native=native)
        except SolverError as e:
            results[k] = e

    for k, p in enumerate(problems):
        if p.integer_vars:
            alone(k)
            continue
        try:
            tabs = build_tableau(p, p, fp_tolerance_factor=fp_tolerance, device=device)
        except SolverError as e:                       # e.g. the unbounded no-constraint special case
            results[k] = e
            continue
        if isinstance(tabs, list):                     # two-phase: (art main), src/simplex.lisp:326-328
            art, main = tabs
            groups2.setdefault((art.matrix.shape, main.matrix.shape, main.is_max), []).append((k, art, main))
        elif not _unit_basis(tabs):
            alone(k)
        else:
            groups.setdefault((tabs.matrix.shape, tabs.is_max), []).append((k, tabs))
    def batch_in_chunks(mb, is_max, rows, cols):
        """The glue's `multibatch-solve-in-chunks`: bounded calls until no member is left at
        MI_MAX_PIVOTS (or max_pivots are used up); -> (statuses, pivots per member)."""
        chunk, done = chunk_pivots(rows, cols), 0
        total = np.zeros(mb.n_lps, dtype=np.int64)
        while True:
            cap = min(chunk, max_pivots - done) if max_pivots > 0 else chunk
            st, npv = mb.solve(is_max=is_max, fp_tolerance=fp_tolerance, max_pivots=cap)
            total += npv
            done += cap
            if (max_pivots > 0 and done >= max_pivots) or not (st == capi.MI_MAX_PIVOTS).any():
                return st, total

    def adopt(t, mb, q):
        G, gb = mb.download(q)
        old, t._handle = t._handle, None
        if old:
            capi.lib().mi355x_tab_destroy(old)
        t._matrix, t._basis, t._stale, t._light = G, gb, False, None

    for (shape, is_max), members in groups.items():
        if len(members) == 1:
            alone(members[0][0])
            continue
        rows, cols = shape
        mb = MultiDeviceBatch.from_arrays(np.stack([t.matrix for _, t in members]),
                                          np.stack([t.basis_columns for _, t in members]), n_devices=devices)
        st, npv = batch_in_chunks(mb, is_max, rows, cols)
        for q, (k, t) in enumerate(members):
            try:
                _raise_for(int(st[q]))
            except SolverError as e:
                results[k] = e
                continue
            adopt(t, mb, q)
            t.n_pivots = int(npv[q])
            results[k] = t
    # two-phase members of one shape: phase 1 in bounded calls on the batch of artificial tableaux, the
    # per-member step between the phases (feasibility test, drive-out pivots, hand-over:
    # mi355x_multibatch_two_phase_handover), phase 2 in bounded calls on the batch of main tableaux
    for (ashape, mshape, is_max), members in groups2.items():
        if len(members) == 1:
            alone(members[0][0])
            continue
        amb = MultiDeviceBatch.from_arrays(np.stack([a.matrix for _, a, _ in members]),
                                           np.stack([a.basis_columns for _, a, _ in members]), n_devices=devices)
        mmb = MultiDeviceBatch.from_arrays(np.stack([t.matrix for _, _, t in members]),
                                           np.stack([t.basis_columns for _, _, t in members]), n_devices=devices)
        st1, np1 = batch_in_chunks(amb, False, ashape[0], ashape[1])
        between, nd = amb.two_phase_handover(mmb, fp_tolerance=fp_tolerance, phase1_status=st1)
        st2, np2 = batch_in_chunks(mmb, is_max, ashape[0], ashape[1])
        for q, (k, a, t) in enumerate(members):
            st = int(st2[q]) if int(between[q]) == capi.MI_OK else int(between[q])
            try:
                _raise_for(st)
            except SolverError as e:
                results[k] = e
                continue
            adopt(t, mmb, q)
            t.n_pivots = (int(np1[q] + nd[q]), int(np2[q]))
            results[k] = t
    if errorp:
        for r in results:
            if isinstance(r, Exception):
                raise r
    return results


def _unit_basis(tableau):
    """The glue's `unit-basis-p`: every basis column is exactly the unit vector of its row."""
    M, b = tableau.matrix, tableau.basis_columns
    if len(set(b.tolist())) != len(b) or (len(b) and (b.min() < 0 or b.max() >= M.shape[1] - 1)):
        return False
    cols = M[:, b]
    unit = np.zeros_like(cols)
    unit[np.arange(len(b)), np.arange(len(b))] = 1.0
    return bool(np.array_equal(cols, unit) and not np.signbit(cols).any())
