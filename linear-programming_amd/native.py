"""ctypes view of the NATIVE host side of the hook (csrc/host_problem.cpp): problem ->
build-tableau -> n-solve-tableau on the GPU -> light solution object, all in C++ behind the C
ABI (mi355x_problem_*, mi355x_build_tableau, mi355x_simplex_solver, mi355x_solution_*).  This
is what a non-Lisp, non-Python caller links against; simplex.py keeps the reference-named Python
mirror used by most tests."""
import ctypes

import numpy as np

from . import capi
from .conditions import (InfeasibleProblemError, SolverError, UnboundedProblemError,
                         UnsupportedConstraintError)

_OPS = {"<=": 0, ">=": 1, "=": 2}


def _arr(xs, dt):
    a = np.ascontiguousarray(xs, dtype=dt)
    return a, a.ctypes.data_as(ctypes.c_void_p)


class NativeProblem:
    """A mi355x_problem built from a `Problem` (src/problem.lisp:45-53 in parsed form)."""

    def __init__(self, problem):
        L = capi.lib()
        self.problem = problem
        self.index = {v: i for i, v in enumerate(problem.vars)}
        h = ctypes.c_void_p()
        capi.check(L.mi355x_problem_create(ctypes.byref(h), int(problem.type == "max"),
                                           len(problem.vars)), "mi355x_problem_create")
        self._h = h
        v, vp = _arr([self.index[x] for x, _ in problem.objective_func], np.int64)
        c, cp = _arr([float(k) for _, k in problem.objective_func], np.float64)
        capi.check(L.mi355x_problem_set_objective(h, vp, cp, len(v)), "set_objective")
        for var, (lb, ub) in problem.var_bounds:
            capi.check(L.mi355x_problem_set_bounds(h, self.index[var], int(lb is not None),
                                                   float(lb or 0), int(ub is not None),
                                                   float(ub or 0)), "set_bounds")
        for var in problem.integer_vars:
            capi.check(L.mi355x_problem_set_integer(h, self.index[var]), "set_integer")
        for op, expr, rhs in problem.constraints:
            v, vp = _arr([self.index[x] for x, _ in expr], np.int64)
            c, cp = _arr([float(k) for _, k in expr], np.float64)
            capi.check(L.mi355x_problem_add_constraint(h, _OPS.get(op, -1), vp, cp, len(v),
                                                       float(rhs)), "add_constraint")

    def build_tableau(self):
        """[(matrix, basis)] for a single-phase problem, [(art...), (main...)] for two-phase."""
        L = capi.lib()
        two = ctypes.c_int(0)
        out = []
        rc = L.mi355x_build_tableau(self._h, 0, None, None, None, None, ctypes.byref(two))
        if rc == capi.MI_UNBOUNDED:
            raise UnboundedProblemError()
        capi.check(rc, "mi355x_build_tableau")
        for which in ([1, 0] if two.value else [0]):
            r, c = ctypes.c_int64(0), ctypes.c_int64(0)
            capi.check(L.mi355x_build_tableau(self._h, which, ctypes.byref(r), ctypes.byref(c), None,
                                              None, None), "mi355x_build_tableau")
            M = np.empty((r.value, c.value))
            b = np.empty(r.value - 1, dtype=np.int64)
            capi.check(L.mi355x_build_tableau(self._h, which, None, None,
                                              M.ctypes.data_as(ctypes.c_void_p),
                                              b.ctypes.data_as(ctypes.c_void_p) if b.size else None,
                                              None), "mi355x_build_tableau")
            out.append((M, b))
        return out

    def var_mapping(self, var):
        k, c, o = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_double(0)
        capi.check(capi.lib().mi355x_var_mapping(self._h, self.index[var], ctypes.byref(k),
                                                 ctypes.byref(c), ctypes.byref(o)), "var_mapping")
        kind = ("positive", "negative", "signed")[k.value]
        return (kind, c.value) if kind == "signed" else (kind, c.value, o.value)

    def solve_in_chunks(self, fp_tolerance=1024, device=0, max_pivots=0, chunk=None):
        """The Lisp glue's `solve-natively`, call for call: mi355x_simplex_solver_begin, ..._step in
        bounded chunks (`solve-in-chunks`; the phases of a two-phase problem included), ..._finish;
        the job is abandoned on every other way out."""
        from .simplex import _solve_in_chunks, _raise_for
        L = capi.lib()
        job, s, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(0)
        rc = L.mi355x_simplex_solver_begin(self._h, float(fp_tolerance), device, ctypes.byref(job))
        if rc == capi.MI_UNBOUNDED:                    # the no-constraint special case, simplex.lisp:170,174
            raise UnboundedProblemError()
        if rc == capi.MI_UNSUPPORTED:
            raise UnsupportedConstraintError(("integer",) + tuple(self.problem.integer_vars),
                                             "mi355x-simplex")
        capi.check(rc, "mi355x_simplex_solver_begin")
        consumed = False
        try:
            def call(cap):
                rc = capi.check(L.mi355x_simplex_solver_step(job, int(cap), ctypes.byref(n)),
                                "mi355x_simplex_solver_step")
                return rc, int(n.value)
            rows = len(self.problem.constraints) + 1
            rc, _ = _solve_in_chunks(call, rows, rows + len(self.problem.vars), int(max_pivots), chunk=chunk)
            _raise_for(rc)
            consumed = True
            capi.check(L.mi355x_simplex_solver_finish(job, ctypes.byref(s)), "mi355x_simplex_solver_finish")
            return NativeSolution(self, s)
        finally:
            if not consumed:
                L.mi355x_simplex_solver_abandon(job)

    def solve(self, fp_tolerance=1024, device=0):
        """mi355x_simplex_solver: returns a NativeSolution or raises the reference's errors."""
        s = ctypes.c_void_p()
        rc = capi.lib().mi355x_simplex_solver(self._h, float(fp_tolerance), device, ctypes.byref(s))
        if rc == capi.MI_UNBOUNDED:
            raise UnboundedProblemError()
        if rc == capi.MI_INFEASIBLE:
            raise InfeasibleProblemError()
        if rc == capi.MI_UNSUPPORTED:
            raise UnsupportedConstraintError(("integer",) + tuple(self.problem.integer_vars),
                                             "mi355x-simplex")
        if rc in (capi.MI_ART_NONZERO, capi.MI_ART_STUCK):
            raise SolverError("artificial variable could not be removed from the basis")
        capi.check(rc, "mi355x_simplex_solver")
        return NativeSolution(self, s)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            capi.lib().mi355x_problem_destroy(h)


class NativeSolution:
    """The glue's MI355X-SOLUTION: the library's light solution object (objective row, RHS column,
    basis, var-mapping) behind the four solution-* generics (src/solver.lisp:59-80)."""

    def __init__(self, nproblem, handle):
        self.nproblem, self._h = nproblem, handle
        self.problem = nproblem.problem                 # solution-problem

    def objective_value(self):
        x = ctypes.c_double(0)
        capi.check(capi.lib().mi355x_solution_objective_value(self._h, ctypes.byref(x)), "objective")
        return x.value

    def variable(self, var):
        if var == self.nproblem.problem.objective_var:
            return self.objective_value()
        if var not in self.nproblem.index:
            raise KeyError("%s is not a variable in the tableau" % (var,))
        x = ctypes.c_double(0)
        capi.check(capi.lib().mi355x_solution_variable(self._h, self.nproblem.index[var],
                                                       ctypes.byref(x)), "variable")
        return x.value

    def reduced_cost(self, var):
        if var not in self.nproblem.index:
            raise KeyError("%s is not a variable in the tableau" % (var,))
        x = ctypes.c_double(0)
        rc = capi.lib().mi355x_solution_reduced_cost(self._h, self.nproblem.index[var], ctypes.byref(x))
        if rc == capi.MI_BAD_ARG:
            raise ValueError("%s has no lower bound" % (var,))
        capi.check(rc, "reduced_cost")
        return x.value

    def pivots(self):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        capi.check(capi.lib().mi355x_solution_pivots(self._h, ctypes.byref(a), ctypes.byref(b)), "pivots")
        return a.value, b.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            capi.lib().mi355x_solution_destroy(h)


def solve_many(problems, fp_tolerance=1024, devices=1, device_ids=None, max_pivots=0, chunk=None):
    """The glue's native `mi355x-solve-problems`, call for call: every problem marshalled
    (mi355x_problem_*), ONE mi355x_simplex_solver_many_begin for the list (the library groups the
    members by tableau shape and sense into multi-device batches), ..._many_step in bounded chunks,
    ..._many_finish.  Returns a list parallel to `problems`: a NativeSolution or the exception the
    one-problem hook would raise for that member."""
    from .simplex import _raise_for, chunk_pivots
    L = capi.lib()
    nps = [NativeProblem(p) for p in problems]
    n = len(nps)
    arr = (ctypes.c_void_p * n)(*[q._h.value for q in nps])
    ids = None if device_ids is None else (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
    job = ctypes.c_void_p()
    capi.check(L.mi355x_simplex_solver_many_begin(arr, n, float(fp_tolerance), int(devices), ids, ctypes.byref(job)),
               "mi355x_simplex_solver_many_begin")
    status = (ctypes.c_int32 * n)()
    consumed = False
    try:
        rows = max(len(p.constraints) for p in problems) + 1
        cols = rows + max(len(p.vars) for p in problems)
        step = chunk or chunk_pivots(rows, cols)
        done = 0
        while True:
            cap = min(step, max_pivots - done) if max_pivots > 0 else step
            rc = capi.check(L.mi355x_simplex_solver_many_step(job, int(cap), status), "mi355x_simplex_solver_many_step")
            done += cap
            if rc != capi.MI_MAX_PIVOTS or (max_pivots > 0 and done >= max_pivots):
                break
        out = (ctypes.c_void_p * n)()
        consumed = True
        capi.check(L.mi355x_simplex_solver_many_finish(job, status, out), "mi355x_simplex_solver_many_finish")
    finally:
        if not consumed:
            L.mi355x_simplex_solver_many_abandon(job)
    results = []
    for k in range(n):
        st = int(status[k])
        if st == capi.MI_OPTIMAL:
            results.append(NativeSolution(nps[k], ctypes.c_void_p(out[k])))
            continue
        try:
            if st == capi.MI_UNSUPPORTED:
                raise UnsupportedConstraintError(("integer",) + tuple(problems[k].integer_vars), "mi355x-simplex")
            if st == capi.MI_RUNNING:
                st = capi.MI_MAX_PIVOTS
            _raise_for(st)
            raise SolverError("status %d" % st)
        except SolverError as e:
            results.append(e)
    return results


_READ_CASE = {"upcase": 0, "downcase": 1, "preserve": 2, "invert": 3}


def read_mps(text, problem_type=None, rhs_id=None, read_case="upcase", single_variable_rows="reference"):
    """read-mps (src/external-formats.lisp:78-348) through the native reader
    (csrc/mps_reader.cpp): fixed-width MPS text -> `Problem`.  problem_type: 'max' / 'min' /
    None (the file's OBJSENSE section decides).  single_variable_rows: "reference" (default: folded
    into bounds exactly as src/external-formats.lisp:312-323 does, quirks included) or "as-meant"."""
    import json
    from .conditions import ParsingError
    from .problem import Problem
    L = capi.lib()
    data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    h = ctypes.c_void_p()
    rc = L.mi355x_problem_read_mps_ex(data, len(data),
                                      {None: -1, "max": 1, "min": 0}[problem_type],
                                      None if rhs_id is None else rhs_id.encode("utf-8"),
                                      _READ_CASE[read_case],
                                      {"reference": 0, "as-meant": 1}[single_variable_rows], ctypes.byref(h))
    if rc in (capi.MI_BAD_ARG, capi.MI_UNSUPPORTED):
        raise ParsingError(L.mi355x_last_error().decode("utf-8", "replace"))
    capi.check(rc, "mi355x_problem_read_mps")
    # (MI_OK with a note: the reference's loop changed what a single-variable row means -- see the header)
    note = L.mi355x_last_error().decode("utf-8", "replace")
    if note.startswith("mps note:"):
        import warnings
        warnings.warn(note, stacklevel=2)
    try:
        n = L.mi355x_problem_to_json(h, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        L.mi355x_problem_to_json(h, buf, n + 1)
        d = json.loads(buf.value.decode("utf-8"))
        names = [L.mi355x_mps_var_name(i).decode("utf-8") for i in range(L.mi355x_mps_var_count())]
        objective_var = L.mi355x_mps_objective_name().decode("utf-8")
    finally:
        L.mi355x_problem_destroy(h)
    return Problem(type=d["type"], vars=names, objective_var=objective_var,
                   objective_func=[(names[v], c) for v, c in d["objective"]],
                   integer_vars=[names[v] for v in d["integer"]],
                   var_bounds=[(names[v], (lb, ub)) for v, lb, ub in d["bounds"]],
                   constraints=[(op, [(names[v], c) for v, c in e], rhs)
                                for op, e, rhs in d["constraints"]])
