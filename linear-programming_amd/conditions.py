"""The reference's error taxonomy for the solver boundary (src/conditions.lisp:15-77).

C status codes of the C ABI map onto these exactly as the Lisp glue maps them onto the
conditions of the same names."""


class ParsingError(Exception):
    """parsing-error (src/conditions.lisp:15)."""


class SolverError(Exception):
    """solver-error (src/conditions.lisp:43): base class for errors of the solving algorithm."""


class UnboundedProblemError(SolverError):
    """unbounded-problem-error (src/conditions.lisp:47)."""

    def __str__(self):
        return "Problem is unbounded"


class InfeasibleProblemError(SolverError):
    """infeasible-problem-error (src/conditions.lisp:55)."""

    def __str__(self):
        return "Problem has no feasible region"


class UnsupportedConstraintError(SolverError):
    """unsupported-constraint-error (src/conditions.lisp:69-77), initargs :constraint :solver-name."""

    def __init__(self, constraint, solver_name):
        super().__init__(constraint, solver_name)
        self.constraint = constraint
        self.solver_name = solver_name

    def __str__(self):
        return "%r cannot be handled by the %s solver" % (self.constraint, self.solver_name)
