"""linear-programming_amd -- MI355X-native dense-simplex backend for the Common Lisp library
neil-lindquist/linear-programming, behind that library's `*solver*` hook.

The product is ``libmi355x_simplex.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/mi355x_simplex.h``).  This Python package is the host-side mirror of the reference's
operator interface for the hot path (``src/simplex.lisp`` / ``src/solver.lisp`` names, argument
meaning and error behaviour) over that C ABI via ctypes, used by the parity tests and the
benchmark; the production binding is the CFFI glue in ``lisp/``.

There is no CPU fallback: importing works anywhere, computing needs the built extension and a
GPU, and fails loudly otherwise.

(The directory name contains a hyphen; import it with
``importlib.import_module("linear-programming_amd")``.)
"""
from . import batch, capi, native, simplex, solver, synth       # noqa: F401
from .native import NativeProblem, NativeSolution               # noqa: F401
from .batch import TableauBatch, MultiDeviceBatch               # noqa: F401
from .conditions import (SolverError, UnboundedProblemError, InfeasibleProblemError,   # noqa: F401
                         UnsupportedConstraintError, ParsingError)
from .problem import Problem                                    # noqa: F401
from .simplex import (Tableau, build_tableau, pivot_row, n_pivot_row, solve_tableau,    # noqa: F401
                      n_solve_tableau, copy_tableau, tableau_objective_value,
                      tableau_variable, tableau_reduced_cost, find_entering_column,
                      find_pivoting_row, simplex_solver, mi355x_simplex_solver, mi355x_solve_problems,
                      with_tableau_variables)
from .solver import (solve_problem, solve_problems, solution_problem, solution_objective_value,         # noqa: F401
                     solution_variable, solution_reduced_cost, with_solution_variables)

__version__ = "0.1.0"
