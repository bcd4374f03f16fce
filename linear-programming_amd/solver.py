"""Host-side mirror of the reference's `linear-programming/solver` package (src/solver.lisp):
the `*solver*` hook and the four solution-* generics.

In the Lisp deployment nothing in src/solver.lisp changes: the glue's
`mi355x-simplex-solver` is simply the value of `linear-programming:*solver*`
(`(let ((*solver* 'mi355x-simplex-solver)) (solve-problem problem))`).  On the native route it
returns a MI355X-SOLUTION, for which the glue adds one method to each of the four generics
(lisp/mi355x-simplex.lisp); on the build-tableau route it returns a `tableau`, served by the
reference's own methods at src/solver.lisp:61-80.  This module restates that two-class dispatch
for the Python-side tests.
"""
from . import simplex
from .native import NativeSolution

#: `*solver*` (src/solver.lisp:39-49): a function taking a problem and backend-specific keyword
#: arguments and returning a solution object.  Defaults to the MI355X backend here.
SOLVER = simplex.mi355x_simplex_solver


def solve_problem(problem, **kwargs):
    """solve-problem (src/solver.lisp:53-56): (apply *solver* problem args)."""
    return SOLVER(problem, **kwargs)


def solve_problems(problems, **kwargs):
    """The glue's `mi355x-solve-problems`: what [solve_problem(p) for p in problems] returns, with
    the independent LPs solved side by side (the hook itself takes one problem per call,
    src/solver.lisp:53-56)."""
    return simplex.mi355x_solve_problems(problems, **kwargs)


def solution_problem(solution):
    """solution-problem (src/solver.lisp:59-62)."""
    return solution.problem


def solution_objective_value(solution):
    """solution-objective-value (src/solver.lisp:64-67)."""
    if isinstance(solution, NativeSolution):          # (defmethod ... ((solution mi355x-solution)))
        return solution.objective_value()
    return simplex.tableau_objective_value(solution)


def solution_variable(solution, variable):
    """solution-variable (src/solver.lisp:69-72)."""
    if isinstance(solution, NativeSolution):
        return solution.variable(variable)
    return simplex.tableau_variable(solution, variable)


def solution_reduced_cost(solution, variable):
    """solution-reduced-cost (src/solver.lisp:74-80)."""
    if isinstance(solution, NativeSolution):
        return solution.reduced_cost(variable)
    return simplex.tableau_reduced_cost(solution, variable)


def with_solution_variables(var_list, solution):
    """with-solution-variables (src/solver.lisp:96-115) as a function: (values, reduced_cost)
    where `values` maps each requested variable (or, for a Problem, its objective variable and
    every variable) to its value in the solution and `reduced_cost(var)` is the locally bound
    `reduced-cost` macro."""
    from .problem import Problem
    if isinstance(var_list, Problem):
        names = [var_list.objective_var] + list(var_list.vars)
    else:
        names = list(var_list)
    values = {v: solution_variable(solution, v) for v in names}
    return values, (lambda var: solution_reduced_cost(solution, var))
