"""The reference's parsed `problem` struct (src/problem.lisp:45-53) -- the INPUT type of the
solver boundary.  The DSL parser that produces it (src/problem.lisp:73-205) is upstream of
the hot path and out of scope: problems are constructed directly in parsed form."""
from dataclasses import dataclass, field
from typing import Any, List, Optional, Tuple


@dataclass
class Problem:
    type: str = "max"                                   # 'max | 'min
    vars: List[str] = field(default_factory=list)       # problem-vars (column order)
    objective_var: Optional[str] = None                 # problem-objective-var
    objective_func: List[Tuple[str, Any]] = field(default_factory=list)   # alist (var . coef)
    integer_vars: List[str] = field(default_factory=list)
    var_bounds: List[Tuple[str, Tuple[Any, Any]]] = field(default_factory=list)  # (var . (lb . ub))
    constraints: List[Tuple[str, List[Tuple[str, Any]], Any]] = field(default_factory=list)

    @classmethod
    def from_dict(cls, d):
        """From the JSON form used by tests/golden/reference_cases.json."""
        return cls(type=d["type"], vars=list(d["vars"]), objective_var=d.get("objective_var"),
                   objective_func=[(v, c) for v, c in d["objective"]],
                   integer_vars=list(d.get("integer_vars", [])),
                   var_bounds=[(b[0], (b[1], b[2])) for b in d.get("bounds", [])],
                   constraints=[(op, [(v, c) for v, c in e], rhs)
                                for op, e, rhs in d.get("constraints", [])])
