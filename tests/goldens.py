"""Loader for tests/golden/reference_cases.json (vectors transcribed from the
reference's own tests) -- shared by the oracle tests and the GPU parity tests."""
import json
import os
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EPS = 1.1102230246251568e-16


def frac(x, float32=False):
    if isinstance(x, str):
        return Fraction(x)
    if isinstance(x, float):
        # float literals in the fixture are Lisp single-float literals when the
        # case says so; otherwise exact small decimals such as 1.0 / 2.0
        return Fraction(float(np.float32(x))) if float32 else Fraction(x)
    return Fraction(x)


def fmat(rows):
    return [[frac(v) for v in row] for row in rows]


def load():
    with open(os.path.join(HERE, "golden", "reference_cases.json")) as f:
        return json.load(f)


def problem(case):
    """The case's problem with every number turned into an exact Fraction."""
    f32 = bool(case.get("float32_literals"))
    p = case["problem"]
    return {
        "type": p["type"], "vars": list(p["vars"]), "objective_var": p.get("objective_var"),
        "objective": [[v, frac(c, f32)] for v, c in p["objective"]],
        "bounds": [[b[0], None if b[1] is None else frac(b[1], f32),
                    None if b[2] is None else frac(b[2], f32)] for b in p["bounds"]],
        "constraints": [[op, [[v, frac(c, f32)] for v, c in e], frac(r, f32)]
                        for op, e, r in p["constraints"]],
    }


def to_f64(tab):
    """rational_ref.Tableau -> (float64 matrix, int64 basis)."""
    M = np.array([[float(v) for v in row] for row in tab.matrix], dtype=np.float64)
    return np.ascontiguousarray(M), np.array(tab.basis, dtype=np.int64)


def problem_dict(case):
    """The case's problem in the JSON dict layout but with exact Fractions as numbers."""
    p = problem(case)
    return {"type": p["type"], "vars": p["vars"], "objective_var": p["objective_var"],
            "objective": p["objective"], "bounds": p["bounds"], "constraints": p["constraints"]}
