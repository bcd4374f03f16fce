"""Generates tests/golden/reference_exports.json: the :export lists of the reference packages the Lisp
glue imports from (src/simplex.lisp, src/problem.lisp, src/solver.lisp, src/conditions.lisp of
neil-lindquist/linear-programming).  The fixture is DATA -- package name -> exported symbol names, i.e. the
reference's public API surface -- so that the glue's (:import-from ...) lists can be checked where the
reference tree is absent (the GPU box); tests/test_capi_symbols.py re-derives it from /root/reference where
that exists and fails if the fixture has gone stale.
    python tools/gen_reference_exports.py [/root/reference]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("simplex", "problem", "solver", "conditions")


def exports_of(path):
    """(package name, sorted exported symbols) of the (uiop:define-package / defpackage ...) form in `path`."""
    src = re.sub(r";[^\n]*", "", open(path).read())
    m = re.search(r"\((?:uiop:define-package|defpackage)\s+:?([\w/.-]+)", src)
    assert m, path
    pkg = m.group(1).lower()
    # the package form ends where its parentheses balance
    i, depth = m.start(), 0
    while True:
        depth += {"(": 1, ")": -1}.get(src[i], 0)
        i += 1
        if depth == 0:
            break
    form = src[m.start():i]
    syms = set()
    for ex in re.finditer(r"\(:export\b([^()]*)\)", form):
        syms |= {s.lower().lstrip("#:") for s in ex.group(1).split()}
    return pkg, sorted(syms)


def derive(ref):
    out = {}
    for f in FILES:
        pkg, syms = exports_of(os.path.join(ref, "src", f + ".lisp"))
        out[pkg] = {"file": "src/%s.lisp" % f, "exports": syms}
    return out


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    data = derive(ref)
    dst = os.path.join(ROOT, "tests", "golden", "reference_exports.json")
    json.dump(data, open(dst, "w"), indent=1, sort_keys=True)
    print(dst, {k: len(v["exports"]) for k, v in data.items()})
