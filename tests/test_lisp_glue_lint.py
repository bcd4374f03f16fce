"""The Lisp glue cannot be loaded here (no Common Lisp in the image).  tools/lisp_lint.py READS it with a real
s-expression reader and walks every form with the lexical environment a compiler would keep: operators resolve
(own definitions, the reference's exported functions with the reference's lambda lists, the C bindings, CFFI /
SBCL operators, standard Common Lisp), calls have an accepted argument count and declared keywords only,
variables are bound where they are used, setf places expand, conditions and classes take the initargs they are
given, foreign type keywords exist, return-from names a block in scope, exports are defined.

The second test is the guard's own test: fifteen-odd one-token slips of the kind an unexecuted file collects --
each must be found."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lisp_lint  # noqa: E402


def test_reference_signature_fixture_is_current():
    """tests/golden/reference_signatures.json (argument counts / keywords of the reference's functions, initargs
    of its conditions: interface DATA) is what the reference tree yields, wherever that tree is present."""
    fix = lisp_lint.load_signatures()
    assert set(fix) == {"linear-programming/simplex", "linear-programming/problem", "linear-programming/solver",
                        "linear-programming/conditions"}
    # what the glue leans on (src/simplex.lisp:142, src/conditions.lisp:69-73, src/solver.lisp:59-80)
    bt = fix["linear-programming/simplex"]["functions"]["build-tableau"]
    assert bt["min"] == 2 and bt["keys"] == [":fp-tolerance-factor"]
    assert fix["linear-programming/conditions"]["conditions"]["unsupported-constraint-error"]["initargs"] == [":constraint", ":solver-name"]
    assert fix["linear-programming/solver"]["functions"]["solution-variable"]["min"] == 2
    if os.path.isdir("/root/reference/src"):
        import json
        fresh = json.loads(json.dumps(lisp_lint.reference_signatures("/root/reference")))
        stored = json.load(open(lisp_lint.SIGNATURES))
        assert fresh == stored, "stale fixture: python tools/lisp_lint.py --signatures"


@pytest.mark.parametrize("features", [("sbcl", "unix"), ("unix",)], ids=["sbcl", "portable"])
def test_glue_reads_and_every_name_resolves(features):
    findings, linter = lisp_lint.lint(features=features)
    assert findings == [], "\n".join(findings)
    # (the walk saw the file: its definitions, the bindings of the C ABI, the reference symbols it uses)
    assert len(linter.forms) > 90 and len(linter.functions) > 80 and "with-foreign-fp-mode" in linter.macros
    assert {"mi355x-solution", "mi355x-error"} <= set(linter.classes)
    assert {"build-tableau", "tableau-matrix", "problem-constraints", "unsupported-constraint-error"} <= linter.used


SLIPS = [
    ("(tableau-matrix tableau))\n         (rows (array-dimension matrix 0))",
     "(tableau-matrx tableau))\n         (rows (array-dimension matrix 0))", "undefined operator tableau-matrx"),
    ("(%tab-create out rows cols pm pb device)", "(%tab-create out rows cols pm pb)", "called with 5 argument(s), takes 6"),
    ("(values flat basis rows cols)))\n\n(defun vectors->tableau", "(values flat basis rows colz)))\n\n(defun vectors->tableau",
     "unbound variable colz"),
    (":fp-tolerance-factor fp-tolerance))\n        (factor", ":fp-tolerance fp-tolerance))\n        (factor",
     "build-tableau does not take the keyword :fp-tolerance"),
    (":solver-name \"mi355x-simplex\"))\n  (when (and native", ":solver \"mi355x-simplex\"))\n  (when (and native",
     "no slot takes the initarg :solver"),
    ("(cffi:mem-ref out :pointer) flat basis))))", "(cffi:mem-ref out :ptr) flat basis))))", "unknown foreign type :ptr"),
    ("(return-from unit-basis-p nil)", "(return-from unit-basis nil)", "no such block in scope"),
    ("(setf (mi355x-solution-handle solution) (cffi:null-pointer))", "(setf (mi355x-solution-problem solution) (cffi:null-pointer))",
     "no setf expansion is known for mi355x-solution-problem"),
    ("(error 'mi355x-error :code status :message \"unknown status\")", "(error 'mi355x-error :code status :msg \"unknown status\")",
     "no slot takes the initarg :msg"),
    ("(solve-natively problem (coerce fp-tolerance 'double-float) device max-pivots)",
     "(solve-natively problem (coerce fp-tolerance 'double-float) device)", "called with 3 argument(s), takes 4"),
    (":native native)\n                     (error (c) c)))))", ":natve native)\n                     (error (c) c)))))",
     "does not take the keyword :natve"),
    ("for i from 0 do (setf (cffi:mem-aref ids :int i) d)))\n        (cffi:with-pointer-to-vector-data (pm flat)",
     "for i from 0 do (setf (cffi:mem-aref ids :int j) d)))\n        (cffi:with-pointer-to-vector-data (pm flat)", "unbound variable j"),
    ("#:free-solution", "#:free-solutions", "exports free-solutions, which the file does not define"),
    ("(defmethod solution-variable ((solution mi355x-solution) variable)", "(defmethod solution-variable ((solution mi355x-solution))",
     "takes 1 required argument(s), the generic function 2"),
    ("(handler-case (progn (signal-outcome status) nil)\n    (error (c) c))", "(handler-case (progn (signal-outcome status) nil)\n    (eror (c) c))",
     "unknown type eror"),
    ("(make-array (* rows cols) :element-type 'double-float))\n         (basis-src", "(make-array (* rows cols) :element-type 'double-flaot))\n         (basis-src",
     "unknown type double-flaot"),
    ("(destructuring-bind (lb . ub) (cdr entry)", "(destructuring-bind (lo . ub) (cdr entry)", "unbound variable lb"),
    ("(lambda (cap) (%solver-step job cap n-pivots))", "(lambda (cap) (%solver-step jop cap n-pivots))", "unbound variable jop"),
    ("(problem-objective-func problem) var-index\n", "(problem-objective-function problem) var-index\n", "undefined operator problem-objective-function"),
    ("(if (listp devices) (length devices) devices))", "(if (listp devices) (length devices) devices)", "unbalanced"),
]


def test_the_lint_finds_one_token_slips():
    src = open(lisp_lint.GLUE).read()
    for old, new, expect in SLIPS:
        assert src.count(old) >= 1, "the slip no longer applies to the glue: %r" % old
        try:
            findings, _ = lisp_lint.lint(text=src.replace(old, new, 1))
        except lisp_lint.ReadError as e:
            findings = [str(e)]
        assert any(expect in f for f in findings), "%r: expected a finding with %r, got %s" % (new, expect, findings[:3])


def test_reader_handles_the_syntax_the_glue_and_the_reference_use():
    R = lisp_lint.Reader
    forms = R("(a . b) #'f `(x ,y ,@z) #\\( \"s\\\"t\" #| c #| n |# |# ; c\n 'q #+sbcl 1 #-sbcl 2 #+(or x sbcl) 3 1d0 -2.5 7/2 #2A() #:u :k").read_all()
    assert forms[0] == ["a", ".", "b"] and forms[1] == ["function", "f"]
    assert forms[2] == ["quasiquote", ["x", ["unquote", "y"], ["unquote-splicing", "z"]]]
    assert isinstance(forms[3], lisp_lint.Char) and forms[4] == 's"t' and forms[5] == ["quote", "q"]
    assert forms[6:] == [1, 3, 1.0, -2.5, 3.5, ["quote", []], "#:u", ":k"]
    with pytest.raises(lisp_lint.ReadError):
        R("(a (b)").read_all()
    with pytest.raises(lisp_lint.ReadError):
        R("(a))").read_all()
