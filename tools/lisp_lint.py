"""A reader and a scope-aware static check for the Common Lisp glue (linear-programming_amd/lisp/mi355x-simplex.lisp).

No Common Lisp implementation exists in the build image, so the glue has never been read by a Lisp reader, let
alone compiled.  This is the closest substitute that runs here: the file is READ into forms (a real s-expression
reader: strings, characters, block comments, quote / backquote / unquote, #' #: #+ #-, dotted pairs), and every
form is WALKED with the lexical environment a compiler would keep:

  * every operator resolves: a function / macro the glue defines, a local flet / labels function, a symbol the
    glue imports from the reference (checked against the reference's lambda lists, tests/golden/
    reference_signatures.json), a binding of the C ABI (cffi:defcfun), a package-qualified CFFI / SBCL operator
    from a short list, or a standard Common Lisp operator;
  * every call has an argument count its lambda list accepts, and only keywords it declares;
  * every variable read or assigned is bound: a parameter, a let / loop / dotimes / multiple-value-bind /
    destructuring-bind / handler-case / with-foreign-* variable in scope, a global the glue defines, or a constant;
  * every (setf place) is a place a compiler knows how to expand;
  * every condition / class named in error, make-condition, make-instance, handler-case, typep exists and takes
    the initargs it is given;
  * every foreign type keyword is one CFFI defines; every return-from names a block in scope;
  * every symbol the package exports is defined.

What it cannot see: types, run-time values, macro-expansion subtleties of LOOP beyond the clauses handled here.
    python tools/lisp_lint.py [file.lisp]          # prints the findings, exit status 1 if any
    python tools/lisp_lint.py --signatures [/root/reference]   # regenerates tests/golden/reference_signatures.json
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "linear-programming_amd", "lisp", "mi355x-simplex.lisp")
SIGNATURES = os.path.join(ROOT, "tests", "golden", "reference_signatures.json")
FEATURES = {"sbcl", "unix", "linux", "x86-64", "64-bit", "common-lisp", "ansi-cl"}


# ------------------------------------------------------------------------------------------- reader
class Sym(str):
    """A symbol, lower-cased, with its package prefix as written (cffi:mem-ref, :keyword, #:uninterned)."""
    line = 0


class Str(str):
    pass


class Char(str):
    pass


class Form(list):
    line = 0


class ReadError(Exception):
    pass


_NUM = re.compile(r"^[+-]?(\d+\.?\d*([eEdDfFsSlL][+-]?\d+)?|\.\d+([eEdDfFsSlL][+-]?\d+)?|\d+/\d+)$")
_TERM = set(" \t\n\r\f()'`,;\"")


class Reader:
    def __init__(self, text):
        self.s, self.i, self.line = text, 0, 1

    def error(self, msg):
        raise ReadError("line %d: %s" % (self.line, msg))

    def peek(self):
        return self.s[self.i] if self.i < len(self.s) else ""

    def next(self):
        c = self.s[self.i]
        self.i += 1
        if c == "\n":
            self.line += 1
        return c

    def skip_ws(self):
        while self.i < len(self.s):
            c = self.peek()
            if c in " \t\n\r\f":
                self.next()
            elif c == ";":
                while self.i < len(self.s) and self.peek() != "\n":
                    self.next()
            elif c == "#" and self.s[self.i + 1:self.i + 2] == "|":
                start, depth = self.line, 0
                while True:
                    if self.i >= len(self.s):
                        raise ReadError("line %d: unterminated #| comment" % start)
                    two = self.s[self.i:self.i + 2]
                    if two == "#|":
                        depth += 1
                        self.next(), self.next()
                    elif two == "|#":
                        depth -= 1
                        self.next(), self.next()
                        if depth == 0:
                            break
                    else:
                        self.next()
            else:
                return

    def read_all(self):
        out = []
        while True:
            self.skip_ws()
            if self.i >= len(self.s):
                return out
            f = self.read()
            if f is not _SKIP:
                out.append(f)

    def token(self):
        start = self.i
        while self.i < len(self.s) and self.peek() not in _TERM:
            if self.peek() == "\\":
                self.next()
            if self.peek() == "|":
                self.next()
                while self.peek() != "|":
                    if not self.peek():
                        self.error("unterminated |symbol|")
                    self.next()
            self.next()
        return self.s[start:self.i]

    def wrap(self, name, line):
        inner = self.read_form()
        f = Form([mksym(name, line), inner])
        f.line = line
        return f

    def read_form(self):
        """the next form that is not skipped by #+ / #-"""
        while True:
            self.skip_ws()
            if self.i >= len(self.s):
                self.error("end of file inside a form")
            f = self.read()
            if f is not _SKIP:
                return f

    def read(self):
        self.skip_ws()
        line = self.line
        c = self.next()
        if c == "(":
            out = Form()
            out.line = line
            while True:
                self.skip_ws()
                if self.i >= len(self.s):
                    raise ReadError("line %d: unbalanced '('" % line)
                if self.peek() == ")":
                    self.next()
                    return out
                f = self.read()
                if f is not _SKIP:
                    out.append(f)
        if c == ")":
            self.error("unbalanced ')'")
        if c == "'":
            return self.wrap("quote", line)
        if c == "`":
            return self.wrap("quasiquote", line)
        if c == ",":
            if self.peek() == "@":
                self.next()
                return self.wrap("unquote-splicing", line)
            return self.wrap("unquote", line)
        if c == '"':
            buf = []
            while True:
                if self.i >= len(self.s):
                    raise ReadError("line %d: unterminated string" % line)
                d = self.next()
                if d == "\\":
                    buf.append(self.next())
                elif d == '"':
                    return Str("".join(buf))
                else:
                    buf.append(d)
        if c == "#":
            d = self.next()
            if d == "'":
                return self.wrap("function", line)
            if d == "\\":
                ch = self.next()
                rest = self.token() if self.peek() not in _TERM else ""
                return Char(ch + rest)
            if d == ":":
                return mksym("#:" + self.token().lower(), line)
            if d in "+-":
                feature = self.read_form()
                form = self.read_form()
                return form if feature_holds(feature) == (d == "+") else _SKIP
            if d == "(":
                self.i -= 1
                v = self.read()
                f = Form([mksym("vector", line)] + list(v))
                f.line = line
                return Form([mksym("quote", line), f])
            if d == ".":
                return self.wrap("read-eval", line)
            if d.isdigit():                                      # #2A(...) and the like: literal data
                while self.peek().isdigit():
                    self.next()
                self.next()
                lit = Form([mksym("quote", line), self.read_form()])
                lit.line = line
                return lit
            self.error("reader macro #%s is not handled" % d)
        self.i -= 1
        tok = self.token()
        if not tok:
            self.error("unexpected character %r" % c)
        if _NUM.match(tok):
            t = re.sub(r"[dDfFsSlL]", "e", tok)
            if "/" in t:
                a, b = t.split("/")
                return int(a) / int(b) if int(a) % int(b) else int(a) // int(b)
            try:
                return int(t)
            except ValueError:
                return float(t)
        return mksym(tok.lower(), line)


_SKIP = object()


def mksym(name, line):
    s = Sym(name)
    s.line = line
    return s


def feature_holds(f):
    if isinstance(f, Sym):
        return f.lstrip(":") in FEATURES
    if isinstance(f, list) and f:
        op = f[0].lstrip(":")
        if op == "or":
            return any(feature_holds(x) for x in f[1:])
        if op == "and":
            return all(feature_holds(x) for x in f[1:])
        if op == "not":
            return not feature_holds(f[1])
    return False


def read_file(path):
    return Reader(open(path).read()).read_all()


# ------------------------------------------------------------------------------------------- lambda lists
LL_KEYWORDS = {"&optional", "&rest", "&key", "&allow-other-keys", "&aux", "&body", "&whole", "&environment"}


class Sig:
    """What a call must satisfy: min / max positional arguments (max None = any), keywords (None = any)."""

    def __init__(self, lo=0, hi=None, keys=None, allow_other=False, what="function"):
        self.lo, self.hi, self.keys, self.allow_other, self.what = lo, hi, keys, allow_other, what

    def to_json(self):
        return {"min": self.lo, "max": self.hi, "keys": sorted(self.keys) if self.keys is not None else None,
                "allow_other_keys": self.allow_other, "what": self.what}

    @staticmethod
    def from_json(d):
        return Sig(d["min"], d["max"], set(d["keys"]) if d["keys"] is not None else None, d["allow_other_keys"], d["what"])


def parse_lambda_list(ll, specialized=False):
    """-> (Sig, [(variable, default form or None)] in binding order).  specialized: a defmethod's list, whose
    required parameters may be (var class)."""
    lo = hi = 0
    keys, allow, mode, rest = None, False, "req", False
    binds = []
    for item in ll:
        if isinstance(item, Sym) and item in LL_KEYWORDS:
            mode = {"&optional": "opt", "&rest": "rest", "&body": "rest", "&key": "key", "&aux": "aux",
                    "&allow-other-keys": "allow", "&whole": "whole", "&environment": "whole"}[item]
            if mode == "key":
                keys = set() if keys is None else keys
            if mode == "allow":
                allow = True
            continue
        if mode == "req":
            lo += 1
            hi += 1
            if isinstance(item, list):
                if specialized:
                    binds.append((item[0], None))
                else:                                            # (destructuring: macros, destructuring-bind)
                    for v in flatten_vars(item):
                        binds.append((v, None))
            else:
                binds.append((item, None))
        elif mode == "opt":
            hi += 1
            if isinstance(item, list):
                binds.append((item[0], item[1] if len(item) > 1 else None))
                if len(item) > 2:
                    binds.append((item[2], None))
            else:
                binds.append((item, None))
        elif mode in ("rest", "whole"):
            rest = rest or mode == "rest"
            binds.append((item, None))
            if mode == "whole":
                mode = "req"
        elif mode == "key":
            var, default, svar = item, None, None
            if isinstance(item, list):
                var, default = item[0], (item[1] if len(item) > 1 else None)
                svar = item[2] if len(item) > 2 else None
            kw = ":" + var
            if isinstance(var, list):                            # ((:keyword var) default)
                kw, var = var[0], var[1]
            keys.add(str(kw))
            binds.append((var, default))
            if svar is not None:
                binds.append((svar, None))
        elif mode == "aux":
            if isinstance(item, list):
                binds.append((item[0], item[1] if len(item) > 1 else None))
            else:
                binds.append((item, None))
    sig = Sig(lo, None if (rest or keys is not None) else hi, keys, allow)
    if keys is not None and not rest:
        sig.hi_positional = hi
    sig.positional = hi
    sig.rest = rest
    return sig, binds


def flatten_vars(tree):
    """variables of a destructuring pattern (a tree of symbols, possibly dotted; NIL ignores a position)"""
    out = []
    if isinstance(tree, Sym):
        if tree not in LL_KEYWORDS and tree not in ("nil", "."):
            out.append(tree)
    elif isinstance(tree, list):
        for x in tree:
            out += flatten_vars(x)
    return out


# ------------------------------------------------------------------------------------------- what Common Lisp defines
def _sig(lo, hi=None, keys=None):
    return Sig(lo, hi, set(keys) if keys is not None else None, False, "cl")


_SEQ_KEYS = [":key", ":test", ":test-not", ":start", ":end", ":from-end", ":count", ":initial-value"]
CL_FUNCTIONS = {
    "+": _sig(0), "-": _sig(1), "*": _sig(0), "/": _sig(1), "1+": _sig(1, 1), "1-": _sig(1, 1),
    "=": _sig(1), "/=": _sig(1), "<": _sig(1), ">": _sig(1), "<=": _sig(1), ">=": _sig(1),
    "max": _sig(1), "min": _sig(1), "abs": _sig(1, 1), "floor": _sig(1, 2), "ceiling": _sig(1, 2), "truncate": _sig(1, 2),
    "round": _sig(1, 2), "mod": _sig(2, 2), "rem": _sig(2, 2), "expt": _sig(2, 2), "sqrt": _sig(1, 1), "float-sign": _sig(1, 2),
    "zerop": _sig(1, 1), "plusp": _sig(1, 1), "minusp": _sig(1, 1), "evenp": _sig(1, 1), "oddp": _sig(1, 1),
    "integerp": _sig(1, 1), "floatp": _sig(1, 1), "numberp": _sig(1, 1), "rationalp": _sig(1, 1), "realp": _sig(1, 1),
    "symbolp": _sig(1, 1), "stringp": _sig(1, 1), "listp": _sig(1, 1), "consp": _sig(1, 1), "null": _sig(1, 1), "atom": _sig(1, 1),
    "arrayp": _sig(1, 1), "vectorp": _sig(1, 1), "functionp": _sig(1, 1), "keywordp": _sig(1, 1), "hash-table-p": _sig(1, 1),
    "not": _sig(1, 1), "eq": _sig(2, 2), "eql": _sig(2, 2), "equal": _sig(2, 2), "equalp": _sig(2, 2), "identity": _sig(1, 1),
    "car": _sig(1, 1), "cdr": _sig(1, 1), "caar": _sig(1, 1), "cadr": _sig(1, 1), "cdar": _sig(1, 1), "cddr": _sig(1, 1),
    "caddr": _sig(1, 1), "cdddr": _sig(1, 1), "first": _sig(1, 1), "second": _sig(1, 1), "third": _sig(1, 1), "fourth": _sig(1, 1),
    "rest": _sig(1, 1), "last": _sig(1, 2), "nth": _sig(2, 2), "nthcdr": _sig(2, 2), "cons": _sig(2, 2), "list": _sig(0), "list*": _sig(1),
    "append": _sig(0), "nconc": _sig(0), "reverse": _sig(1, 1), "nreverse": _sig(1, 1), "length": _sig(1, 1), "elt": _sig(2, 2),
    "copy-list": _sig(1, 1), "copy-seq": _sig(1, 1), "subseq": _sig(2, 3), "member": _sig(2, None, [":key", ":test", ":test-not"]),
    "assoc": _sig(2, None, [":key", ":test", ":test-not"]), "mapcar": _sig(2), "mapc": _sig(2), "mapcan": _sig(2), "map": _sig(3),
    "maphash": _sig(2, 2), "reduce": _sig(2, None, _SEQ_KEYS), "find": _sig(2, None, _SEQ_KEYS), "find-if": _sig(2, None, _SEQ_KEYS),
    "position": _sig(2, None, _SEQ_KEYS), "position-if": _sig(2, None, _SEQ_KEYS), "count": _sig(2, None, _SEQ_KEYS),
    "count-if": _sig(2, None, _SEQ_KEYS), "remove": _sig(2, None, _SEQ_KEYS), "remove-if": _sig(2, None, _SEQ_KEYS),
    "remove-if-not": _sig(2, None, _SEQ_KEYS), "sort": _sig(2, None, [":key"]), "stable-sort": _sig(2, None, [":key"]),
    "every": _sig(2), "some": _sig(2), "notany": _sig(2), "notevery": _sig(2), "remove-duplicates": _sig(1, None, _SEQ_KEYS),
    "funcall": _sig(1), "apply": _sig(2), "values": _sig(0), "values-list": _sig(1, 1),
    "aref": _sig(1), "svref": _sig(2, 2), "row-major-aref": _sig(2, 2), "array-dimension": _sig(2, 2), "array-dimensions": _sig(1, 1),
    "array-total-size": _sig(1, 1), "array-rank": _sig(1, 1), "array-element-type": _sig(1, 1),
    "make-array": _sig(1, None, [":element-type", ":initial-element", ":initial-contents", ":adjustable", ":fill-pointer",
                                 ":displaced-to", ":displaced-index-offset"]),
    "make-list": _sig(1, None, [":initial-element"]), "vector": _sig(0), "make-string": _sig(1, None, [":initial-element", ":element-type"]),
    "make-hash-table": _sig(0, None, [":test", ":size", ":rehash-size", ":rehash-threshold"]), "gethash": _sig(2, 3), "remhash": _sig(2, 2),
    "hash-table-count": _sig(1, 1), "clrhash": _sig(1, 1),
    "coerce": _sig(2, 2), "float": _sig(1, 2), "typep": _sig(2, 3), "type-of": _sig(1, 1), "subtypep": _sig(2, 3),
    "error": _sig(1), "warn": _sig(1), "signal": _sig(1), "cerror": _sig(2), "make-condition": _sig(1),
    "format": _sig(2), "princ": _sig(1, 2), "prin1": _sig(1, 2), "print": _sig(1, 2), "terpri": _sig(0, 1), "write-string": _sig(1),
    "make-instance": _sig(1), "slot-value": _sig(2, 2), "slot-boundp": _sig(2, 2), "class-of": _sig(1, 1),
    "symbol-name": _sig(1, 1), "symbol-value": _sig(1, 1), "symbol-function": _sig(1, 1), "intern": _sig(1, 2), "find-symbol": _sig(1, 2),
    "string=": _sig(2), "string-equal": _sig(2), "string": _sig(1, 1), "string-upcase": _sig(1), "string-downcase": _sig(1),
    "concatenate": _sig(1), "ash": _sig(2, 2), "logand": _sig(0), "logior": _sig(0), "logxor": _sig(0),
    "constantly": _sig(1, 1), "complement": _sig(1, 1), "get-internal-real-time": _sig(0, 0), "sleep": _sig(1, 1),
    "asdf:load-system": _sig(1),
}
# macros and special operators that evaluate ALL their arguments as ordinary forms (an implicit progn or the like)
CL_PROGN_LIKE = {
    "progn": _sig(0), "prog1": _sig(1), "prog2": _sig(2), "when": _sig(1), "unless": _sig(1), "if": _sig(2, 3),
    "and": _sig(0), "or": _sig(0), "unwind-protect": _sig(1), "multiple-value-list": _sig(1, 1),
    "multiple-value-prog1": _sig(1), "locally": _sig(0), "assert": _sig(1), "time": _sig(1, 1), "nth-value": _sig(2, 2),
    "multiple-value-call": _sig(1),
}
CL_CONDITIONS = {"condition", "error", "warning", "simple-error", "simple-condition", "simple-warning", "type-error",
                 "arithmetic-error", "division-by-zero", "floating-point-overflow", "floating-point-invalid-operation",
                 "serious-condition", "storage-condition", "program-error", "control-error", "cell-error",
                 "unbound-variable", "undefined-function", "style-warning"}
CL_TYPES = CL_CONDITIONS | {
    "t", "nil", "null", "list", "cons", "symbol", "keyword", "number", "real", "rational", "integer", "fixnum", "bignum", "ratio",
    "float", "single-float", "double-float", "short-float", "long-float", "complex", "character", "string", "simple-string",
    "vector", "simple-vector", "array", "simple-array", "bit", "bit-vector", "hash-table", "function", "sequence", "boolean",
    "signed-byte", "unsigned-byte", "standard-object", "package", "pathname", "stream"}
CL_CONSTANTS = {"t", "nil", "pi", "most-positive-fixnum", "most-negative-fixnum", "most-positive-double-float",
                "double-float-epsilon", "double-float-negative-epsilon", "least-positive-double-float",
                "*standard-output*", "*error-output*", "*package*", "*features*", "*print-pretty*"}
CL_SETF_PLACES = {"aref", "svref", "row-major-aref", "gethash", "car", "cdr", "first", "second", "third", "rest", "nth", "elt",
                  "slot-value", "symbol-value", "symbol-function", "cadr", "cddr", "caar", "cdar", "values", "the", "subseq", "getf",
                  "cffi:mem-ref", "cffi:mem-aref", "cffi:foreign-slot-value"}
# operators of other packages the glue may use, each with its arity (everything else package-qualified is a finding)
FOREIGN_FUNCTIONS = {
    "cffi:null-pointer": _sig(0, 0), "cffi:null-pointer-p": _sig(1, 1), "cffi:pointer-eq": _sig(2, 2),
    "cffi:mem-ref": _sig(2, 3), "cffi:mem-aref": _sig(2, 3), "cffi:foreign-alloc": _sig(1, None, [":initial-element", ":initial-contents", ":count", ":null-terminated-p"]),
    "cffi:foreign-free": _sig(1, 1), "cffi:foreign-string-to-lisp": _sig(1), "cffi:inc-pointer": _sig(2, 2), "cffi:pointerp": _sig(1, 1),
    "cffi:use-foreign-library": _sig(1, 1), "cffi:load-foreign-library": _sig(1),
    "sb-ext:finalize": _sig(2, None, [":dont-save"]), "sb-ext:cancel-finalization": _sig(1, 1), "sb-ext:gc": _sig(0, None, [":full"]),
}
CFFI_TYPES = {":int", ":int8", ":int16", ":int32", ":int64", ":uint8", ":uint16", ":uint32", ":uint64", ":char", ":unsigned-char",
              ":short", ":unsigned-short", ":unsigned-int", ":long", ":unsigned-long", ":long-long", ":unsigned-long-long",
              ":float", ":double", ":pointer", ":string", ":void", ":boolean", ":size"}
LOOP_KEYWORDS = {"for", "as", "with", "and", "do", "doing", "collect", "collecting", "append", "appending", "nconc", "nconcing",
                 "sum", "summing", "count", "counting", "maximize", "maximizing", "minimize", "minimizing", "into", "when", "if",
                 "unless", "else", "end", "while", "until", "always", "never", "thereis", "finally", "initially", "return", "repeat",
                 "named", "in", "on", "across", "from", "upfrom", "downfrom", "to", "upto", "downto", "below", "above", "by", "=",
                 "then", "being", "the", "each", "of", "of-type", "using", "hash-key", "hash-keys", "hash-value", "hash-values", "it"}


# ------------------------------------------------------------------------------------------- the walk
class Env:
    def __init__(self, parent=None):
        self.parent, self.vars, self.funcs, self.blocks = parent, set(), {}, set()

    def child(self):
        return Env(self)

    def has_var(self, v):
        e = self
        while e:
            if v in e.vars:
                return True
            e = e.parent
        return False

    def func(self, f):
        e = self
        while e:
            if f in e.funcs:
                return e.funcs[f]
            e = e.parent
        return None

    def has_block(self, b):
        e = self
        while e:
            if b in e.blocks:
                return True
            e = e.parent
        return False


def is_keyword(x):
    return isinstance(x, Sym) and x.startswith(":")


def quoted_symbol(x):
    if isinstance(x, list) and len(x) == 2 and x[0] == "quote" and isinstance(x[1], Sym):
        return x[1]
    return None


class Linter:
    def __init__(self, forms, reference=None):
        self.forms = forms
        self.findings = []
        self.functions = {}      # name -> Sig (global functions, generics, macros the glue defines, C bindings)
        self.macros = {}         # name -> (Sig, lambda list)
        self.globals = set()
        self.classes = {}        # name -> set of initargs (classes and conditions the glue defines)
        self.setf_places = set()
        self.imports = {}        # symbol -> package
        self.exports = []
        self.used = set()
        self.reference = reference or {}      # package -> {symbol -> signature json}
        self.ref_sigs = {}
        self.ref_conditions = {}
        self.package = None

    def report(self, where, msg):
        self.findings.append("line %d: %s" % (getattr(where, "line", 0), msg))

    # ---- pass 1: what the file defines
    def collect(self):
        for f in self.forms:
            if not isinstance(f, list) or not f or not isinstance(f[0], Sym):
                continue
            op = f[0]
            if op == "defpackage":
                self.package = f[1].lstrip(":#")
                for clause in f[2:]:
                    if clause[0] == ":import-from":
                        pkg = clause[1].lstrip(":#")
                        for s in clause[2:]:
                            self.imports[s.lstrip("#:")] = pkg
                    elif clause[0] == ":export":
                        self.exports += [(s.lstrip("#:"), s) for s in clause[1:]]
            elif op in ("defconstant", "defvar", "defparameter"):
                self.globals.add(str(f[1]))
            elif op in ("defun", "defgeneric"):
                self.define(f[1], parse_lambda_list(f[2])[0], f)
            elif op == "defmethod":
                ll = next(x for x in f[2:] if isinstance(x, list))
                sig = parse_lambda_list(ll, specialized=True)[0]
                if str(f[1]) not in self.functions and str(f[1]) not in self.imports:
                    self.define(f[1], sig, f)
            elif op == "defmacro":
                sig, _ = parse_lambda_list(f[2])
                sig.what = "macro"
                self.macros[str(f[1])] = (sig, f[2])
            elif op == "cffi:defcfun":
                name = f[1][1] if isinstance(f[1], list) else f[1]
                n = len(f) - 3
                self.define(name, Sig(n, n, None, False, "C binding of %s" % (f[1][0] if isinstance(f[1], list) else f[1])), f)
            elif op in ("define-condition", "defclass"):
                initargs = set()
                for slot in f[3]:
                    opts = slot[1:] if isinstance(slot, list) else []
                    for k, v in zip(opts[0::2], opts[1::2]):
                        if k == ":initarg":
                            initargs.add(str(v))
                        elif k in (":reader", ":accessor"):
                            self.define(v, Sig(1, 1, None, False, "slot reader"), f)
                            if k == ":accessor":
                                self.setf_places.add(str(v))
                        elif k == ":writer":
                            self.define(v, Sig(2, 2, None, False, "slot writer"), f)
                self.classes[str(f[1])] = (initargs, [str(p) for p in f[2]], op)
        for pkg, table in self.reference.items():
            for name, d in table.get("functions", {}).items():
                self.ref_sigs[(pkg, name)] = Sig.from_json(d)
            for name, d in table.get("conditions", {}).items():
                self.ref_conditions[(pkg, name)] = d

    def define(self, name, sig, where):
        if str(name) in self.functions:
            self.report(where, "%s is defined twice" % name)
        self.functions[str(name)] = sig

    # ---- resolution
    def function_sig(self, name, env, where):
        """the Sig of operator `name`, or None if it is unknown (reported)"""
        local = env.func(name)
        if local is not None:
            return local
        if name in self.functions:
            return self.functions[name]
        if name in self.imports:
            self.used.add(str(name))
            sig = self.ref_sigs.get((self.imports[name], str(name)))
            if sig is None and self.reference:
                self.report(where, "%s is imported from %s, whose source defines no function of that name" % (name, self.imports[name]))
                return Sig(0, None)
            return sig or Sig(0, None)
        if name in CL_FUNCTIONS:
            return CL_FUNCTIONS[name]
        if name in FOREIGN_FUNCTIONS:
            return FOREIGN_FUNCTIONS[name]
        if ":" in name and not name.startswith(":"):
            pkg, _, sym = name.partition(":")
            sym = sym.lstrip(":")
            if pkg in self.reference or pkg == "linear-programming":
                for (p, s), sig in self.ref_sigs.items():
                    if s == sym and (p == pkg or pkg == "linear-programming"):
                        return sig
        self.report(where, "undefined operator %s" % name)
        return None

    def check_call(self, name, sig, args, where):
        if sig is None:
            return
        n = len(args)
        positional = getattr(sig, "positional", sig.hi if sig.hi is not None else sig.lo)
        if sig.keys is not None:
            # positional arguments up to the declared ones, then keyword / value pairs
            npos = positional if n >= positional else n
            rest = args[npos:]
            if n < sig.lo:
                self.report(where, "%s (%s) called with %d argument(s), needs at least %d" % (name, sig.what, n, sig.lo))
                return
            # optional positionals of CL functions (gethash etc.) never combine with keys here
            if not getattr(sig, "rest", False) and sig.what != "cl" and len(rest) % 2:
                self.report(where, "%s: odd number of keyword arguments" % name)
            if len(rest) % 2 == 0 or sig.what != "cl":
                for k in rest[0::2]:
                    if is_keyword(k):
                        # (&allow-other-keys lets a misspelt keyword through silently: a call from INSIDE the glue
                        # to one of its own functions names declared keywords only)
                        if str(k) not in sig.keys and (not sig.allow_other or sig.what == "function") and str(k) != ":allow-other-keys":
                            self.report(where, "%s does not take the keyword %s (it takes %s)" % (name, k, " ".join(sorted(sig.keys))))
                    elif sig.what != "cl" and not getattr(sig, "rest", False):
                        self.report(where, "%s: %r where a keyword is expected" % (name, k))
            return
        if n < sig.lo or (sig.hi is not None and n > sig.hi):
            want = "%d" % sig.lo if sig.hi == sig.lo else ("%d..%s" % (sig.lo, "*" if sig.hi is None else sig.hi))
            self.report(where, "%s (%s) called with %d argument(s), takes %s" % (name, sig.what, n, want))

    def check_type(self, t, where):
        """a type specifier as data (the operand of quote)"""
        if isinstance(t, Sym):
            if t not in CL_TYPES and t not in self.classes and not self.is_ref_condition(t):
                self.report(where, "unknown type %s" % t)
        elif isinstance(t, list) and t:
            head = t[0]
            if head in ("or", "and", "not", "member", "eql", "satisfies", "values", "function", "integer", "mod", "real", "float",
                        "double-float", "single-float", "rational"):
                if head in ("or", "and", "not"):
                    for x in t[1:]:
                        self.check_type(x, where)
                return
            if head in ("simple-array", "array", "vector", "simple-vector"):
                if len(t) > 1 and t[1] != "*":
                    self.check_type(t[1], where)
                return
            if head in ("signed-byte", "unsigned-byte", "cons", "complex", "string", "simple-string"):
                return
            self.report(where, "unknown compound type (%s ...)" % head)

    def is_ref_condition(self, name):
        return name in self.imports and (self.imports[name], str(name)) in self.ref_conditions

    def check_initargs(self, cls, args, where, what):
        """args: the initarg / value list of make-instance / make-condition / error"""
        if cls in self.classes:
            known = set()
            todo, seen = [cls], set()
            unknown_parent = False
            while todo:
                c = todo.pop()
                if c in seen:
                    continue
                seen.add(c)
                if c in self.classes:
                    known |= self.classes[c][0]
                    todo += self.classes[c][1]
                elif self.is_ref_condition(c):
                    known |= set(self.ref_conditions[(self.imports[c], c)]["initargs"])
                    self.used.add(c)
                elif c not in CL_TYPES:
                    unknown_parent = True
            if unknown_parent:
                return
        elif self.is_ref_condition(cls):
            self.used.add(str(cls))
            known = set(self.ref_conditions[(self.imports[cls], str(cls))]["initargs"])
        elif cls in CL_CONDITIONS:
            known = {":format-control", ":format-arguments", ":datum", ":expected-type", ":name", ":operation", ":operands"}
        else:
            self.report(where, "%s names %s, which is neither defined here, nor imported, nor a standard class" % (what, cls))
            return
        if len(args) % 2:
            self.report(where, "%s %s: odd number of initargs" % (what, cls))
        for k in args[0::2]:
            if is_keyword(k) and str(k) not in known:
                self.report(where, "%s %s: no slot takes the initarg %s (known: %s)" % (what, cls, k, " ".join(sorted(known)) or "none"))

    # ---- pass 2
    def run(self):
        self.collect()
        top = Env()
        for f in self.forms:
            self.toplevel(f, top)
        for name, sym in self.exports:
            if name not in self.functions and name not in self.classes and name not in self.macros and name not in self.globals:
                self.report(sym, "the package exports %s, which the file does not define" % name)
        return self.findings

    def toplevel(self, f, env):
        if not isinstance(f, list) or not f:
            return
        op = f[0]
        if op in ("defpackage", "in-package", "cffi:define-foreign-library", "cffi:defcfun", "declaim"):
            if op == "cffi:defcfun":
                ret = f[2]
                if ret not in CFFI_TYPES:
                    self.report(f, "defcfun %s: unknown foreign type %s" % (f[1], ret))
                for a in f[3:]:
                    if not (isinstance(a, list) and len(a) == 2 and a[1] in CFFI_TYPES):
                        self.report(f, "defcfun %s: argument %r is not (name foreign-type)" % (f[1], a))
            return
        if op in ("defconstant", "defvar", "defparameter"):
            if len(f) > 2:
                self.walk(f[2], env)
            return
        if op == "defun":
            e = env.child()
            e.blocks.add(str(f[1]))
            self.walk_lambda(f[2], f[3:], e)
            return
        if op == "defmethod":
            i = 2
            while not isinstance(f[i], list):
                i += 1                                          # (qualifiers)
            ll = f[i]
            name = str(f[1])
            sig_m = parse_lambda_list(ll, specialized=True)[0]
            generic = self.function_sig(name, env, f)
            if generic is not None and name in self.imports and (generic.lo != sig_m.lo):
                self.report(f, "method on %s takes %d required argument(s), the generic function %d" % (name, sig_m.lo, generic.lo))
            for item in ll:
                if isinstance(item, list) and isinstance(item[0], Sym) and item[0] not in LL_KEYWORDS and len(item) == 2 \
                        and isinstance(item[1], Sym):
                    self.check_type(item[1], f)
            e = env.child()
            e.blocks.add(name)
            self.walk_lambda(ll, f[i + 1:], e, specialized=True)
            return
        if op == "defmacro":
            e = env.child()
            _, binds = parse_lambda_list(f[2])
            for v, _d in binds:
                e.vars.add(str(v))
            for b in f[3:]:
                self.walk_template_or_form(b, e)
            return
        if op in ("define-condition", "defclass"):
            for p in f[2]:
                if p not in self.classes and p not in CL_TYPES and not self.is_ref_condition(p):
                    self.report(f, "%s %s: unknown parent %s" % (op, f[1], p))
                if self.is_ref_condition(p):
                    self.used.add(str(p))
            for opt in f[4:]:
                if isinstance(opt, list) and opt and opt[0] == ":report" and isinstance(opt[1], list):
                    self.walk(opt[1], env)
            return
        if op == "cffi:use-foreign-library":
            return
        self.walk(f, env)

    def walk_template_or_form(self, f, env):
        """a macro's body: ordinary forms, except that inside a backquote only the unquoted parts are code"""
        if isinstance(f, list) and f and f[0] == "quasiquote":
            self.walk_quasi(f[1], env)
        elif isinstance(f, Str):
            return
        else:
            self.walk(f, env)

    def walk_quasi(self, f, env):
        if isinstance(f, list):
            if f and f[0] in ("unquote", "unquote-splicing"):
                self.walk(f[1], env)
                return
            for x in f:
                self.walk_quasi(x, env)

    def body(self, forms, env):
        """an implicit progn that may start with a documentation string and declarations"""
        forms = list(forms)
        if len(forms) > 1 and isinstance(forms[0], Str):
            forms = forms[1:]
        for f in forms:
            if isinstance(f, list) and f and f[0] == "declare":
                for d in f[1:]:
                    if d and d[0] in ("ignore", "ignorable"):
                        for v in d[1:]:
                            if isinstance(v, Sym) and not env.has_var(v):
                                self.report(f, "(declare (%s %s)): no such variable in scope" % (d[0], v))
                continue
            self.walk(f, env)

    def walk_lambda(self, ll, body, env, specialized=False):
        _, binds = parse_lambda_list(ll, specialized)
        for v, default in binds:
            if default is not None:
                self.walk(default, env)
            env.vars.add(str(v))
        self.body(body, env)

    def walk_place(self, place, env, where):
        if isinstance(place, Sym):
            self.walk(place, env)
            return
        if isinstance(place, list) and place and isinstance(place[0], Sym):
            head = str(place[0])
            if head not in CL_SETF_PLACES and head not in self.setf_places:
                self.report(where, "(setf (%s ...)): no setf expansion is known for %s" % (head, head))
            if head in ("cffi:mem-ref", "cffi:mem-aref") or head in CL_SETF_PLACES and head not in self.setf_places:
                self.walk(place, env)
            else:
                for a in place[1:]:
                    self.walk(a, env)
            return
        self.report(where, "setf of %r" % (place,))

    def walk(self, f, env):
        if isinstance(f, Sym):
            if f.startswith(":") or f in CL_CONSTANTS:
                return
            if env.has_var(f) or f in self.globals:
                return
            if ":" in f:
                pkg, _, s = f.partition(":")
                if pkg.startswith("linear-programming"):
                    return
            self.report(f, "unbound variable %s" % f)
            return
        if not isinstance(f, list):
            return
        if not f:
            return                                               # () = NIL
        head = f[0]
        if isinstance(head, list):
            if head and head[0] == "lambda":
                self.walk(head, env)
                for a in f[1:]:
                    self.walk(a, env)
            else:
                self.report(f, "a list in operator position: %r" % (head,))
            return
        if not isinstance(head, Sym):
            self.report(f, "%r in operator position" % (head,))
            return
        op = str(head)
        args = f[1:]
        handler = getattr(self, "op_" + re.sub(r"[^a-z0-9]", "_", op), None)
        if op in SPECIAL and handler:
            handler(f, args, env)
            return
        if op in CL_PROGN_LIKE:
            self.check_call(op, CL_PROGN_LIKE[op], args, f)
            for a in args:
                self.walk(a, env)
            return
        if op in self.macros:
            sig, ll = self.macros[op]
            self.check_call(op, sig, args, f)
            for a in args:                                       # (&body macros: the arguments are forms)
                self.walk(a, env)
            return
        sig = self.function_sig(op, env, f)
        self.check_call(op, sig, args, f)
        for a in args:
            self.walk(a, env)

    # ---- special operators and macros with their own syntax
    def op_quote(self, f, args, env):
        pass

    def op_function(self, f, args, env):
        x = args[0]
        if isinstance(x, Sym):
            self.function_sig(str(x), env, f)
        elif isinstance(x, list) and x and x[0] == "lambda":
            self.walk(x, env)
        elif isinstance(x, list) and x and x[0] == "setf":
            pass
        else:
            self.report(f, "#'%r" % (x,))

    def op_lambda(self, f, args, env):
        self.walk_lambda(args[0], args[1:], env.child())

    def op_let(self, f, args, env):
        e = env.child()
        for b in args[0]:
            if isinstance(b, list):
                if len(b) > 2:
                    self.report(f, "let binding %s has %d forms" % (b[0], len(b) - 1))
                if len(b) > 1:
                    self.walk(b[1], env)
                e.vars.add(str(b[0]))
            else:
                e.vars.add(str(b))
        self.body(args[1:], e)

    def op_let_(self, f, args, env):
        e = env.child()
        for b in args[0]:
            if isinstance(b, list):
                if len(b) > 2:
                    self.report(f, "let* binding %s has %d forms" % (b[0], len(b) - 1))
                if len(b) > 1:
                    self.walk(b[1], e)
                e = e.child()
                e.vars.add(str(b[0]))
            else:
                e = e.child()
                e.vars.add(str(b))
        self.body(args[1:], e)

    def _flet(self, f, args, env, recursive):
        e = env.child()
        for d in args[0]:
            sig, _ = parse_lambda_list(d[1])
            sig.what = "local function"
            e.funcs[str(d[0])] = sig
        for d in args[0]:
            inner = (e if recursive else env).child()
            inner.blocks.add(str(d[0]))
            self.walk_lambda(d[1], d[2:], inner)
        self.body(args[1:], e)

    def op_flet(self, f, args, env):
        self._flet(f, args, env, False)

    def op_labels(self, f, args, env):
        self._flet(f, args, env, True)

    def op_setf(self, f, args, env):
        if len(args) % 2:
            self.report(f, "setf with an odd number of arguments")
        for place, value in zip(args[0::2], args[1::2]):
            self.walk_place(place, env, f)
            self.walk(value, env)

    op_setq = op_setf

    def _modify(self, f, args, env):
        self.walk_place(args[0], env, f)
        for a in args[1:]:
            self.walk(a, env)

    op_incf = op_decf = _modify

    def op_push(self, f, args, env):
        self.walk(args[0], env)
        self.walk_place(args[1], env, f)

    def op_pop(self, f, args, env):
        self.walk_place(args[0], env, f)

    def op_cond(self, f, args, env):
        for clause in args:
            if not isinstance(clause, list) or not clause:
                self.report(f, "cond clause %r" % (clause,))
                continue
            for x in clause:
                self.walk(x, env)

    def _case(self, f, args, env, types=False):
        self.walk(args[0], env)
        for clause in args[1:]:
            if not isinstance(clause, list) or not clause:
                self.report(f, "case clause %r" % (clause,))
                continue
            if types and clause[0] not in ("t", "otherwise"):
                self.check_type(clause[0], f)
            for x in clause[1:]:
                self.walk(x, env)

    def op_case(self, f, args, env):
        self._case(f, args, env)

    op_ecase = op_ccase = op_case

    def op_typecase(self, f, args, env):
        self._case(f, args, env, True)

    op_etypecase = op_typecase

    def op_dotimes(self, f, args, env):
        spec = args[0]
        self.walk(spec[1], env)
        e = env.child()
        e.vars.add(str(spec[0]))
        e.blocks.add("nil")
        if len(spec) > 2:
            self.walk(spec[2], e)
        self.body(args[1:], e)

    op_dolist = op_dotimes

    def op_multiple_value_bind(self, f, args, env):
        self.walk(args[1], env)
        e = env.child()
        for v in args[0]:
            e.vars.add(str(v))
        self.body(args[2:], e)

    def op_destructuring_bind(self, f, args, env):
        self.walk(args[1], env)
        e = env.child()
        for v in flatten_vars(args[0]):
            e.vars.add(str(v))
        self.body(args[2:], e)

    def op_handler_case(self, f, args, env):
        self.walk(args[0], env)
        for clause in args[1:]:
            if clause[0] == ":no-error":
                self.walk_lambda(clause[1], clause[2:], env.child())
                continue
            self.check_type(clause[0], f)
            e = env.child()
            for v in clause[1]:
                e.vars.add(str(v))
            self.body(clause[2:], e)

    def op_handler_bind(self, f, args, env):
        for b in args[0]:
            self.check_type(b[0], f)
            self.walk(b[1], env)
        self.body(args[1:], env)

    def op_block(self, f, args, env):
        e = env.child()
        e.blocks.add(str(args[0]))
        self.body(args[1:], e)

    def op_return(self, f, args, env):
        if not env.has_block("nil"):
            self.report(f, "(return) outside a block named NIL")
        for a in args:
            self.walk(a, env)

    def op_return_from(self, f, args, env):
        if not env.has_block(str(args[0])):
            self.report(f, "(return-from %s): no such block in scope" % args[0])
        for a in args[1:]:
            self.walk(a, env)

    def op_the(self, f, args, env):
        self.check_type(args[0], f)
        self.walk(args[1], env)

    def op_declare(self, f, args, env):
        pass

    def op_check_type(self, f, args, env):
        self.walk_place(args[0], env, f)
        self.check_type(args[1], f)

    def op_typep(self, f, args, env):
        self.walk(args[0], env)
        q = args[1] if len(args) > 1 else None
        if isinstance(q, list) and q and q[0] == "quote":
            self.check_type(q[1], f)
        else:
            self.walk(q, env)

    def op_coerce(self, f, args, env):
        self.check_call("coerce", CL_FUNCTIONS["coerce"], args, f)
        self.op_typep(f, args, env)

    def op_make_array(self, f, args, env):
        self.check_call("make-array", CL_FUNCTIONS["make-array"], args, f)
        self.walk(args[0], env)
        for k, v in zip(args[1::2], args[2::2]):
            if k == ":element-type" and isinstance(v, list) and v and v[0] == "quote":
                self.check_type(v[1], f)
            else:
                self.walk(v, env)

    def _signal(self, f, args, env, what):
        first = args[0] if args else None
        cls = quoted_symbol(first)
        if cls is not None:
            self.check_initargs(str(cls), args[1:], f, what)
        else:
            self.walk(first, env)
        for a in args[1:]:
            self.walk(a, env)

    def op_error(self, f, args, env):
        self._signal(f, args, env, "error")

    def op_signal(self, f, args, env):
        self._signal(f, args, env, "signal")

    def op_warn(self, f, args, env):
        self._signal(f, args, env, "warn")

    def op_make_condition(self, f, args, env):
        self._signal(f, args, env, "make-condition")

    def op_make_instance(self, f, args, env):
        self._signal(f, args, env, "make-instance")

    def op_loop(self, f, args, env):
        e = env.child()
        e.blocks.add("nil")
        if all(isinstance(a, list) for a in args):               # the simple loop
            for a in args:
                self.walk(a, e)
            return
        # extended loop: variables first (for / as / with / into), then every other non-keyword item is a form
        var_positions = set()
        for i, a in enumerate(args):
            if isinstance(a, Sym) and a in ("for", "as", "with", "into") and i + 1 < len(args):
                var_positions.add(i + 1)
                for v in flatten_vars(args[i + 1]):
                    e.vars.add(str(v))
            if isinstance(a, Sym) and a == "and" and i + 2 < len(args) and isinstance(args[i + 2], Sym) and args[i + 2] in ("=", "in", "from", "across", "on", "below"):
                var_positions.add(i + 1)
                for v in flatten_vars(args[i + 1]):
                    e.vars.add(str(v))
            if isinstance(a, Sym) and a == "named" and i + 1 < len(args):
                var_positions.add(i + 1)
                e.blocks.add(str(args[i + 1]))
        for i, a in enumerate(args):
            if i in var_positions:
                continue
            if isinstance(a, Sym):
                if a in LOOP_KEYWORDS:
                    continue
                if i == 0 or not isinstance(args[i - 1], Sym) or args[i - 1] not in LOOP_KEYWORDS:
                    self.report(a, "loop: %s is neither a loop keyword nor in a position where a form is expected" % a)
                    continue
            self.walk(a, e)

    # ---- CFFI / SBCL macros
    def _foreign_spec(self, spec, env, e, f):
        if not (isinstance(spec, list) and 2 <= len(spec) <= 3):
            self.report(f, "foreign object spec %r is not (var type [count])" % (spec,))
            return
        t = spec[1]
        if is_keyword(t):
            if str(t) not in CFFI_TYPES:
                self.report(f, "unknown foreign type %s" % t)
        else:
            self.walk(t, env)
        if len(spec) == 3:
            self.walk(spec[2], env)
        e.vars.add(str(spec[0]))

    def op_cffi_with_foreign_object(self, f, args, env):
        e = env.child()
        self._foreign_spec(args[0], env, e, f)
        self.body(args[1:], e)

    def op_cffi_with_foreign_objects(self, f, args, env):
        e = env.child()
        for spec in args[0]:
            self._foreign_spec(spec, e, e, f)
        self.body(args[1:], e)

    def op_cffi_with_pointer_to_vector_data(self, f, args, env):
        spec = args[0]
        if not (isinstance(spec, list) and len(spec) == 2 and isinstance(spec[0], Sym)):
            self.report(f, "with-pointer-to-vector-data spec %r is not (pointer-var vector)" % (spec,))
            return
        self.walk(spec[1], env)
        e = env.child()
        e.vars.add(str(spec[0]))
        self.body(args[1:], e)

    def _mem(self, f, args, env):
        self.check_call(str(f[0]), FOREIGN_FUNCTIONS[str(f[0])], args, f)
        self.walk(args[0], env)
        if len(args) > 1:
            if is_keyword(args[1]):
                if str(args[1]) not in CFFI_TYPES:
                    self.report(f, "unknown foreign type %s" % args[1])
            else:
                self.walk(args[1], env)
        for a in args[2:]:
            self.walk(a, env)

    op_cffi_mem_ref = op_cffi_mem_aref = _mem

    def op_sb_int_with_float_traps_masked(self, f, args, env):
        for t in args[0]:
            if str(t) not in (":overflow", ":invalid", ":divide-by-zero", ":inexact", ":underflow"):
                self.report(f, "unknown floating point trap %s" % t)
        self.body(args[1:], env)


SPECIAL = {"quote", "function", "lambda", "let", "let*", "flet", "labels", "setf", "setq", "incf", "decf", "push", "pop", "cond",
           "case", "ecase", "ccase", "typecase", "etypecase", "dotimes", "dolist", "multiple-value-bind", "destructuring-bind",
           "handler-case", "handler-bind", "block", "return", "return-from", "the", "declare", "check-type", "typep", "coerce",
           "make-array", "error", "signal", "warn", "make-condition", "make-instance", "loop", "cffi:with-foreign-object",
           "cffi:with-foreign-objects", "cffi:with-pointer-to-vector-data", "cffi:mem-ref", "cffi:mem-aref",
           "sb-int:with-float-traps-masked"}


# ------------------------------------------------------------------------------------------- the reference's side
def reference_signatures(ref):
    """Lambda lists (as call constraints) of every function, generic function and structure accessor, and the
    initargs of every condition, that the reference's hot-path packages define -- read with the reader above.
    DATA about the reference's interface (names, argument counts, keywords), not its source."""
    out = {}
    for stem in ("simplex", "problem", "solver", "conditions"):
        forms = read_file(os.path.join(ref, "src", stem + ".lisp"))
        pkg = next(str(f[1]).lstrip(":#") for f in forms if isinstance(f, list) and f and f[0] in ("defpackage", "uiop:define-package"))
        functions, conditions = {}, {}
        for f in forms:
            if not isinstance(f, list) or not f or not isinstance(f[0], Sym):
                continue
            op = f[0]
            if op in ("defun", "defgeneric", "defmacro") and isinstance(f[1], Sym):
                sig = parse_lambda_list(f[2])[0]
                sig.what = {"defun": "reference function", "defgeneric": "reference generic function", "defmacro": "reference macro"}[op]
                d = sig.to_json()
                d["positional"] = sig.positional
                functions[str(f[1])] = d
            elif op == "defstruct":
                name = f[1][0] if isinstance(f[1], list) else f[1]
                conc = str(name) + "-"
                if isinstance(f[1], list):
                    for o in f[1][1:]:
                        if isinstance(o, list) and o[0] == ":conc-name":
                            conc = str(o[1]) if len(o) > 1 and o[1] != "nil" else ""
                for slot in f[2:]:
                    if isinstance(slot, Str):
                        continue
                    sname = slot[0] if isinstance(slot, list) else slot
                    functions[conc + str(sname)] = {"min": 1, "max": 1, "keys": None, "allow_other_keys": False,
                                                    "what": "reference structure accessor", "positional": 1}
            elif op == "define-condition":
                initargs = []
                for slot in f[3]:
                    opts = slot[1:] if isinstance(slot, list) else []
                    for k, v in zip(opts[0::2], opts[1::2]):
                        if k == ":initarg":
                            initargs.append(str(v))
                        elif k in (":reader", ":accessor"):
                            functions[str(v)] = {"min": 1, "max": 1, "keys": None, "allow_other_keys": False,
                                                 "what": "reference condition reader", "positional": 1}
                conditions[str(f[1])] = {"parents": [str(p) for p in f[2]], "initargs": sorted(initargs)}
        # a condition takes its parents' initargs too
        for name, c in conditions.items():
            todo = list(c["parents"])
            while todo:
                p = todo.pop()
                if p in conditions:
                    c["initargs"] = sorted(set(c["initargs"]) | set(conditions[p]["initargs"]))
                    todo += conditions[p]["parents"]
        out[pkg] = {"file": "src/%s.lisp" % stem, "functions": functions, "conditions": conditions}
    return out


def load_signatures():
    data = json.load(open(SIGNATURES))
    for table in data.values():
        for d in table["functions"].values():
            d.setdefault("positional", d["max"] if d["max"] is not None else d["min"])
    return data


def lint(path=GLUE, reference=None, features=None, text=None):
    """-> (findings, linter).  features: the *features* the #+ / #- conditionals are read under (default: SBCL on
    Linux; without "sbcl" the portable branches are the ones walked).  text: the source itself instead of a path."""
    global FEATURES
    saved = FEATURES
    if features is not None:
        FEATURES = set(features)
    try:
        forms = Reader(text).read_all() if text is not None else read_file(path)
    finally:
        FEATURES = saved
    linter = Linter(forms, load_signatures() if reference is None else reference)
    findings = linter.run()
    # Sig.from_json drops `positional`: restore it for keyword checks of reference functions
    return findings, linter


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--signatures":
        ref = sys.argv[2] if len(sys.argv) > 2 else "/root/reference"
        data = reference_signatures(ref)
        json.dump(data, open(SIGNATURES, "w"), indent=1, sort_keys=True)
        print(SIGNATURES, {k: (len(v["functions"]), len(v["conditions"])) for k, v in data.items()})
        sys.exit(0)
    found, L = lint(sys.argv[1] if len(sys.argv) > 1 else GLUE)
    for line in found:
        print(line)
    print("%d finding(s); %d top-level forms, %d functions / bindings, %d macros, %d classes / conditions" %
          (len(found), len(L.forms), len(L.functions), len(L.macros), len(L.classes)))
    sys.exit(1 if found else 0)
