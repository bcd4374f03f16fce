"""ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

ctypes loader for ``liboracle.so`` (oracle/simplex_oracle.c, the plain-C CPU
restatement of the reference's double-float simplex hot path,
src/simplex.lisp:337-461) plus thin numpy wrappers.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / reported CPU baseline.
Nothing under ``linear-programming_amd/`` imports it.

Parity status: pinned against the reference's own known-answer tests
(tests/golden/reference_cases.json; see tests/test_oracle_golden.py).  The
reference itself is Common Lisp and cannot be compiled or run in this image,
so there is no ``oracle/_ref``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

OPTIMAL, UNBOUNDED, INFEASIBLE, MAX_PIVOTS, ART_NONZERO, ART_STUCK = 0, 1, 2, 3, 4, 5
EPSILON = 1.1102230246251568e-16  # CL double-float-epsilon


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "simplex_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, dbl, p = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
        L.orc_epsilon.restype = dbl
        L.orc_omp_threads.restype = ctypes.c_int
        L.orc_set_omp_threads.restype = ctypes.c_int
        L.orc_set_omp_threads.argtypes = [ctypes.c_int]
        L.orc_price.restype = i64
        L.orc_price.argtypes = [p, i64, i64, i64, ctypes.c_int, dbl]
        L.orc_ratio.restype = i64
        L.orc_ratio.argtypes = [p, i64, i64, i64, i64, dbl]
        for f in (L.orc_pivot, L.orc_pivot_omp):
            f.restype = None
            f.argtypes = [p, i64, i64, i64, p, i64, i64]
        L.orc_solve.restype = ctypes.c_int
        L.orc_solve.argtypes = [p, i64, i64, i64, p, ctypes.c_int, dbl, i64, p, p, p, i64,
                                ctypes.c_int]
        L.orc_solve_two_phase.restype = ctypes.c_int
        L.orc_solve_two_phase.argtypes = [p, i64, i64, i64, p, p, i64, i64, p, ctypes.c_int,
                                          dbl, p]
        for f in (L.orc_fp_eq, L.orc_fp_lt, L.orc_fp_gt, L.orc_fp_le, L.orc_fp_ge):
            f.restype = ctypes.c_int
            f.argtypes = [dbl, dbl, dbl]
        _lib = L
    return _lib


def omp_threads():
    """Threads the OpenMP variant uses."""
    return int(lib().orc_omp_threads())


def usable_cores():
    """Cores this process may really use: min(affinity, cgroup CPU quota).  OpenMP only sees the former."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:                                                   # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def set_omp_threads(n):
    """Threads the OpenMP variant uses from now on (returns what OpenMP reports afterwards)."""
    return int(lib().orc_set_omp_threads(int(n)))


def _chk(M, basis=None):
    assert M.dtype == np.float64 and M.ndim == 2 and M.flags.c_contiguous
    if basis is not None:
        assert basis.dtype == np.int64 and basis.flags.c_contiguous
        assert basis.shape[0] == M.shape[0] - 1


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def price(M, is_max=True, factor=1024.0):
    """find-entering-column (src/simplex.lisp:362-379); -1 for NIL."""
    _chk(M)
    R, C = M.shape
    return int(lib().orc_price(_ptr(M), C, R - 1, C - 1, int(bool(is_max)), float(factor)))


def ratio(M, ec, factor=1024.0):
    """find-pivoting-row (src/simplex.lisp:382-389); -1 for NIL."""
    _chk(M)
    R, C = M.shape
    return int(lib().orc_ratio(_ptr(M), C, R - 1, C - 1, int(ec), float(factor)))


def pivot(M, basis, ec, cr, omp=False):
    """n-pivot-row (src/simplex.lisp:337-359), in place."""
    _chk(M, basis)
    R, C = M.shape
    f = lib().orc_pivot_omp if omp else lib().orc_pivot
    f(_ptr(M), C, R, C, _ptr(basis), int(ec), int(cr))


def solve(M, basis, is_max=True, factor=1024.0, max_pivots=0, trace_cap=0, omp=False):
    """n-solve-tableau single phase (src/simplex.lisp:453-461), in place.

    Returns (status, n_pivots, trace) with trace an (n, 2) int64 array of
    (entering column, pivot row) for the first ``trace_cap`` pivots."""
    _chk(M, basis)
    R, C = M.shape
    n = ctypes.c_int64(0)
    tec = np.full(max(trace_cap, 1), -1, dtype=np.int64)
    tcr = np.full(max(trace_cap, 1), -1, dtype=np.int64)
    st = lib().orc_solve(_ptr(M), C, R, C, _ptr(basis), int(bool(is_max)), float(factor),
                         int(max_pivots), ctypes.byref(n), _ptr(tec), _ptr(tcr),
                         int(trace_cap), int(bool(omp)))
    k = min(int(n.value), trace_cap)
    return int(st), int(n.value), np.stack([tec[:k], tcr[:k]], axis=1)


def solve_two_phase(art, art_basis, main, main_basis, main_is_max=True, factor=1024.0):
    """n-solve-tableau two-phase branch (src/simplex.lisp:402-452), in place."""
    _chk(art, art_basis)
    _chk(main, main_basis)
    assert art.shape[0] == main.shape[0]
    npv = np.zeros(2, dtype=np.int64)
    st = lib().orc_solve_two_phase(_ptr(art), art.shape[1], art.shape[0], art.shape[1],
                                   _ptr(art_basis), _ptr(main), main.shape[1], main.shape[1],
                                   _ptr(main_basis), int(bool(main_is_max)), float(factor),
                                   _ptr(npv))
    return int(st), npv


def fp_compare(fn, a, b, factor=16.0):
    f = {"fp=": lib().orc_fp_eq, "fp<": lib().orc_fp_lt, "fp>": lib().orc_fp_gt,
         "fp<=": lib().orc_fp_le, "fp>=": lib().orc_fp_ge}[fn]
    return bool(f(float(a), float(b), float(factor)))
