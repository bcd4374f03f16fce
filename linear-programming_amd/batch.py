"""Batches of independent same-shape LPs (BASELINE config 4) over the mi355x_batch_* entry
points: n-solve-tableau (src/simplex.lisp:453-461) applied to every tableau of the batch, one
(select, update) launch pair advancing all unfinished LPs by one pivot.  Independent units: to
use several GPUs give every rank its own sub-batch (no communication)."""
import ctypes

import numpy as np

from . import capi


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class TableauBatch:
    def __init__(self, handle, n_lps, rows, cols):
        self._h, self.n_lps, self.rows, self.cols = handle, int(n_lps), int(rows), int(cols)

    @classmethod
    def from_arrays(cls, matrices, bases, device=0):
        """matrices: (n_lps, rows, cols) float64; bases: (n_lps, rows-1) int64."""
        M = np.ascontiguousarray(matrices, dtype=np.float64)
        B = np.ascontiguousarray(bases, dtype=np.int64)
        n, R, C = M.shape
        assert B.shape == (n, R - 1)
        h = ctypes.c_void_p()
        capi.check(capi.lib().mi355x_batch_create(ctypes.byref(h), n, R, C, _ptr(M), _ptr(B), device),
                   "mi355x_batch_create")
        return cls(h, n, R, C)

    @classmethod
    def synthetic(cls, n_lps, n_vars, n_cons, seeds, device=0):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (n_lps,)
        h = ctypes.c_void_p()
        capi.check(capi.lib().mi355x_batch_create_synthetic(ctypes.byref(h), n_lps, n_vars, n_cons,
                                                            _ptr(seeds), device),
                   "mi355x_batch_create_synthetic")
        return cls(h, n_lps, n_cons + 1, n_vars + n_cons + 1)

    def solve(self, is_max=True, fp_tolerance=1024, max_pivots=0):
        """Returns (status int32[n_lps], n_pivots int64[n_lps])."""
        st = np.zeros(self.n_lps, dtype=np.int32)
        npv = np.zeros(self.n_lps, dtype=np.int64)
        capi.check(capi.lib().mi355x_batch_solve(self._h, int(bool(is_max)), float(fp_tolerance),
                                                 int(max_pivots), _ptr(st), _ptr(npv)),
                   "mi355x_batch_solve")
        return st, npv

    def download(self, k):
        """(matrix, basis) of LP k."""
        M = np.empty((self.rows, self.cols), dtype=np.float64)
        b = np.empty(self.rows - 1, dtype=np.int64)
        capi.check(capi.lib().mi355x_batch_download(self._h, int(k), _ptr(M), _ptr(b), None, None),
                   "mi355x_batch_download")
        return M, b

    def objective_values(self):
        out = np.empty(self.n_lps)
        last = np.empty(self.rows)
        for k in range(self.n_lps):
            capi.check(capi.lib().mi355x_batch_download(self._h, k, None, None, None, _ptr(last)),
                       "mi355x_batch_download")
            out[k] = last[-1]
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                capi.lib().mi355x_batch_destroy(h)
            except Exception:
                pass
