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

    def solve_async(self, is_max=True, fp_tolerance=1024, max_pivots=0):
        """Start the solve on a worker thread of the library (mi355x_batch_solve_async)."""
        capi.check(capi.lib().mi355x_batch_solve_async(self._h, int(bool(is_max)), float(fp_tolerance),
                                                       int(max_pivots)), "mi355x_batch_solve_async")

    def sync(self):
        """Wait for solve_async; returns what solve() returns."""
        st = np.zeros(self.n_lps, dtype=np.int32)
        npv = np.zeros(self.n_lps, dtype=np.int64)
        capi.check(capi.lib().mi355x_batch_sync(self._h, _ptr(st), _ptr(npv)), "mi355x_batch_sync")
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


class MultiDeviceBatch:
    """One batch spread over several GPUs behind ONE handle (mi355x_multibatch_*): LP k lives in
    sub-batch k // ceil(n_lps / n_devices); solve() runs every sub-batch's loop on a worker thread
    of the library, so one host thread drives all devices (BASELINE config 4: 1024 LPs, 8 GPUs)."""

    def __init__(self, handle, n_lps, rows, cols):
        self._h, self.n_lps, self.rows, self.cols = handle, int(n_lps), int(rows), int(cols)

    @staticmethod
    def _devs(device_ids):
        if device_ids is None:
            return None, None
        arr = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
        return arr, arr

    @classmethod
    def from_arrays(cls, matrices, bases, n_devices, device_ids=None):
        M = np.ascontiguousarray(matrices, dtype=np.float64)
        B = np.ascontiguousarray(bases, dtype=np.int64)
        n, R, C = M.shape
        keep, devs = cls._devs(device_ids)
        h = ctypes.c_void_p()
        capi.check(capi.lib().mi355x_multibatch_create(ctypes.byref(h), n, R, C, _ptr(M), _ptr(B), int(n_devices),
                                                       devs), "mi355x_multibatch_create")
        return cls(h, n, R, C)

    @classmethod
    def synthetic(cls, n_lps, n_vars, n_cons, seeds, n_devices, device_ids=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        keep, devs = cls._devs(device_ids)
        h = ctypes.c_void_p()
        capi.check(capi.lib().mi355x_multibatch_create_synthetic(ctypes.byref(h), n_lps, n_vars, n_cons, _ptr(seeds),
                                                                 int(n_devices), devs),
                   "mi355x_multibatch_create_synthetic")
        return cls(h, n_lps, n_cons + 1, n_vars + n_cons + 1)

    def info(self):
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        capi.check(capi.lib().mi355x_multibatch_info(self._h, ctypes.byref(a), ctypes.byref(b)), "mi355x_multibatch_info")
        return {"n_sub_batches": a.value, "n_devices_used": b.value}

    def solve(self, is_max=True, fp_tolerance=1024, max_pivots=0):
        st = np.zeros(self.n_lps, dtype=np.int32)
        npv = np.zeros(self.n_lps, dtype=np.int64)
        capi.check(capi.lib().mi355x_multibatch_solve(self._h, int(bool(is_max)), float(fp_tolerance), int(max_pivots),
                                                      _ptr(st), _ptr(npv)), "mi355x_multibatch_solve")
        return st, npv

    def download(self, k):
        M = np.empty((self.rows, self.cols), dtype=np.float64)
        b = np.empty(self.rows - 1, dtype=np.int64)
        capi.check(capi.lib().mi355x_multibatch_download(self._h, int(k), _ptr(M), _ptr(b), None, None),
                   "mi355x_multibatch_download")
        return M, b

    def two_phase_handover(self, main, fp_tolerance=1024, phase1_status=None):
        """The step between the phases, member by member (mi355x_multibatch_two_phase_handover):
        self = the batch of artificial tableaux after phase 1.  -> (status per member: MI_OK = ready
        for phase 2, otherwise final; drive-out pivots per member)."""
        st = np.zeros(self.n_lps, dtype=np.int32)
        nd = np.zeros(self.n_lps, dtype=np.int64)
        p1 = None if phase1_status is None else np.ascontiguousarray(phase1_status, dtype=np.int32)
        capi.check(capi.lib().mi355x_multibatch_two_phase_handover(self._h, main._h, float(fp_tolerance),
                                                                   None if p1 is None else _ptr(p1), _ptr(st), _ptr(nd)),
                   "mi355x_multibatch_two_phase_handover")
        return st, nd

    def solve_two_phase(self, main, main_is_max=True, fp_tolerance=1024):
        """self = the batch of ARTIFICIAL tableaux, `main` the matching batch of main tableaux
        (n-solve-tableau's two-phase branch, src/simplex.lisp:402-452, member by member, drive-out
        pivots included): -> (status per member, pivots (n, 2))."""
        st = np.zeros(self.n_lps, dtype=np.int32)
        npv = np.zeros((self.n_lps, 2), dtype=np.int64)
        capi.check(capi.lib().mi355x_multibatch_solve_two_phase(self._h, main._h, int(bool(main_is_max)), float(fp_tolerance),
                                                                _ptr(st), _ptr(npv)), "mi355x_multibatch_solve_two_phase")
        return st, npv

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                capi.lib().mi355x_multibatch_destroy(h)
            except Exception:
                pass
