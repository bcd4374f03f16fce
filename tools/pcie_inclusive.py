"""End-to-end timing through the host-buffer boundary (config 3): upload of the dense tableau,
full solve, download -- the PCIe-inclusive rate noted in DESIGN.md (never bench.py's `value`)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = 8192, 4096
M, b = lp.synth.tableau(n, m, lp.synth.seed_for(3))
out = np.empty_like(M); bo = np.empty_like(b)
for rep in range(2):
    h = ctypes.c_void_p(); npv = ctypes.c_int64(0)
    t0 = time.perf_counter()
    lp.capi.check(L.mi355x_tab_create(ctypes.byref(h), M.shape[0], M.shape[1], M.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), 0), "create")
    t1 = time.perf_counter()
    rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(npv))
    t2 = time.perf_counter()
    lp.capi.check(L.mi355x_tab_download(h, out.ctypes.data_as(ctypes.c_void_p), bo.ctypes.data_as(ctypes.c_void_p), None, None), "download")
    t3 = time.perf_counter()
    L.mi355x_tab_destroy(h)
    print("rep %d: upload %.1f ms (%.1f GB/s), solve %.1f ms (%d pivots, rc %d), download %.1f ms (%.1f GB/s); "
          "pivots/s incl. PCIe %.0f vs %.0f solve-only" % (rep, (t1-t0)*1e3, M.nbytes/(t1-t0)/1e9, (t2-t1)*1e3, npv.value, rc,
          (t3-t2)*1e3, M.nbytes/(t3-t2)/1e9, npv.value/(t3-t0), npv.value/(t2-t1)), flush=True)
