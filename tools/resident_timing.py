"""Where the resident solve (k_resident) spends a pivot.  Builds an instrumented copy of the
library (-DMI355X_RES_TIMING: the leader thread accumulates wall_clock64 deltas of the phases of
every pivot in the handle's otherwise unused `rhs` buffer), loads THAT copy and solves config 2.

    python tools/resident_timing.py [n_vars n_cons]
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build
out = os.path.join(ROOT, "gpurun_out", "libmi355x_simplex_res_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in _build.sources()):
    _build.build(extra_flags=["-DMI355X_RES_TIMING"], out=out)
os.environ["MI355X_SIMPLEX_LIB"] = out
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 512)
for rep in range(3):
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(2, rep), 0, -1, 0), "create")
    npv = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 0, 1), "prepare")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    acc = np.zeros(512)
    nacc = min(512, (m + 1) // 16 * 16)          # (the buffer holds one double per tableau row)
    L.mi355x_debug_rhs(h, acc.ctypes.data_as(ctypes.c_void_p), nacc, 1)
    t0 = time.perf_counter()
    rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(npv))
    dt = time.perf_counter() - t0
    L.mi355x_debug_rhs(h, acc.ctypes.data_as(ctypes.c_void_p), nacc, 0)
    L.mi355x_tab_destroy(h)
    c = max(acc[0], 1)
    ph = acc[1:8] / c * 0.01
    print("%d x %d: rc %d, %d pivots in %.3f ms = %.2f us/pivot (%.0f pivots/s) | us per pivot, leader: polls issued + strip update %.2f  records %.2f  column + owner's pivot row %.2f  "
          "prow %.2f  obj+pricing+publish %.2f  rhs + ratio test ahead %.2f  bookkeeping %.2f  = %.2f" % (
              n, m, rc, npv.value, dt * 1e3, dt / max(npv.value, 1) * 1e6, npv.value / dt, *ph, ph.sum()))
    per = acc.reshape(32, 16)
    live = per[:, 0] > 0
    if live.sum() > 1 and rep == 2:
        print("      per workgroup, us per pivot waiting for records | share of pivots as the owner of the entering column:")
        print("      " + "  ".join("%d: %.2f|%.0f%%" % (w, per[w, 2] / per[w, 0] * 0.01, 100 * per[w, 12] / per[w, 0]) for w in range(32) if live[w]))
    print("      inside obj+pricing+publish: objective entries %.2f  pricing %.2f  candidate column %.2f  publish %.2f" % tuple(acc[8:12] / c * 0.01))
