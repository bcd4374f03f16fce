"""Where the persistent look-ahead kernel (k_la_block) spends a step: needs the library built
with -DMI355X_LA_TIMING (the leader thread then accumulates wall_clock64 deltas of the six phases
of every step in the handle's otherwise unused `rhs` buffer).

    hipcc ... -DMI355X_LA_TIMING ...   (see linear-programming_amd/build.py for the flags)
    python tools/la_timing.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
L.mi355x_debug_rhs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
n, m = 8192, 4096
h = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
npv = ctypes.c_int64(0)
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
L.mi355x_tab_sync(h, ctypes.byref(npv))
out = np.zeros(128)
L.mi355x_debug_rhs(h, out.ctypes.data_as(ctypes.c_void_p), 128, 1)
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 1600, 0), "run")
L.mi355x_tab_sync(h, ctypes.byref(npv))
L.mi355x_debug_rhs(h, out.ctypes.data_as(ctypes.c_void_p), 128, 0)
d = out.reshape(16, 8)
if d[:, 6].sum() == 0:
    sys.exit("no samples: the library was built without -DMI355X_LA_TIMING")
print("us per step: price-reduce | price-exchange | column+chain | ratio-reduce | ratio-exchange | row+chain+bookkeeping")
for J in range(16):
    c = d[J, 6]
    print(J, int(c), " ".join("%7.2f" % (d[J, k] / c * 0.01) for k in range(6)), "  total %.2f" % (d[J, :6].sum() / c * 0.01))
