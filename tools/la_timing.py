"""Where the persistent look-ahead kernel (k_la_block) spends a step.  Builds an instrumented
copy of the library (-DMI355X_LA_TIMING: the leader thread accumulates wall_clock64 deltas of the
phases of every step in the handle's otherwise unused `rhs` buffer) next to the product library,
loads THAT copy and runs config 3.

    python tools/la_timing.py [one_xcd (1|0)] [n_vars n_cons]
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build
out = os.path.join(ROOT, "tools", "libmi355x_simplex_la_timing.so")   # (in-tree: travels to the GPU box; *.so is git-ignored)
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in _build.sources()):
    _build.build(extra_flags=["-DMI355X_LA_TIMING"], out=out)
os.environ["MI355X_SIMPLEX_LIB"] = out
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
one_xcd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8192, 4096)
L.mi355x_tune_set_la_one_xcd(one_xcd)
h = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
npv = ctypes.c_int64(0)
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
L.mi355x_tab_sync(h, ctypes.byref(npv))
NS = 24 * 24
out = np.zeros(NS)
L.mi355x_debug_rhs(h, out.ctypes.data_as(ctypes.c_void_p), NS, 1)
L.mi355x_tab_timing_enable(h, 1)
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 1680, 0), "run")
L.mi355x_tab_sync(h, ctypes.byref(npv))
L.mi355x_debug_rhs(h, out.ctypes.data_as(ctypes.c_void_p), NS, 0)
d = out.reshape(24, 24)
if d[:, 0].sum() == 0:
    sys.exit("no samples: the library was built without -DMI355X_LA_TIMING")
print("one_xcd=%d  %d x %d   us per step (leader thread):" % (one_xcd, n, m))
print(" J     n | price-xchg column+chain ratio-xchg row+chain+bk |  total || price: ->published  ->records in  ->result  extra polls || ratio: ...")
tot = 0.0
for J in range(24):
    c = d[J, 0]
    if c == 0:
        continue
    row = [d[J, k] / c * 0.01 for k in (1, 2, 3, 4)]
    tot += sum(row)
    px = [d[J, k] / c * 0.01 for k in (5, 6, 7)] + [d[J, 8] / c]
    rx = [d[J, k] / c * 0.01 for k in (9, 10, 11)] + [d[J, 12] / c]
    print("%2d %5d | " % (J, int(c)) + " ".join("%7.2f" % x for x in row) + " | %6.2f || " % sum(row)
          + " ".join("%5.2f" % x for x in px) + " || " + " ".join("%5.2f" % x for x in rx))
print("sum over the steps of a block: %.1f us" % tot)
for kind, name in ((1, "look-ahead"), (0, "sweep")):
    nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    L.mi355x_tab_timing_read_kind(h, kind, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
    print("%s: %d launches, avg %.1f us, min %.1f us (HIP events)" % (name, nl.value, sm.value / max(nl.value, 1) * 1e3, mn.value * 1e3))
