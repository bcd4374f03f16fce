"""Where a wave of k_sweepw_ring spends a step: the first wave of every other workgroup accumulates, per step,
the time it waits for its rows (the counted s_waitcnt), the LDS read + the next requests, the links, and the issue of
its stores (-DMI355X_SWEEP_TIMING=2; wall_clock64, 10 ns).     python tools/sweep_phases.py [--skew N]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build
out = os.path.join(ROOT, "tools", "libmi355x_simplex_sweep_phases.so")
deps = _build.sources() + [os.path.join(_build.CSRC, f) for f in os.listdir(_build.CSRC) if f.endswith((".inc", ".h"))]
if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in deps):
    _build.build(extra_flags=["-DMI355X_SWEEP_TIMING=2"], out=out)
os.environ["MI355X_SIMPLEX_LIB"] = out
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
if "--skew" in sys.argv:
    L.mi355x_tune_set_sweep_skew(int(sys.argv[sys.argv.index("--skew") + 1]))
n, m = 8192, 4096
h = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
npv = ctypes.c_int64(0)
NS = m + 1
buf = np.zeros(NS)
if "--first" in sys.argv:                       # (a build whose arithmetic is wrong on purpose: only the first block's pass counts)
    L.mi355x_debug_rhs(h, buf.ctypes.data_as(ctypes.c_void_p), NS, 1)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 24, 1), "first block")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
else:
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    block = L.mi355x_tab_block_size(h)
    L.mi355x_debug_rhs(h, buf.ctypes.data_as(ctypes.c_void_p), NS, 1)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 4 * block, 0), "run")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
L.mi355x_debug_rhs(h, buf.ctypes.data_as(ctypes.c_void_p), NS, 0)
d = buf[:9 * (NS // 9)].reshape(-1, 9)
d = d[d[:, 2] > 0]
if not len(d):
    sys.exit("no samples: the library was built without -DMI355X_SWEEP_TIMING=2")
life = (d[:, 2] - d[:, 0]) * 0.01
pro = (d[:, 1] - d[:, 0]) * 0.01
steps = d[:, 4]
ph = d[:, 5:9] * 0.01
tile = (d[:, 3] // 17).astype(int)
third = np.minimum(tile // 15, 2)
print("%d workgroups sampled (first wave); per step [us]: wait for rows / LDS read + requests / links / stores issued" % len(d))
for g, name in ((0, "first third (oldest)"), (1, "second third"), (2, "last third (youngest)")):
    sel = third == g
    if sel.any():
        per = ph[sel] / steps[sel, None]
        print("  %-22s steps %4.1f  lifetime %5.1f  prologue %4.1f | %5.2f / %5.2f / %5.2f / %5.2f  = %5.2f per step" %
              (name, steps[sel].mean(), life[sel].mean(), pro[sel].mean(), *per.mean(axis=0), per.sum(axis=1).mean()))
per = ph / steps[:, None]
print("  all: %5.2f / %5.2f / %5.2f / %5.2f; of a workgroup's lifetime: waiting %.0f %%, links %.0f %%, prologue %.0f %%" %
      (*per.mean(axis=0), 100 * (ph[:, 0] / life).mean(), 100 * (ph[:, 2] / life).mean(), 100 * (pro / life).mean()))
L.mi355x_tab_destroy(h)
