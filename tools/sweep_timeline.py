"""When do the workgroups of one k_sweepw_ring pass start, finish their prologue and end?  Builds an
instrumented copy of the library (-DMI355X_SWEEP_TIMING: the first thread of every workgroup writes
wall_clock64 stamps, its XCC id and HW_ID into the handle's otherwise unused `rhs` buffer), runs config 3 and
prints the timeline of the LAST pass of the request: launch skew, prologue, end-time distribution (the tail a
static one-round schedule leaves), per XCD.

    python tools/sweep_timeline.py [n_vars n_cons] [--tr ROWS] [--xmap 0|1] [--skew ROWS]
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build
out = os.path.join(ROOT, "tools", "libmi355x_simplex_sweep_timing.so")   # (in-tree: travels to the GPU box; *.so is git-ignored)
os.makedirs(os.path.dirname(out), exist_ok=True)
deps = _build.sources() + [os.path.join(_build.CSRC, f) for f in os.listdir(_build.CSRC) if f.endswith((".inc", ".h"))]
if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in deps):
    _build.build(extra_flags=["-DMI355X_SWEEP_TIMING"], out=out)
os.environ["MI355X_SIMPLEX_LIB"] = out
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, m = (int(args[0]), int(args[1])) if len(args) >= 2 else (8192, 4096)
if "--tr" in sys.argv:
    L.mi355x_tune_set_sweep_shape(int(sys.argv[sys.argv.index("--tr") + 1]), -1)
if "--skew" in sys.argv:
    L.mi355x_tune_set_sweep_skew(int(sys.argv[sys.argv.index("--skew") + 1]))
if "--xmap" in sys.argv:
    L.mi355x_tune_set_sweep_xcd_map(int(sys.argv[sys.argv.index("--xmap") + 1]))
h = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
npv = ctypes.c_int64(0)
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
L.mi355x_tab_sync(h, ctypes.byref(npv))
block = L.mi355x_tab_block_size(h)
NS = m + 1
buf = np.zeros(NS)
for rep in range(3):
    L.mi355x_debug_rhs(h, buf.ctypes.data_as(ctypes.c_void_p), NS, 1)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 4 * block, 0), "run")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    L.mi355x_debug_rhs(h, buf.ctypes.data_as(ctypes.c_void_p), NS, 0)
    nwg = NS // 5
    d = buf[:5 * nwg].reshape(nwg, 5)
    d = d[d[:, 2] > 0]
    if not len(d):
        sys.exit("no samples: the library was built without -DMI355X_SWEEP_TIMING")
    t0 = d[:, 0].min()
    start, pro, end = (d[:, 0] - t0) * 0.01, (d[:, 1] - d[:, 0]) * 0.01, (d[:, 2] - t0) * 0.01
    life = end - start
    span = end.max()
    q = lambda a: " ".join("%6.1f" % np.percentile(a, p) for p in (0, 10, 50, 90, 100))
    print("pass %d: %d x %d, block %d, %d workgroups with samples; span (first start -> last end) %.1f us" % (rep, n, m, block, len(d), span))
    print("  percentiles 0/10/50/90/100 [us]: start %s | prologue %s | end %s | lifetime %s" % (q(start), q(pro), q(end), q(life)))
    print("  idle behind the workgroups' ends: %.1f %% of (workgroups x span);  started late (after 5 us): %d" %
          (100.0 * (span - end).sum() / (len(d) * span), int((start > 5).sum())))
    xcc = d[:, 3].astype(int)
    print("  per XCD (workgroups: mean end / max end): " + "  ".join("%d: %d %.1f/%.1f" % (x, (xcc == x).sum(), end[xcc == x].mean(), end[xcc == x].max())
                                                                    for x in sorted(set(xcc))))
    cu = (d[:, 4].astype(np.int64) >> 8) & 0xf
    se = (d[:, 4].astype(np.int64) >> 13) & 0x7
    key = xcc * 1000 + se * 16 + cu
    per_cu = np.array([end[key == k].max() for k in sorted(set(key))])
    print("  distinct (XCD, SE, CU) slots seen: %d; last end per slot percentiles: %s" % (len(per_cu), q(per_cu)))
np.save(os.path.join(ROOT, "gpurun_out", "sweep_timeline_last.npy"), d)   # (raw samples of the last pass, workgroup-linear order)
L.mi355x_tab_destroy(h)
