"""Which waves of the persistent look-ahead arrive late at an exchange: publish time of every wave's
record, per exchange of the last block (instrumented build).   python tools/la_skew.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build
out = os.path.join(ROOT, "tools", "libmi355x_simplex_la_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
out = out.replace("la_timing", "la_skew")
_build.build(extra_flags=["-DMI355X_LA_TIMING=2"], out=out)
os.environ["MI355X_SIMPLEX_LIB"] = out
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = 8192, 4096
h = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
npv = ctypes.c_int64(0)
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 160, 1), "run")
L.mi355x_tab_sync(h, ctypes.byref(npv))
N = 512 + 32 * 72
buf = np.zeros(N)
L.mi355x_debug_rhs(h, buf.ctypes.data_as(ctypes.c_void_p), N, 0)
d = buf[512:].reshape(32, 72)[:, :68] * 0.01          # us
t0 = d.min()
print("exchange (step, kind): first publish (us since block start) | spread last-first | median-first | 5 latest records (wave index: delay)")
prev_last = None
for e in range(32):
    row = d[e] - t0
    first, last = row.min(), row.max()
    late = np.argsort(row)[-5:][::-1]
    gap = "" if prev_last is None else " | since previous exchange complete: first %+.2f last %+.2f" % (first - prev_last, last - prev_last)
    print("J=%2d %s: first %7.2f | spread %5.2f | median %5.2f | %s%s" % (e // 2, "price" if e % 2 == 0 else "ratio", first, last - first,
          np.median(row) - first, " ".join("%d:%.2f" % (k, row[k] - first) for k in late), gap))
    prev_last = last
