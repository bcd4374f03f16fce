"""Where the persistent block launch of a column shard (k_shard_la_block) spends a step: an instrumented
copy of the library (-DMI355X_LA_TIMING: the leader thread accumulates wall_clock64 deltas per phase in the
shard's otherwise unused rhs buffer), one 8-GPU-sized shard of config 5 (32769 x 8193 stored).
    python tools/shard_la_timing.py [hop (1|0)] [n_vars n_cons]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build
out = os.path.join(ROOT, "gpurun_out", "libmi355x_simplex_la_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(_build.CSRC, f)) for f in os.listdir(_build.CSRC)):
    _build.build(extra_flags=["-DMI355X_LA_TIMING"], out=out)
os.environ["MI355X_SIMPLEX_LIB"] = out
os.environ["MI355X_COLPART_FORCE_RCCL"] = "1"
import importlib
import numpy as np
lp = importlib.import_module("linear-programming_amd")
cp = importlib.import_module("linear-programming_amd.colpart")
L = lp.capi.lib()
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (65536 // 8, 32768)
L.mi355x_tune_set_block(24)
L.mi355x_tune_set_colpart_exchange(2)
L.mi355x_tune_set_shard_self_hop(hop)
tab = cp.NativeColumnPartition.synthetic(n, m, lp.synth.seed_for(5), 1)
tab.solve_async(336, reset=True); tab.sync()
NS = 24 * 8
buf = np.zeros(NS)
L.mi355x_colpart_debug_rhs(tab._h, 0, buf.ctypes.data_as(__import__("ctypes").c_void_p), NS, 1)
tab.solve_async(24 * 60); tab.sync()
L.mi355x_colpart_debug_rhs(tab._h, 0, buf.ctypes.data_as(__import__("ctypes").c_void_p), NS, 0)
d = buf.reshape(24, 8)
if d[:, 0].sum() == 0:
    sys.exit("no samples: built without -DMI355X_LA_TIMING, or the persistent block launch did not run")
print("hop=%d  %d vars x %d constraints (one shard)   us per step (leader thread), %s" % (hop, n, m, tab.la_stats()))
print(" J     n | price-xchg  ->all-waves  pairs(A)  column+chain  ratio-xchg  row+chain+bk |  total")
tot = 0.0
for J in range(24):
    c = d[J, 0]
    if c == 0:
        continue
    row = [d[J, k] / c * 0.01 for k in (1, 2, 3, 4, 5, 6)]
    tot += sum(row)
    print("%2d %5d | " % (J, int(c)) + " ".join("%10.2f" % x for x in row) + " | %6.2f" % sum(row))
print("sum over the steps of a block: %.1f us" % tot)
