"""Where a look-ahead step of the per-LP batch kernel (k_batch_block) spends its time: instrumented
build (-DMI355X_LA_TIMING), workgroup of LP 0, thread 0.   python tools/batch_timing.py [n_lps]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build
out = os.path.join(ROOT, "gpurun_out", "libmi355x_simplex_la_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
_build.build(extra_flags=["-DMI355X_LA_TIMING"], out=out)
os.environ["MI355X_SIMPLEX_LIB"] = out
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, m = 512, 256
seeds = np.array([lp.synth.seed_for(4, k) for k in range(nl)], dtype=np.uint64)
batch = lp.TableauBatch.synthetic(nl, n, m, seeds)
L.mi355x_batch_prepare(batch._h)
buf = np.zeros(8)
# the batch handle wraps a tab handle as its first member
tab = ctypes.cast(batch._h, ctypes.POINTER(ctypes.c_void_p))[0]
L.mi355x_debug_rhs(ctypes.c_void_p(tab), buf.ctypes.data_as(ctypes.c_void_p), 8, 1)
t0 = time.perf_counter(); st, npv = batch.solve(); dt = time.perf_counter() - t0
L.mi355x_debug_rhs(ctypes.c_void_p(tab), buf.ctypes.data_as(ctypes.c_void_p), 8, 0)
c = buf[0]
print("%d LPs: %.2f M pivots/s, LP 0: %d pivots (max %d), solve %.2f ms" % (nl, npv.sum() / dt / 1e6, npv[0], npv.max(), dt * 1e3))
print("us per step of LP 0 (thread 0): pricing loop %.2f | price reduce %.2f | column+chain %.2f | ratio reduce %.2f | row+chain+books %.2f | sum %.2f"
      % tuple([buf[k] / c * 0.01 for k in range(1, 6)] + [buf[1:6].sum() / c * 0.01]))
