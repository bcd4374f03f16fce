"""Where k_batch_block spends its time (LP 0 of a 128-LP config-4 batch): needs the library built
with -DMI355X_BB_TIMING.  python tools/batch_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
L.mi355x_debug_batch_part.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
n, m, nl = 512, 256, 128
seeds = np.array([lp.synth.seed_for(4, k) for k in range(nl)], dtype=np.uint64)
for rep in range(2):
    batch = lp.TableauBatch.synthetic(nl, n, m, seeds, device=0)
    L.mi355x_batch_prepare(batch._h)
    out = np.zeros(8)
    L.mi355x_debug_batch_part(batch._h, out.ctypes.data_as(ctypes.c_void_p), 8, 1)
    st, npv = batch.solve()
    L.mi355x_debug_batch_part(batch._h, out.ctypes.data_as(ctypes.c_void_p), 8, 0)
    del batch
if out[4] == 0:
    sys.exit("no samples: the library was built without -DMI355X_BB_TIMING")
print("LP 0: %d look-ahead steps, %d sweeps (%.1f pivots each)" % (out[4], out[6], out[7] / out[6]))
print("us per step: pricing %.2f | column + chain + ratio %.2f | row + chain %.2f | RHS + bookkeeping %.2f"
      % tuple(out[k] / out[4] * 0.01 for k in range(4)))
print("us per sweep: %.1f" % (out[5] / out[6] * 0.01))
