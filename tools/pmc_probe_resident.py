"""HBM-traffic probe of the resident solve for rocprofv3 --pmc runs: a device-to-device copy of the
config-2 tableau (known bytes: calibrates FETCH_SIZE / WRITE_SIZE), then the whole solve of config 2
as ONE k_resident launch -- whose HBM traffic should be the tableau once each way plus the exchange
buffers, however many pivots it makes -- and a 128-LP config-4 batch (one launch as well)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = 1024, 512
h = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(2), 0, -1, 0), "create")
h2 = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_copy(ctypes.byref(h2), h), "copy")      # reads + writes rows*ld*8 bytes (dense, padded)
npv = ctypes.c_int64(0)
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 0, 1), "prepare")
L.mi355x_tab_sync(h, ctypes.byref(npv))
print("resident:", L.mi355x_tab_resident(h))
rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(npv))
print("cfg2 rc", rc, "pivots", npv.value)
c, cols, ld = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
L.mi355x_tab_layout(h, ctypes.byref(c), ctypes.byref(cols), ctypes.byref(ld))
print("layout compact=%d stored_cols=%d stored_ld=%d rows=%d dense_ld=%d" % (c.value, cols.value, ld.value, m + 1, (n + m + 1 + 15) // 16 * 16))
seeds = np.array([lp.synth.seed_for(4, i) for i in range(128)], dtype=np.uint64)
b = lp.TableauBatch.synthetic(128, 512, 256, seeds)
lp.capi.check(L.mi355x_batch_prepare(b._h), "prepare")
st, pv = b.solve()
print("cfg4 batch: pivots", int(pv.sum()), "all optimal", bool((st == 0).all()))
