"""HBM-traffic probe for rocprofv3 --pmc runs: a device-to-device copy of the tableau (known
bytes: calibrates FETCH_SIZE / WRITE_SIZE), then N pivots of config 3."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = 8192, 4096
if len(sys.argv) > 5:                                     # another shape (an 8-GPU-sized shard of config 5: 8192 32768)
    n, m = int(sys.argv[4]), int(sys.argv[5])
if len(sys.argv) > 2:                                     # pivots per sweep (blocked pivoting; 0 = by size)
    L.mi355x_tune_set_block(int(sys.argv[2]))
if len(sys.argv) > 3:                                     # sweep implementation (mi355x_tune_set_sweep_impl)
    L.mi355x_tune_set_sweep_impl(int(sys.argv[3]))
L.mi355x_tune_set_prime(0)                                # (the empty priming blocks would count as launches of the profiled kernels)
if os.environ.get("PROBE_XMAP"):                          # k_sweepw_ring: workgroups -> tiles by XCD
    L.mi355x_tune_set_sweep_xcd_map(int(os.environ["PROBE_XMAP"]))
if os.environ.get("PROBE_RING"):
    L.mi355x_tune_set_sweepw_ring(int(os.environ["PROBE_RING"]))
h = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3 if m == 4096 else 5), 0, -1, 0), "create")
h2 = ctypes.c_void_p()
lp.capi.check(L.mi355x_tab_copy(ctypes.byref(h2), h), "copy")      # reads + writes rows*ld*8 bytes
lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, int(sys.argv[1]) if len(sys.argv) > 1 else 30, 1), "run")
npv = ctypes.c_int64(0)
print("rc", L.mi355x_tab_sync(h, ctypes.byref(npv)), "pivots", npv.value)
c, cols, ld = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
L.mi355x_tab_layout(h, ctypes.byref(c), ctypes.byref(cols), ctypes.byref(ld))
print("layout compact=%d stored_cols=%d stored_ld=%d rows=%d dense_ld=%d block=%d" % (c.value, cols.value, ld.value, m + 1, (n + m + 1 + 15) // 16 * 16, L.mi355x_tab_block_size(h)))
