"""A/B of the resident batch solve with the whole strip in registers (two workgroups per CU) against
24 of its 64 columns in LDS (three per CU) in ONE run on one box: config-4 batches of 128 / 1 024 LPs
to optimality, results compared bit for bit.  python tools/resident_lds_ab.py"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, importlib
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
def batch(nl, reps=3, keep=False):
    n, m = 512, 256
    seeds = np.array([lp.synth.seed_for(4, i) for i in range(nl)], dtype=np.uint64)
    best, res = 0, None
    for rep in range(reps):
        b = lp.TableauBatch.synthetic(nl, n, m, seeds)
        lp.capi.check(L.mi355x_batch_prepare(b._h), "prepare")
        L.mi355x_batch_timing_enable(b._h, 1)
        t0 = time.perf_counter()
        st, npv = b.solve()
        dt = time.perf_counter() - t0
        nlch, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        L.mi355x_batch_timing_read(b._h, ctypes.byref(nlch), ctypes.byref(sm), ctypes.byref(mn))
        print("      %d LPs rep %d: %.2f ms wall, %d launch(es) %.2f ms on the device, %d pivots, statuses %s"
              % (nl, rep, dt * 1e3, nlch.value, sm.value, npv.sum(), np.unique(st).tolist()))
        best = max(best, npv.sum() / (sm.value * 1e-3))           # (device time: the wall clock also waits for the generator)
        if keep and rep == 0:
            res = (st.copy(), npv.copy(), [b.download(k)[0].copy() for k in range(0, nl, 7)])
    return best, res
ref = {}
for rnd in range(2):
    for mode in (0, 1):
        L.mi355x_tune_set_resident_lds(mode)
        out = []
        for nl in (128, 1024):
            v, res = batch(nl, keep=(rnd == 0))
            out.append(v)
            if res is not None:
                if nl in ref:
                    same = np.array_equal(ref[nl][0], res[0]) and np.array_equal(ref[nl][1], res[1]) and \
                        all(np.array_equal(x.view(np.int64), y.view(np.int64)) for x, y in zip(ref[nl][2], res[2]))
                    print("   %d LPs: results identical to mode 0: %s" % (nl, same))
                else:
                    ref[nl] = res
        print("strip mode %2d: batch128 %.2f M pivots/s   batch1024 %.2f M pivots/s" % (mode, out[0] / 1e6, out[1] / 1e6), flush=True)
L.mi355x_tune_set_resident_lds(0)
