import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = 512, 256
for nl in (128, 1024):
    seeds = np.array([lp.synth.seed_for(4, k) for k in range(nl)], dtype=np.uint64)
    for impl in (0, 1):
        for tr in (0, 8, 16, 64):
            L.mi355x_tune_set_sweep_impl(impl); L.mi355x_tune_set_sweep_shape(tr, -1)
            res = []
            for rep in range(3):
                batch = lp.TableauBatch.synthetic(nl, n, m, seeds)
                L.mi355x_batch_prepare(batch._h)
                t0 = time.perf_counter(); st, npv = batch.solve(); dt = time.perf_counter() - t0
                res.append(npv.sum() / dt / 1e6)
                del batch
            print("lps %4d sweep impl %d tr %2d: %s M pivots/s" % (nl, impl, tr, " ".join("%.2f" % r for r in res)), flush=True)
