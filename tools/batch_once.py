"""One config-4 batch solve (for rocprofv3 --kernel-trace --stats).  python tools/batch_once.py [n_lps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, m = 512, 256
seeds = np.array([lp.synth.seed_for(4, k) for k in range(nl)], dtype=np.uint64)
for rep in range(3):
    batch = lp.TableauBatch.synthetic(nl, n, m, seeds)
    L.mi355x_batch_prepare(batch._h)
    t0 = time.perf_counter(); st, npv = batch.solve(); dt = time.perf_counter() - t0
    print("%d LPs: %.2f M pivots/s, %d pivots, max %d per LP, %.2f ms" % (nl, npv.sum() / dt / 1e6, npv.sum(), npv.max(), dt * 1e3))
