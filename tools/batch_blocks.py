"""Steady-state throughput of the per-LP batch kernels (config 4 shape) for the block sizes of
k_batch_block, three repetitions each in one process (the first launch of a kernel pays one-time
costs).  python tools/batch_blocks.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = 512, 256
for nl in (128, 1024):
    seeds = np.array([lp.synth.seed_for(4, k) for k in range(nl)], dtype=np.uint64)
    for bb in (1, 4, 8, 16, 0, 8, 16):
        L.mi355x_tune_set_batch_block(bb)
        res = []
        for rep in range(3):
            batch = lp.TableauBatch.synthetic(nl, n, m, seeds, device=0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st, npv = batch.solve()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res.append(npv.sum() / dt / 1e6)
            del batch
        print("lps %4d block %2d: %s M pivots/s" % (nl, bb, " ".join("%.2f" % r for r in res)), flush=True)
