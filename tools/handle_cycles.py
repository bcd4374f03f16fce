"""Device memory over create / solve / trace / download / destroy cycles, per tuning mode and with random
shapes (the set-up that exposed the trace read-back leak of round 4).  python tools/handle_cycles.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, importlib, torch
lp = importlib.import_module("linear-programming_amd")
L = lp.capi.lib()
def ptr(a): return a.ctypes.data_as(ctypes.c_void_p)
rng = np.random.default_rng(1)
MODES = [("resident", 0, 16, 1, 0, 0), ("persistent", 0, 16, 1, 0, 1), ("two-launch", 1, 16, 1, 0, 0),
         ("per-pivot", 0, 1, 1, 0, 0), ("dense-1wg", 0, 16, 0, 1, 0), ("dense-split", 0, 16, 0, 2, 0),
         ("persistent-24", 0, 24, 1, 2, 0), ("two-launch-24", 1, 24, 1, 2, 0), ("wide-28", 0, 28, 1, 2, 0)]
for name, la, blk, cmp_, sel, res in MODES:
    L.mi355x_tune_set_lookahead_mode(la); L.mi355x_tune_set_block(blk); L.mi355x_tune_set_compact(cmp_); L.mi355x_tune_set_select_mode(sel)
    L.mi355x_tune_set_resident(res)
    torch.cuda.synchronize(); f0 = torch.cuda.mem_get_info()[0]
    CYC = 1200
    for it in range(CYC):
        n = int(rng.integers(1, 701)); m = int(rng.integers(1, 401))
        M0 = np.zeros((m + 1, n + m + 1)); M0[:m, :n] = rng.uniform(0.1, 1.0, (m, n)); M0[np.arange(m), n + np.arange(m)] = 1.0
        M0[:m, -1] = rng.uniform(1, 5, m); M0[m, :n] = -rng.uniform(0.5, 2.0, n)
        b0 = np.arange(n, n + m, dtype=np.int64)
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create(ctypes.byref(h), m + 1, n + m + 1, ptr(M0), ptr(b0), 0), "create")
        k = ctypes.c_int64(0)
        L.mi355x_tab_solve(h, 1, 1024.0, 300, ctypes.byref(k))
        ec = np.empty(304, dtype=np.int64); cr = np.empty(304, dtype=np.int64); nn = ctypes.c_int64(0)
        L.mi355x_tab_trace(h, ptr(ec), ptr(cr), 304, ctypes.byref(nn))
        G = np.empty_like(M0); bg = np.empty_like(b0)
        lp.capi.check(L.mi355x_tab_download(h, ptr(G), ptr(bg), None, None), "download")
        L.mi355x_tab_destroy(h)
    torch.cuda.synchronize()
    print("%-14s: free memory %+.1f MB after %d cycles (%.3f MB per cycle)" % (name, (torch.cuda.mem_get_info()[0] - f0) / 1e6, CYC, (f0 - torch.cuda.mem_get_info()[0]) / 1e6 / CYC), flush=True)
L.mi355x_tune_set_lookahead_mode(0); L.mi355x_tune_set_block(0); L.mi355x_tune_set_compact(1); L.mi355x_tune_set_select_mode(0); L.mi355x_tune_set_resident(0)
