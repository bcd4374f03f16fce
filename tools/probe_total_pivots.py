import ctypes, sys, time
sys.path.insert(0, "/root/repo")
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
for cfg, (n, m) in {3: (8192, 4096), 2: (1024, 512)}.items():
    for rank in range(2):
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(cfg, rank), 0, -1, 0), "create")
        npv = ctypes.c_int64(0)
        t0 = time.perf_counter()
        rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(npv))
        print("cfg", cfg, "rank", rank, "rc", rc, "pivots", npv.value, "seconds", time.perf_counter() - t0, flush=True)
        L.mi355x_tab_destroy(h)
