"""Is the first request after a big free late at its START or at its END?"""
import ctypes, os, sys, time, threading
sys.path.insert(0, os.getcwd())
import torch
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
def one(tag):
    h = ctypes.c_void_p(); k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), 8192, 4096, lp.synth.seed_for(3, 500), 0, -1, 0), "c")
    lp.capi.check(L.mi355x_tab_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "s")
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "w"); L.mi355x_tab_sync(h, ctypes.byref(k))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    seen = {}
    def watch():
        while "e0" not in seen:
            if "rec" in seen and e0.query(): seen["e0"] = time.perf_counter()
    th = threading.Thread(target=watch); th.start()
    t0 = time.perf_counter()
    e0.record(); seen["rec"] = 1
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 4200, 0), "r")
    e1.record()
    t1 = time.perf_counter()
    L.mi355x_tab_sync(h, ctypes.byref(k))
    t2 = time.perf_counter()
    th.join()
    torch.cuda.synchronize()
    print("%-28s enqueue %.2f ms, e0 seen complete %.2f ms after t0, return %.2f ms, GPU clock %.2f ms" % (tag, (t1-t0)*1e3, (seen["e0"]-t0)*1e3, (t2-t0)*1e3, e0.elapsed_time(e1)), flush=True)
    L.mi355x_tab_destroy(h)
one("fresh process")
one("again")
x = torch.empty(20 * 1024**3, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
one("after freeing 20 GB (torch)")
one("again")
hb = ctypes.c_void_p(); k = ctypes.c_int64(0)
lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(hb), 65536, 32768, lp.synth.seed_for(5), 0, -1, 0), "big")
lp.capi.check(L.mi355x_tab_solve_async(hb, 1, 1024.0, 56, 1), "w"); L.mi355x_tab_sync(hb, ctypes.byref(k))
L.mi355x_tab_destroy(hb)
one("after a config-5 handle")
one("again")
