"""How long does the persistent look-ahead (k_la_block) take while another stream keeps HBM busy?
(The question behind overlapping the look-ahead of block B+1 with the sweep of block B.)
    python tools/la_under_load.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
n, m = 8192, 4096


def run(load):
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
    npv = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    L.mi355x_tab_timing_enable(h, 1)
    side = torch.cuda.Stream()
    if load:
        a = torch.empty(64 << 20, dtype=torch.float64, device="cuda")      # 512 MB
        b = torch.empty_like(a)
        with torch.cuda.stream(side):
            for _ in range(load):
                b.copy_(a, non_blocking=True)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 16 * 40, 0), "run")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    busy = load and not side.query()
    torch.cuda.synchronize()
    out = []
    for kind in (1, 0):
        nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        L.mi355x_tab_timing_read_kind(h, kind, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
        out.append("%s avg %6.1f min %6.1f us (%d)" % ("look-ahead" if kind else "sweep", sm.value / max(nl.value, 1) * 1e3, mn.value * 1e3, nl.value))
    print("copy loop on a second stream: %-5s (still running at the end: %s) | %s | lost=%d" % (bool(load), busy, " | ".join(out), L.mi355x_tab_la_lost(h)), flush=True)
    L.mi355x_tab_destroy(h)


run(0)
run(400)
