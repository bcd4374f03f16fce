"""Feasibility of overlapping the look-ahead of block b+1 with the sweep of block b (round-3 review,
item 5's "new lever"): how long does the persistent look-ahead (k_la_block, config 3) take NEXT TO a
stream of sweeps when the two are kept apart by CU masks -- the look-ahead alone on one XCD (its
workgroups share that XCD's L2 anyway), the sweeps on the other seven?  Round 2 measured 222 us
instead of 121 us per block without masks, which made the overlap worthless.
hipExtStreamCreateWithCUMask: bit i of the mask = CU i; both plausible numberings of the 256 CUs are
tried (XCD-major: CUs 0..31 = XCD 0; interleaved: CU i on XCD i % 8).
    python tools/la_cu_mask.py"""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # (HIP context; the library shares torch's runtime)
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
n, m = 8192, 4096
torch.zeros(1, device="cuda")


def masked_stream(cus):
    words = (ctypes.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s


def handle(stream):
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
    k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm"); L.mi355x_tab_sync(h, ctypes.byref(k))
    if stream is not None:
        lp.capi.check(L.mi355x_tab_set_stream(h, stream, 0), "set_stream")
    return h


def experiment(name, la_cus, sweep_cus, load):
    s1 = masked_stream(la_cus) if la_cus is not None else None
    s2 = masked_stream(sweep_cus) if sweep_cus is not None else None
    h1, h2 = handle(s1), handle(s2)
    k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h1, 1, 1024.0, 32, 0), "warm on the stream"); L.mi355x_tab_sync(h1, ctypes.byref(k))
    L.mi355x_tab_timing_enable(h1, 1)
    sweep_us = ctypes.c_double(0)
    th = None
    if load:
        th = threading.Thread(target=lambda: L.mi355x_debug_repeat_sweep(h2, 1500, ctypes.byref(sweep_us)))
        th.start()
        time.sleep(0.03)
    lp.capi.check(L.mi355x_tab_solve_async(h1, 1, 1024.0, 16 * 40, 0), "run")
    L.mi355x_tab_sync(h1, ctypes.byref(k))
    still = th.is_alive() if th else False
    if th:
        th.join()
    out = []
    for kind in (1, 0):
        nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        L.mi355x_tab_timing_read_kind(h1, kind, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
        out.append("%s avg %6.1f min %6.1f us" % ("look-ahead" if kind else "own sweep", sm.value / max(nl.value, 1) * 1e3, mn.value * 1e3))
    print("%-58s | %s | neighbour's sweeps %6.1f us each (running throughout: %s) | lost=%d"
          % (name, " | ".join(out), sweep_us.value, still, L.mi355x_tab_la_lost(h1)), flush=True)
    L.mi355x_tab_destroy(h1); L.mi355x_tab_destroy(h2)


ALL = list(range(256))
for conv, xcd0 in (("XCD-major", list(range(32))), ("interleaved", list(range(0, 256, 8)))):
    rest = [c for c in ALL if c not in xcd0]
    experiment("look-ahead alone, unmasked", None, None, False)
    experiment("look-ahead alone on 'XCD 0' (%s)" % conv, xcd0, None, False)
    experiment("look-ahead unmasked next to unmasked sweeps", None, None, True)
    experiment("look-ahead on 'XCD 0' next to sweeps on the rest (%s)" % conv, xcd0, rest, True)
    experiment("look-ahead unmasked next to sweeps on the rest (%s)" % conv, None, rest, True)
