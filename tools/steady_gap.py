"""Where does the time between the kernels of a config-3 block go?  (Round-4 review: the driver's
box showed 16.5 us per pivot in steady state where the kernels add up to 12.9.)

Runs the steady-state leg of bench.py (1 600 pivots on a warm config-3 tableau) and prints, per
run: how long the host needed to ENQUEUE the request (mi355x_tab_solve_async returning), how long
the request took in total, the event-timed kernel averages and the gap per block = (total -
sum of kernel time) / blocks.  If enqueue ~= total the run is bound by the host's launch rate.
With --load N, N busy-loop processes per host core keep the host cores occupied meanwhile.

    python tools/steady_gap.py [--load 2] [--pivots 1600] [--events 0|1] [--repeat 3]
"""
import argparse
import ctypes
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tests.helpers import lp_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--load", type=float, default=0, help="busy-loop processes per host core (may be a fraction)")
ap.add_argument("--pivots", type=int, default=1600)
ap.add_argument("--events", type=int, default=1, help="event pairs around every 4th block (as bench.py)")
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--block", type=int, default=0, help="pivots per sweep (0 = by size)")
ap.add_argument("--tr", type=int, default=0, help="rows per sweep workgroup (0 = default)")
ap.add_argument("--nt", type=int, default=-1, help="non-temporal sweep accesses (0 / 1, -1 = by size)")
ap.add_argument("--wait", type=int, default=2, help="status read-back: 2 published control block + memory poll (default), 1 polled stream query, 0 hipStreamSynchronize")
ap.add_argument("--xmap", type=int, default=-1, help="k_sweepw_ring: workgroups -> tiles by XCD (1) or in grid order (0); -1 = the library's default")
ap.add_argument("--skew", type=int, default=-1, help="k_sweepw_ring, one round of workgroups: rows by which the thirds of the tiles differ (0 = equal tiles); -1 = the library's default")
ap.add_argument("--prime", type=int, default=1, help="0: without the empty priming blocks in front of a handle's first request (profiling runs)")
ap.add_argument("--one-xcd", type=int, default=1, help="persistent look-ahead: all workgroups on one XCD (1, default) or wherever they land (0)")
ap.add_argument("--ring", type=int, default=1, help="wide sweeps through the LDS ring (1, default) or the register form (0)")
args = ap.parse_args()

lp = lp_amd()
L = lp.capi.lib()
n, m = args.n, args.m
L.mi355x_tune_set_sweepw_ring(args.ring)
L.mi355x_tune_set_ctl_wait(args.wait)
L.mi355x_tune_set_prime(args.prime)
L.mi355x_tune_set_la_one_xcd(args.one_xcd)
if args.xmap >= 0:
    L.mi355x_tune_set_sweep_xcd_map(args.xmap)
if args.skew >= 0:
    L.mi355x_tune_set_sweep_skew(args.skew)
if args.block:
    L.mi355x_tune_set_block(args.block)
if args.tr or args.nt >= 0:
    L.mi355x_tune_set_sweep_shape(args.tr, args.nt)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print("%s: %s (a quota throttles the whole container once the burners have used it up: waits then end "
              "on 100 ms period boundaries)" % (f, open(f).read().strip()), flush=True)

burners = []
if args.load:
    ncpu = os.cpu_count() or 1
    for _ in range(int(args.load * ncpu)):
        burners.append(subprocess.Popen(["sh", "-c", "while :; do :; done"], preexec_fn=os.setsid))
    time.sleep(1.0)

try:
    for rep in range(args.repeat):
        h = ctypes.c_void_p()
        k = ctypes.c_int64(0)
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3, 500 + rep), 0, -1, 0), "create")
        lp.capi.check(L.mi355x_tab_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "set_stream")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()                       # (their first use is not the run's business)
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "warm")
        L.mi355x_tab_sync(h, ctypes.byref(k))
        if args.events:
            L.mi355x_tab_timing_enable(h, 4)
        bk = L.mi355x_tab_block_size(h)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, args.pivots, 0), "run")
        e1.record()
        t1 = time.perf_counter()
        rc = L.mi355x_tab_sync(h, ctypes.byref(k))
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        gpu_us = e0.elapsed_time(e1) * 1e3
        blocks = -(-args.pivots // bk)
        ev = {}
        for kind, name in ((1, "la"), (0, "sweep")):
            nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
            L.mi355x_tab_timing_read_kind(h, kind, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
            ev[name] = sm.value / nl.value * 1e3 if nl.value else float("nan")
        tot_us = (t2 - t0) * 1e6
        kern = (ev["la"] + ev["sweep"]) * blocks
        print("load=%g wait=%d events=%d block=%d: enqueue %.2f ms; WALL %.2f ms = %.2f us/pivot (%.0f pivots/s); GPU CLOCK %.2f ms = "
              "%.2f us/pivot (%.0f pivots/s); host wait after the GPU %.2f ms; kernels la %.1f + sweep %.1f us per block; "
              "gap_us_per_block (GPU clock) %.1f; lost=%d rc=%d"
              % (args.load, args.wait, args.events, bk, (t1 - t0) * 1e3, tot_us / 1e3, tot_us / args.pivots,
                 args.pivots / (t2 - t0), gpu_us / 1e3, gpu_us / args.pivots, args.pivots / (gpu_us * 1e-6),
                 (tot_us - gpu_us) / 1e3, ev["la"], ev["sweep"], (gpu_us - kern) / blocks,
                 L.mi355x_tab_la_lost(h), rc), flush=True)
        L.mi355x_tab_destroy(h)
finally:
    for p in burners:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
