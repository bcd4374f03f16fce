"""Tuning sweep (not part of the product): time every compiled k_update variant on a
synthetic tableau with per-launch HIP events, print GB/s (algorithmic bytes 2*R*C*8)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import lp_amd  # noqa: E402

lp = lp_amd()
L = lp.capi.lib()


def run(n, m, pivots, variants):
    R, C = m + 1, n + m + 1
    bytes_per = 2 * R * C * 8
    out = []
    for v in variants:
        L.mi355x_tune_set_variant(v)
        name = L.mi355x_tune_variant_name(v).decode()
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3), 0, -1, 0), "create")
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 10, 1), "warm")
        npv = ctypes.c_int64(0)
        L.mi355x_tab_sync(h, ctypes.byref(npv))
        # wall clock for the whole iteration
        t0 = time.perf_counter()
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, pivots, 0), "run")
        rc = L.mi355x_tab_sync(h, ctypes.byref(npv))
        wall = time.perf_counter() - t0
        # per-launch events
        L.mi355x_tab_timing_enable(h, 1)
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, pivots, 0), "run")
        nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        L.mi355x_tab_timing_read(h, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
        L.mi355x_tab_timing_enable(h, 0)
        avg = sm.value / max(nl.value, 1)
        c, cols, ld = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
        L.mi355x_tab_layout(h, ctypes.byref(c), ctypes.byref(cols), ctypes.byref(ld))
        bytes_per = 2 * R * cols.value * 8
        rec = {"variant": v, "name": name, "compact": c.value, "stored_cols": cols.value, "n": n, "m": m, "rc": rc, "pivots_per_s": pivots / wall,
               "wall_us_per_pivot": wall / pivots * 1e6, "update_avg_us": avg * 1e3,
               "update_min_us": mn.value * 1e3, "update_GBps": bytes_per / (avg * 1e-3) / 1e9,
               "update_GBps_best": bytes_per / (mn.value * 1e-3) / 1e9}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        L.mi355x_tab_destroy(h)
    return out


if __name__ == "__main__":
    nv = L.mi355x_tune_variant_count()
    sizes = [(8192, 4096)] if len(sys.argv) < 2 else [tuple(map(int, a.split("x"))) for a in sys.argv[1:]]
    for n, m in sizes:
        run(n, m, 100 if n * m < 10**8 else 24, range(nv))
