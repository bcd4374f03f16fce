"""What do the col operands cost k_sweep16?  The sweep of one pending list launched back to back
(mi355x_debug_repeat_sweep), with the product library and with a copy built with
-DMI355X_SWEEP_FAKE_COL (no scalar col loads at all, every col value +0.0: the same 16 x (mul, sub)
per element, the same rows streamed).
    python tools/sweep_fake_col.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from tests.helpers import lp_amd
    lp = lp_amd(); L = lp.capi.lib()
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), 8192, 4096, lp.synth.seed_for(3), 0, -1, 0), "create")
    npv = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 32, 1), "two blocks")
    L.mi355x_tab_sync(h, ctypes.byref(npv))
    us = ctypes.c_double(0)
    for tr in (0, 16, 64):                      # rows per workgroup: 0 = the launcher's choice (32)
        L.mi355x_tune_set_sweep_shape(tr, -1)
        best = 1e9
        for rep in range(3):
            lp.capi.check(L.mi355x_debug_repeat_sweep(h, 50, ctypes.byref(us)), "repeat")
            best = min(best, us.value)
        print("%s: rows per workgroup %2d: sweep back to back, best of 3 x 50 launches: %.1f us" % (sys.argv[2], tr or 32, best), flush=True)
    sys.exit(0)
import build as _build
fake = os.path.join(ROOT, "gpurun_out", "libmi355x_simplex_fake_col.so")
os.makedirs(os.path.dirname(fake), exist_ok=True)
_build.build(extra_flags=["-DMI355X_SWEEP_FAKE_COL"], out=fake, force=True)
for name, lib in (("product library", None), ("no col loads     ", fake)):
    env = dict(os.environ)
    if lib:
        env["MI355X_SIMPLEX_LIB"] = lib
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", name], env=env, check=False)
