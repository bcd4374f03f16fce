for r in 1 0; do for p in 20 24 44; do echo "== ring=$r pivots=$p"; python tools/steady_gap.py --repeat 4 --pivots $p --events 0 --ring $r 2>&1 | grep load= | cut -c1-230; done; done
echo "== warm 5 like bench"; python - <<'PY'
import ctypes, time, sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
for rep in range(4):
    h = ctypes.c_void_p(); k = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), 8192, 4096, lp.synth.seed_for(3, rep), 0, -1, 0), "c")
    lp.capi.check(L.mi355x_tab_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "s")
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 5, 1), "w"); L.mi355x_tab_sync(h, ctypes.byref(k))
    torch.cuda.synchronize()
    cnt0 = (ctypes.c_int64 * 8)(); L.mi355x_tab_path_counts(h, cnt0)
    t0 = time.perf_counter()
    lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 20, 0), "r")
    t1 = time.perf_counter()
    L.mi355x_tab_sync(h, ctypes.byref(k))
    t2 = time.perf_counter()
    cnt = (ctypes.c_int64 * 8)(); L.mi355x_tab_path_counts(h, cnt)
    print("rep", rep, "enqueue %.1f us total %.1f us" % ((t1-t0)*1e6, (t2-t0)*1e6), "paths", [cnt[i]-cnt0[i] for i in range(8)], flush=True)
    L.mi355x_tab_destroy(h)
PY
