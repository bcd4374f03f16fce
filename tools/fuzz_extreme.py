"""Fuzz of the NaN / inf / subnormal corner: random small LPs whose entries span hundreds of orders
of magnitude (the construction of tests/test_gpu_property.py::test_extreme_magnitudes_bitwise),
many more seeds than the test runs, on several code paths; every pivot trace, status and final
tableau against the oracle.
    python tools/fuzz_extreme.py [cases] [first_seed] [extreme|ordinary]
`ordinary`: the generator of test_random_lps_bitwise (sparse / dense / integer-degenerate data, max
and min problems, up to 700 x 400)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind_arg = sys.argv[3] if len(sys.argv) > 3 else "extreme"
# (name, look-ahead mode, block, compact, select mode, resident mode); the default path is the
# resident solve at these sizes, "persistent" is the blocked path with the resident solve off
MODES = [("resident", 0, 16, 1, 0, 0), ("persistent", 0, 16, 1, 0, 1), ("two-launch", 1, 16, 1, 0, 0),
         ("per-pivot", 0, 1, 1, 0, 0), ("dense-1wg", 0, 16, 0, 1, 0), ("dense-split", 0, 16, 0, 2, 0),
         ("persistent-24", 0, 24, 1, 2, 0), ("two-launch-24", 1, 24, 1, 2, 0), ("wide-28", 0, 28, 1, 2, 0)]


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def free_mb():
    """free device memory (a leak of handles' buffers shows up here long before an allocation fails)"""
    try:
        hip = None
        for name in (None, "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):   # (None: the runtime this process has loaded)
            try:
                hip = ctypes.CDLL(name)
                hip.hipMemGetInfo                            # (AttributeError: not in this one)
                break
            except (OSError, AttributeError):
                hip = None
        if hip is None:
            return float("nan")
        f, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(tot))
        return f.value / 1e6
    except OSError:
        return float("nan")


bad = 0
mem0 = None
t0 = time.time()
meta = np.random.default_rng(seed0)
for case in range(cases):
    if case % 2000 == 1999:
        if mem0 is None:
            mem0 = free_mb()
        else:
            print("   case %d: free device memory %+.0f MB since case 2000" % (case + 1, free_mb() - mem0), flush=True)
    seed = int(meta.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    is_max, cap, lo, hi = 1, 60, 0, 0
    if kind_arg == "extreme":
        n = int(meta.integers(2, 61)); m = int(meta.integers(1, 41))
        lo = int(meta.choice([-300, -160, -20])); hi = int(meta.choice([20, 160, 300]))
        mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(lo, hi + 1, shape)   # noqa: E731
        M0 = np.zeros((m + 1, n + m + 1))
        M0[:m, :n] = mag((m, n)) * rng.choice([1.0, 1.0, -1.0], (m, n))
        M0[np.arange(m), n + np.arange(m)] = 1.0
        M0[:m, -1] = mag(m)
        M0[m, :n] = -mag(n)
    else:
        n = int(meta.integers(1, 701)); m = int(meta.integers(1, 401)); cap = 300
        is_max = int(meta.integers(0, 2)); density = float(meta.choice([1.0, 0.5, 0.1]))
        if meta.integers(0, 2):
            A = rng.integers(0, 4, (m, n)).astype(np.float64); bb = rng.integers(0, 5, m).astype(np.float64)
            c = rng.integers(-2, 5, n).astype(np.float64)
        else:
            A = rng.uniform(-0.5, 1.5, (m, n)); A[rng.uniform(size=(m, n)) > density] = 0.0
            bb = rng.uniform(0.5, 5.0, m); c = rng.uniform(-0.5, 2.0, n)
        M0 = np.zeros((m + 1, n + m + 1))
        M0[:m, :n] = A
        M0[np.arange(m), n + np.arange(m)] = 1.0
        M0[:m, -1] = bb
        M0[m, :n] = -c if is_max else c
    b0 = np.arange(n, n + m, dtype=np.int64)
    M, b = M0.copy(), b0.copy()
    with np.errstate(all="ignore"):
        st_o, npiv, trace = oracle.solve(M, b, is_max=bool(is_max), max_pivots=cap, trace_cap=cap)
    name, la, blk, cmp_, sel, res = MODES[case % len(MODES)]
    L.mi355x_tune_set_lookahead_mode(la); L.mi355x_tune_set_block(blk); L.mi355x_tune_set_compact(cmp_); L.mi355x_tune_set_select_mode(sel)
    L.mi355x_tune_set_resident(res)
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create(ctypes.byref(h), m + 1, n + m + 1, ptr(M0), ptr(b0), 0), "create")
    k = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(h, is_max, 1024.0, cap, ctypes.byref(k))
    ec = np.empty(cap + 4, dtype=np.int64); cr = np.empty(cap + 4, dtype=np.int64); nn = ctypes.c_int64(0)
    L.mi355x_tab_trace(h, ptr(ec), ptr(cr), cap + 4, ctypes.byref(nn))
    got = np.stack([ec[:nn.value], cr[:nn.value]], axis=1)
    G = np.empty_like(M0); bg = np.empty_like(b0)
    lp.capi.check(L.mi355x_tab_download(h, ptr(G), ptr(bg), None, None), "download")
    L.mi355x_tab_destroy(h)
    nan_o, nan_g = np.isnan(M), np.isnan(G)
    ok = (rc, k.value) == (st_o, npiv) and np.array_equal(got, trace) and np.array_equal(nan_o, nan_g) and \
        np.array_equal(G[~nan_g].view(np.int64), M[~nan_o].view(np.int64)) and np.array_equal(bg, b)
    if not ok:
        bad += 1
        print("MISMATCH case %d mode %s: n=%d m=%d seed=%d lo=%d hi=%d: rc %d/%d pivots %d/%d trace %s vs %s" % (
            case, name, n, m, seed, lo, hi, rc, st_o, k.value, npiv, got.tolist()[-3:], trace.tolist()[-3:]), flush=True)
        if bad >= 10:
            break
L.mi355x_tune_set_lookahead_mode(0); L.mi355x_tune_set_block(0); L.mi355x_tune_set_compact(1); L.mi355x_tune_set_select_mode(0)
L.mi355x_tune_set_resident(0)
print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
