"""Stress runs behind DESIGN.md's parity section (not part of the pytest suite: minutes of GPU).

  python tools/stress_handoff.py small REPS     the round-1 one-off: small LPs with entries over
                                                320 orders of magnitude (27 x 27 and neighbours),
                                                solved REPS times each, every trace vs the oracle;
                                                run it under MI355X_POISON_ALLOC=1 as well
  python tools/stress_handoff.py mid REPS       mid-size LPs on 3..17 look-ahead workgroups next to
                                                a stream that saturates HBM, one XCD and spread
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def trace_of(h, k):
    ec = np.empty(max(k, 1), dtype=np.int64); cr = np.empty(max(k, 1), dtype=np.int64); n = ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_trace(h, ptr(ec), ptr(cr), k, ctypes.byref(n)), "trace")
    return np.stack([ec[:k], cr[:k]], axis=1)


def extreme(n, m, seed, lo, hi):
    rng = np.random.default_rng(seed)
    mag = lambda shape: rng.uniform(0.5, 2.0, shape) * 10.0 ** rng.integers(lo, hi + 1, shape)   # noqa: E731
    M0 = np.zeros((m + 1, n + m + 1))
    M0[:m, :n] = mag((m, n)) * rng.choice([1.0, 1.0, -1.0], (m, n))
    M0[np.arange(m), n + np.arange(m)] = 1.0
    M0[:m, -1] = mag(m)
    M0[m, :n] = -mag(n)
    return M0, np.arange(n, n + m, dtype=np.int64)


def run_case(M0, b0, cap, reps, load=None):
    M, b = M0.copy(), b0.copy()
    with np.errstate(all="ignore"):
        st_o, npiv, trace = oracle.solve(M, b, max_pivots=cap, trace_cap=cap, omp=M0.size > 1 << 20)
    R, C = M0.shape
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create(ctypes.byref(h), R, C, ptr(M0), ptr(b0), 0), "create")
    bad = 0
    for rep in range(reps):
        if load:
            load()
        lp.capi.check(L.mi355x_tab_upload(h, ptr(M0), ptr(b0)), "upload")
        k = ctypes.c_int64(0)
        rc = L.mi355x_tab_solve(h, 1, 1024.0, cap, ctypes.byref(k))
        got = trace_of(h, npiv)
        if (rc, k.value) != (st_o, npiv) or not np.array_equal(got, trace):
            bad += 1
            d = np.where((got != trace).any(axis=1))[0]
            print("  MISMATCH rep %d: rc %d/%d pivots %d/%d first differing %s" % (rep, rc, st_o, k.value, npiv, d[:4]), flush=True)
    L.mi355x_tab_destroy(h)
    return bad


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "small"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    total_bad = 0
    if mode == "small":
        cases = [(27, 27, s, -160, 160) for s in range(8)] + [(27, 26, 3, -300, 20), (31, 27, 5, -160, 300),
                                                              (60, 40, 9, -160, 160), (2, 1, 1, -20, 20)]
        for la in (0, 1):
            L.mi355x_tune_set_lookahead_mode(la)
            for (n, m, s, lo, hi) in cases:
                M0, b0 = extreme(n, m, s, lo, hi)
                bad = run_case(M0, b0, 60, reps)
                total_bad += bad
                print("small la_mode=%d %dx%d seed %d [%d,%d]: %d reps, %d mismatches" % (la, n, m, s, lo, hi, reps, bad), flush=True)
        L.mi355x_tune_set_lookahead_mode(0)
    else:
        import torch
        stream = torch.cuda.Stream()
        src = torch.empty(1 << 28, dtype=torch.float64, device="cuda"); dst = torch.empty_like(src)

        def load():
            with torch.cuda.stream(stream):
                for _ in range(3):
                    dst.copy_(src, non_blocking=True)
        for one_xcd in (1, 0):
            L.mi355x_tune_set_la_one_xcd(one_xcd)
            L.mi355x_tune_set_lookahead_mode(2)
            for (n, m) in [(700, 333), (1500, 700), (2600, 2300), (8192, 4096)]:
                M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(2, n))
                bad = run_case(M0, b0, 256, reps if n < 8000 else max(reps // 20, 2), load)
                total_bad += bad
                print("mid one_xcd=%d %dx%d: %d mismatches" % (one_xcd, n, m, bad), flush=True)
        L.mi355x_tune_set_la_one_xcd(1)
        L.mi355x_tune_set_lookahead_mode(0)
        torch.cuda.synchronize()
    print("stress %s done in %.0f s: %d mismatches in total" % (mode, time.time() - t0, total_bad))
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
