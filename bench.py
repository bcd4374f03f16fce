#!/usr/bin/env python
"""bench.py -- simplex pivots/sec + achieved HBM GB/s on the dense 8192 x 4096 f64 tableau.

    python bench.py --gpus N --steps K --warmup W

One "step" is one full simplex iteration of the hot path on the GPU: pricing arg-min over the
reduced costs -> ratio-test arg-min -> Gauss-Jordan rank-1 update of the whole tableau
(src/simplex.lisp:453-461), on BASELINE.json config 3: a dense random LP with 8192 variables and
4096 <=-constraints, i.e. a 4097 x 12289 double-float tableau (402.8 MB), generated directly in
HBM (inputs resident before the timed region starts).

N = 1: one tableau on one GPU.  N > 1 (launched by torch.distributed.run, one rank per GPU):
every rank iterates on its own independent LP of the same shape (different seed) -- the
"independent LPs shard trivially" partition of the north star, no data-path collective -- and
the value is the whole-job aggregate (sum of pivots over ranks / max-over-ranks time), weak
scaling.  `--workload colpart` instead runs ONE tableau column-partitioned over the ranks with
the per-pivot RCCL exchange (all-gather of the local pricing winners + broadcast of the entering
column); see linear-programming_amd/colpart.py.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

# numpy's OpenBLAS wakes one worker per core for a dot product and lets them SPIN for ~100 ms afterwards
# (THREAD_TIMEOUT): 64 spinning threads use up a container's CPU quota (cgroup cpu.max, 16 cores on the
# gpurun boxes) and the whole process is frozen until the next 100 ms period -- measured as a read-back
# 26 - 72 ms late on the request that followed the one dot product of the config-5 leg (DESIGN_experiments.md
# R5.10).  Nothing here needs a threaded BLAS.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (n_vars, n_constraints, config id used for the seed)
    "cfg3": (8192, 4096, 3),
    "cfg2": (1024, 512, 2),
    "cfg5": (65536, 32768, 5),      # the config-5 tableau UNPARTITIONED on one GPU (25.8 GB dense)
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3200)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS) + ["cfg4", "colpart"])
    ap.add_argument("--batch-lps", type=int, default=128,
                    help="cfg4: LPs per GPU (BASELINE config 4 = 1024 LPs over 8 GPUs)")
    ap.add_argument("--alternate-sweep", action="store_true",
                    help="tuning: consecutive update launches sweep the tableau in opposite directions")
    ap.add_argument("--ld-extra", type=int, default=0,
                    help="tuning: extra padding doubles per tableau row (multiple of 16)")
    ap.add_argument("--force-dense", action="store_true",
                    help="tuning: run the solve loop on the dense tableau (no compact representation)")
    ap.add_argument("--batch-mode", type=int, default=0,
                    help="cfg4: 0 auto, 1 lockstep launch pairs, 2 one workgroup per LP")
    ap.add_argument("--colpart-dense", action="store_true",
                    help="colpart: shards hold all logical columns instead of the non-basic ones only")
    ap.add_argument("--colpart-block", type=int, default=0,
                    help="colpart: pivots per sweep of the shards (0 = 16, 1 = per-pivot updates)")
    ap.add_argument("--colpart-vars", type=int, default=0,
                    help="colpart: override the number of variables (constraints = vars/2)")
    ap.add_argument("--block", type=int, default=0,
                    help="tuning: pivots selected ahead and applied per sweep (0 = library default, 1 = off)")
    ap.add_argument("--batch-block", type=int, default=0,
                    help="tuning, cfg4: pivots per pass of the blocked per-LP kernel (0 = default, 1 = off)")
    ap.add_argument("--no-prime", action="store_true",
                    help="profiling runs: without the EMPTY blocks a handle's first request is preceded by (one launch of "
                         "every kernel form its requests can pick) -- they would count as launches of the profiled kernels")
    ap.add_argument("--sweepw-ring", type=int, default=1,
                    help="tuning: wide sweeps through the per-wave LDS ring (1, default) or the register form of round 4 (0)")
    ap.add_argument("--sweep-tr", type=int, default=0, help="tuning: rows per sweep workgroup")
    ap.add_argument("--sweep-nt", type=int, default=-1, help="tuning: non-temporal sweep accesses (0/1)")
    ap.add_argument("--multi-gpu", default="colpart", choices=["colpart", "independent"],
                    help="N > 1, default workload: headline = ONE config-5 tableau column-partitioned over the "
                         "N GPUs with the RCCL exchanges (strong scaling; the independent-LPs figure is "
                         "attached as a secondary field), or independent config-3 LPs only (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pivots", type=int, default=400,
                    help="cpu_baseline: pivots timed with all host threads (a quarter of it single-threaded)")
    ap.add_argument("--no-events", action="store_true",
                    help="do not bracket the update launches with HIP events (roofline = null)")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="bracket every k-th block of the timed region with HIP event pairs (0 = every "
                         "block of a short run, ~50 samples of a long one)")
    ap.add_argument("--no-per-pivot", action="store_true",
                    help="skip the extra per-pivot (k_update) measurement after the timed region")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, cfg3: skip the `other_configs` records (configs 2, 4, 5 and the config-3 "
                         "steady state measured after the timed region)")
    ap.add_argument("--no-colpart-baseline", action="store_true",
                    help="colpart, N > 1: skip the same-workload one-GPU figure measured on rank 0 first")
    ap.add_argument("--no-colpart-ab", action="store_true",
                    help="colpart over RCCL: skip the second leg with the rooted-broadcast exchange")
    return ap.parse_args()


F64_VALU_PEAK_TFLOPS = 39.3     # non-FMA f64 vector rate: 256 CUs x 64 lanes x 2.4 GHz x 1 flop
                                # (the product and the difference are rounded separately: no FMA)


def cpu_baseline(lp, n, m, seed, pivots):
    """The oracle (C restatement of the reference algorithm; the Lisp reference itself cannot
    run here) timed on the host cores on the first `pivots` pivots of the same LP.  Also returns
    what the in-run parity check compares the GPU with: the pivot trace, the RHS column and the
    objective row after those pivots."""
    import oracle
    import numpy as np
    M, b = lp.synth.tableau(n, m, seed)
    # OpenMP sees every host thread (128 on the GPU box) whatever the container's CPU quota is (16 cores there):
    # run with what the quota really grants and report THAT as `cores` (round-5 review)
    threads = oracle.set_omp_threads(min(oracle.omp_threads(), oracle.usable_cores()))
    # the host is shared: one sample swings with whatever else runs on the box (53 ... 235 pivots/s
    # seen for this leg), so the pivots are timed in four consecutive segments (a dense pivot costs
    # the same whichever it is) and the BEST segment is the figure; all of them are listed
    segs, traces, npiv, st = [], [], 0, None
    per = max(1, pivots // 4)
    while npiv < pivots and st in (None, oracle.MAX_PIVOTS):
        k = min(per, pivots - npiv)
        t0 = time.perf_counter()
        st, done, tr = oracle.solve(M, b, max_pivots=k, trace_cap=k, omp=True)
        segs.append(done / (time.perf_counter() - t0))
        traces.append(tr)
        npiv += done
    trace = np.concatenate(traces) if traces else np.zeros((0, 2), dtype=np.int64)
    state = {"status": st, "pivots": npiv, "trace": trace, "rhs": M[:, -1].copy(), "obj": M[m].copy(),
             "basis": b.copy()}
    single = max(2, pivots // 8)
    ones = []
    for _ in range(2):
        t0 = time.perf_counter()
        st1, npiv1, _ = oracle.solve(M, b, max_pivots=single, omp=False)
        ones.append(npiv1 / (time.perf_counter() - t0))
    best = max(segs)
    return {
        "value": best, "unit": "pivots/s", "cores": threads, "kind": "port",
        "host_threads_visible": os.cpu_count(), "cores_by": "min(sched_getaffinity, cgroup cpu.max quota)",
        "sample": "first %d pivots of the same %dx%d LP in %d segments, best segment (all: %s), OpenMP "
                  "row-parallel C restatement of src/simplex.lisp:337-461 (SBCL unavailable in the image); "
                  "single-thread: %.3f pivots/s (best of two runs of %d further pivots)"
                  % (npiv, n, m, len(segs), ", ".join("%.1f" % x for x in segs), max(ones), single),
        "segments": segs, "single_thread_value": max(ones),
        "GBps": 2.0 * (m + 1) * (n + m + 1) * 8 * best / 1e9,
    }, state


def parity_in_run(lp, L, n, m, seed, device, state):
    """the GPU's first K pivots of the benchmark LP against the oracle's --
    pivot sequence, RHS column, objective row and basis, bit for bit (a fresh handle, the default
    solve path, outside the timed region)."""
    import numpy as np
    K = int(state["pivots"])
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, device), "parity handle")
    k = ctypes.c_int64(0)
    rc = L.mi355x_tab_solve(h, 1, 1024.0, K, ctypes.byref(k))
    ec = np.empty(max(K, 1), dtype=np.int64); cr = np.empty(max(K, 1), dtype=np.int64); cnt = ctypes.c_int64(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)                                   # noqa: E731
    lp.capi.check(L.mi355x_tab_trace(h, vp(ec), vp(cr), K, ctypes.byref(cnt)), "trace")
    last_row = np.empty(n + m + 1); last_col = np.empty(m + 1); basis = np.empty(m, dtype=np.int64)
    lp.capi.check(L.mi355x_tab_download(h, None, vp(basis), vp(last_row), vp(last_col)), "download")
    L.mi355x_tab_destroy(h)
    got = np.stack([ec[:K], cr[:K]], axis=1)
    out = {"pivots": K, "status_equal": int(rc) == int(state["status"]) and int(k.value) == K,
           "trace_identical": bool(np.array_equal(got, state["trace"])),
           "rhs_column_bitwise": bool(np.array_equal(last_col.view(np.int64), state["rhs"].view(np.int64))),
           "objective_row_bitwise": bool(np.array_equal(last_row.view(np.int64), state["obj"].view(np.int64))),
           "basis_identical": bool(np.array_equal(basis, state["basis"])),
           "objective_value": float(last_col[-1]),
           "checked_against": "oracle (C restatement of src/simplex.lisp:337-461), same LP, first %d pivots" % K}
    out["identical"] = all(out[x] for x in ("status_equal", "trace_identical", "rhs_column_bitwise",
                                            "objective_row_bitwise", "basis_identical"))
    return out


def per_pivot_record(lp, L, n, m, seed, device, kernel_bytes, restore_block, pivots=192):
    """The north star's own target -- >= 60 % of the HBM roofline on the 8192 x 4096 f64 PIVOT
    kernel -- measured on the per-pivot path (`--block 1`: k_update, the literal n-pivot-row, one
    launch per pivot) in the same run, outside the timed region."""
    L.mi355x_tune_set_block(1)
    try:
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, device), "per-pivot handle")
        npv = ctypes.c_int64(0)
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 32, 1), "warm")
        L.mi355x_tab_sync(h, ctypes.byref(npv))
        L.mi355x_tab_timing_enable(h, 1)
        t0 = time.perf_counter()
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, pivots, 0), "per-pivot run")
        L.mi355x_tab_sync(h, ctypes.byref(npv))
        dt = time.perf_counter() - t0
        nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        L.mi355x_tab_timing_read(h, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
        L.mi355x_tab_destroy(h)
    finally:
        L.mi355x_tune_set_block(restore_block)
    avg_ms = sm.value / max(nl.value, 1)
    ach = kernel_bytes / (avg_ms * 1e-3) / 1e9
    traffic, src = pmc_traffic("cfg3", "k_update")
    return {"kernel": L.mi355x_update_kernel_name().decode(), "what": "per-pivot path (block = 1): one k_update launch per pivot",
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": src,
            "note": "read frac as an upper figure, MALL-assisted: the compact tableau is about the size of the 256 MiB "
                    "Infinity Cache and FETCH_SIZE counts L2 fetches whoever answers them (DESIGN.md 4.1)",
            "kernel_avg_us": avg_ms * 1e3, "kernel_min_us": mn.value * 1e3, "launches_timed": int(nl.value),
            "bytes_moved_per_launch": kernel_bytes, "pivots_per_launch": 1,
            "pivots_per_s_whole_iteration": pivots / dt}


def _events(L, h, kind):
    nl, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    L.mi355x_tab_timing_read_kind(h, kind, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
    return {"launches_timed": int(nl.value), "avg_us": sm.value / nl.value * 1e3 if nl.value else None,
            "min_us": mn.value * 1e3 if nl.value else None}


def other_configs(lp, L, device, block):
    """The other BASELINE configurations, measured in the same run right AFTER the config-3 timed
    region (never part of `value`), each with pivots/s, the per-kernel HIP-event averages and a
    parity flag where the oracle can follow: config 2 (full solve), config 4 (128-LP batch, solved
    to optimality), config 5 as ONE column shard (64 pivots), and config 3 in steady state
    (~4 200 pivots of full blocks, by the wall clock and by the GPU's clock -- the driver's 20 timed
    steps are ONE block of 20)."""
    import numpy as np
    import torch
    import oracle
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)                                   # noqa: E731
    out = {}

    # ---- config 2: 1024 x 512, solved to optimality, bitwise against the oracle
    n, m = 1024, 512
    seed = lp.synth.seed_for(2)
    hw = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(hw), n, m, lp.synth.seed_for(2, 1), 0, -1, device), "cfg2 warm")
    k = ctypes.c_int64(0)
    L.mi355x_tab_solve(hw, 1, 1024.0, 0, ctypes.byref(k))                     # warm: same kernels, another LP
    L.mi355x_tab_destroy(hw)
    # three timed solves on fresh handles of the same LP: the median is reported, all three are listed
    # (a solve that follows host-side work with large device allocations has been seen to run slow once)
    runs2 = []
    for rep_i in range(3):
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, device), "cfg2")
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 0, 1), "cfg2 prepare")   # representation change outside the timing
        L.mi355x_tab_sync(h, ctypes.byref(k))
        resident = bool(L.mi355x_tab_resident(h))
        L.mi355x_tab_timing_enable(h, 1 if resident else 4)   # (blocked path: every fourth block -- the event records are host work inside the timing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.mi355x_tab_solve(h, 1, 1024.0, 0, ctypes.byref(k))
        runs2.append(time.perf_counter() - t0)
        if rep_i < 2:
            L.mi355x_tab_destroy(h)
    dt = sorted(runs2)[1]
    M, b = lp.synth.tableau(n, m, seed)
    so, no, trace = oracle.solve(M, b, trace_cap=1 << 14)
    ec = np.empty(max(no, 1), dtype=np.int64); cr = np.empty(max(no, 1), dtype=np.int64); cnt = ctypes.c_int64(0)
    L.mi355x_tab_trace(h, vp(ec), vp(cr), no, ctypes.byref(cnt))
    last_row = np.empty(n + m + 1); last_col = np.empty(m + 1); basis = np.empty(m, dtype=np.int64)
    L.mi355x_tab_download(h, None, vp(basis), vp(last_row), vp(last_col))
    ident = (int(rc) == so and int(k.value) == no and np.array_equal(np.stack([ec[:no], cr[:no]], axis=1), trace)
             and np.array_equal(last_col.view(np.int64), M[:, -1].view(np.int64))
             and np.array_equal(last_row.view(np.int64), M[m].view(np.int64)) and np.array_equal(basis, b))
    out["cfg2_full_solve"] = {
        "workload": "BASELINE config 2: dense random LP 1024 vars x 512 <=-constraints (513x1537 f64), solved to optimality",
        "value": k.value / dt, "unit": "pivots/s", "pivots": int(k.value), "ms": dt * 1e3, "us_per_pivot": dt / max(k.value, 1) * 1e6,
        "ms_of_the_three_runs": [x * 1e3 for x in runs2], "reported": "median of three solves on fresh handles",
        "path": "resident: the stored 513x1025 tableau in registers (32 workgroups x 256 threads x 64 doubles), one "
                "exchange per pivot, no HBM traffic inside the loop" if resident else "blocked: look-ahead + sweep",
        "kernels": ({"k_resident_whole_solve": _events(L, h, 0)} if resident else
                    {"lookahead_per_block_of_%d" % block: _events(L, h, 1), "sweep_per_block": _events(L, h, 0)}),
        "parity": {"identical": bool(ident), "checked_against": "oracle, whole solve: status, pivot count, pivot "
                   "sequence, RHS column, objective row, basis (bit for bit)"}}
    L.mi355x_tab_destroy(h)

    # ---- config 4: 128 independent 512 x 256 LPs on this GPU (1024 over 8), solved to optimality
    n, m, nl = 512, 256, 128
    seeds = np.array([lp.synth.seed_for(4, i) for i in range(nl)], dtype=np.uint64)
    warm = lp.TableauBatch.synthetic(nl, n, m, seeds[::-1].copy(), device=device)
    warm.solve()
    del warm
    runs4 = []
    for rep_i in range(3):
        batch = lp.TableauBatch.synthetic(nl, n, m, seeds, device=device)
        lp.capi.check(L.mi355x_batch_prepare(batch._h), "mi355x_batch_prepare")
        L.mi355x_batch_timing_enable(batch._h, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, npv = batch.solve()
        runs4.append(time.perf_counter() - t0)
        if rep_i < 2:
            del batch
    dt = sorted(runs4)[1]
    nlch, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    L.mi355x_batch_timing_read(batch._h, ctypes.byref(nlch), ctypes.byref(sm), ctypes.byref(mn))
    checked, same = 16, True
    for i in range(checked):
        Mi, bi = lp.synth.tableau(n, m, int(seeds[i]))
        so, no, _ = oracle.solve(Mi, bi)
        Gi, gb = batch.download(i)
        same = same and int(st[i]) == so and int(npv[i]) == no and np.array_equal(Gi.view(np.int64), Mi.view(np.int64)) \
            and np.array_equal(gb, bi)
    out["cfg4_128_lp_batch"] = {
        "workload": "BASELINE config 4, one GPU's share: 128 independent LPs of 512 vars x 256 <=-constraints "
                    "(257x769 f64 each), every LP solved to optimality",
        "value": float(npv.sum()) / dt, "unit": "pivots/s (aggregate)", "pivots_total": int(npv.sum()), "ms": dt * 1e3,
        "ms_of_the_three_runs": [x * 1e3 for x in runs4], "reported": "median of three solves of fresh batches",
        "pivots_per_lp_min_mean_max": [int(npv.min()), float(npv.mean()), int(npv.max())],
        "all_optimal": bool((st == 0).all()),
        "path": "resident: every LP in registers (8 workgroups per LP, 64 LPs in flight), one exchange per pivot"
                if L.mi355x_tab_resident is not None and nlch.value <= 2 else "blocked: look-ahead per LP + sweeps over all LPs",
        "kernels": {"update_or_resident_launches": {"launches_timed": int(nlch.value),
                                                    "avg_us": sm.value / nlch.value * 1e3 if nlch.value else None,
                                                    "min_us": mn.value * 1e3 if nlch.value else None}},
        "parity": {"identical": bool(same), "checked_against": "oracle, LPs 0..%d of the batch: status, pivot count, "
                   "every entry of the final tableau, basis (bit for bit)" % (checked - 1)}}
    del batch
    torch.cuda.empty_cache()

    # ---- config 4 as a whole on ONE GPU: all 1024 LPs (the first 128 are the batch above)
    nl = 1024
    seeds = np.array([lp.synth.seed_for(4, i) for i in range(nl)], dtype=np.uint64)
    warm = lp.TableauBatch.synthetic(nl, n, m, seeds[::-1].copy(), device=device)
    warm.solve()
    del warm
    runs = []                                  # three timed solves on fresh batches; the median is reported, all are listed
    for rep in range(3):
        batch = lp.TableauBatch.synthetic(nl, n, m, seeds, device=device)
        lp.capi.check(L.mi355x_batch_prepare(batch._h), "mi355x_batch_prepare")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, npv = batch.solve()
        runs.append(time.perf_counter() - t0)
        if rep < 2:
            del batch
    dt = sorted(runs)[1]
    checked, same = (1023, 517, 300, 128), True
    for i in checked:
        Mi, bi = lp.synth.tableau(n, m, int(seeds[i]))
        so, no, _ = oracle.solve(Mi, bi)
        Gi, gb = batch.download(i)
        same = same and int(st[i]) == so and int(npv[i]) == no and np.array_equal(Gi.view(np.int64), Mi.view(np.int64)) \
            and np.array_equal(gb, bi)
    out["cfg4_1024_lp_batch"] = {
        "workload": "BASELINE config 4 as a whole on ONE GPU: 1024 independent LPs of 512 vars x 256 <=-constraints, "
                    "every LP solved to optimality",
        "value": float(npv.sum()) / dt, "unit": "pivots/s (aggregate)", "pivots_total": int(npv.sum()), "ms": dt * 1e3,
        "ms_of_the_three_runs": [x * 1e3 for x in runs], "reported": "median of three solves of fresh batches",
        "all_optimal": bool((st == 0).all()),
        "parity": {"identical": bool(same), "checked_against": "oracle, LPs %s of the batch: status, pivot count, every "
                   "entry of the final tableau, basis (bit for bit)" % (list(checked),)}}
    del batch
    torch.cuda.empty_cache()

    # ---- config 5 as ONE column shard on this GPU (the denominator of the 8-GPU claim)
    try:
        cp = importlib.import_module("linear-programming_amd.colpart")
        base = cp.one_shard_baseline(65536, 32768, lp.synth.seed_for(5), device, "blocks", 0)
        base["workload"] = ("BASELINE config 5 on ONE GPU: the 32769x98305 f64 tableau (25.8 GB dense, 17.2 GB stored) "
                            "as one column shard through mi355x_colpart_*, four full blocks (%d pivots per sweep) after "
                            "one warm-up block" % base["pivots_per_sweep"])
        base["parity"] = {"identical": None, "checked_against": "nothing in this run: no CPU oracle follows 3.2e9 entries "
                          "in bench time (tests/test_gpu_fullsize.py re-derives 64 pivots of this tableau in numpy)"}
        out["cfg5_one_shard"] = base
    except BaseException as e:                       # noqa: BLE001 -- a record is owed whatever happens here
        out["cfg5_one_shard"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- config 5 SOLVED TO OPTIMALITY on this GPU (~135 000 pivots of the reference's loop at 17 GB)
    if os.environ.get("BENCH_SKIP_CFG5_FULL") != "1":
        try:
            n5, m5 = 65536, 32768
            h5 = ctypes.c_void_p()
            lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h5), n5, m5, lp.synth.seed_for(5), 0, -1, device), "cfg5 full")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc5 = L.mi355x_tab_solve(h5, 1, 1024.0, 0, ctypes.byref(k))
            dt5 = time.perf_counter() - t0
            row5, col5, bas5 = np.empty(n5 + m5 + 1), np.empty(m5 + 1), np.empty(m5, dtype=np.int64)
            lp.capi.check(L.mi355x_tab_download(h5, None, vp(bas5), vp(row5), vp(col5)), "cfg5 read-back")
            x5 = np.zeros(n5 + m5)
            x5[bas5] = col5[:m5]
            c5 = 0.5 + lp.synth.splitmix_u01(lp.synth.seed_for(5), n5 * m5 + m5, n5)
            out["cfg5_full_solve_one_gpu"] = {
                "workload": "BASELINE config 5 (65536 vars x 32768 constraints, 25.8 GB dense / 17.2 GB stored) solved to "
                            "optimality as one tableau on ONE GPU",
                "status": int(rc5), "optimal": int(rc5) == 0, "pivots": int(k.value), "wall_s": dt5,
                "value": k.value / dt5, "unit": "pivots/s", "pivots_per_sweep": int(L.mi355x_tab_block_size(h5)),
                "properties": {"min_reduced_cost": float(row5[:n5 + m5].min()), "dual_feasible": bool(row5[:n5 + m5].min() >= -128 * 1.1102230246251568e-16),
                               "min_rhs": float(col5[:m5].min()), "primal_feasible": bool(col5[:m5].min() >= 0.0),
                               "objective": float(col5[m5]),
                               "ctx_recomputed_rel_err": float(abs(float((c5 * x5[:n5]).sum()) - col5[m5]) / abs(col5[m5]))},
                "parity": {"identical": None, "checked_against": "size-independent properties here (tests/test_gpu_optimum.py adds "
                           "Ax <= b with A regenerated); the first 64 pivots bit for bit against the oracle in "
                           "tests/test_gpu_fullsize.py"}}
            L.mi355x_tab_destroy(h5)
            torch.cuda.empty_cache()
        except BaseException as e:                   # noqa: BLE001
            out["cfg5_full_solve_one_gpu"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- config 3, steady state: full blocks only, long enough that the one host wake-up at the end
    # (the thread sleeps on the completion interrupt; on a busy host it is back milliseconds late --
    # the 60.7 k against 77 k of the round-4 driver run was ONE such wake-up on a 21 ms run) is small
    # against the run, and measured by the GPU's own clock next to the wall clock
    # Three LPs of the shape, the same request each; the record is the run with the shortest WALL time, all three
    # are listed with where the host's time went (mi355x_debug_last_wait).  (Until the threaded BLAS was switched
    # off at the top of this file the FIRST of the three returned 26 - 72 ms late in every run: the config-5 leg's
    # one dot product left 63 OpenBLAS workers spinning, the container's CPU quota ran out and the process was
    # frozen until the next period -- the GPU's own clock showed the usual 47.6 ms.  Round 4's single
    # steady-state run, 60.7 k against 77 k, was that request.)
    runs = []
    for rep in range(3):
        n, m = 8192, 4096
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, lp.synth.seed_for(3, 500 + rep), 0, -1, device), "cfg3 steady")
        lp.capi.check(L.mi355x_tab_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "set_stream")
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, 64, 1), "cfg3 steady warm")
        L.mi355x_tab_sync(h, ctypes.byref(k))
        bk = max(L.mi355x_tab_block_size(h), 1)
        blocks = 4200 // bk
        pivots = blocks * bk
        L.mi355x_tab_timing_enable(h, 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, pivots, 0), "cfg3 steady run")
        e1.record()
        t_enq = time.perf_counter() - t0
        rc = L.mi355x_tab_sync(h, ctypes.byref(k))
        dt = time.perf_counter() - t0
        lw = (ctypes.c_double * 2)()
        L.mi355x_debug_last_wait(lw)
        torch.cuda.synchronize()
        gpu_ms = e0.elapsed_time(e1)
        la_ev, sw_ev = _events(L, h, 1), _events(L, h, 0)
        kern_us = (la_ev["avg_us"] or 0.0) + (sw_ev["avg_us"] or 0.0)
        run = {
            "workload": "BASELINE config 3 over %d pivots (%d full blocks of %d) after 64 warm-up pivots" % (pivots, blocks, bk),
            "value": pivots / dt, "unit": "pivots/s", "ms": dt * 1e3, "us_per_pivot": dt / pivots * 1e6,
            "gpu_clock": {"what": "the same run between two events on the launch stream: first launch to last kernel's end, "
                                  "without the host's wake-up after it",
                          "ms": gpu_ms, "pivots_per_s": pivots / (gpu_ms * 1e-3), "us_per_pivot": gpu_ms * 1e3 / pivots},
            "host_enqueue_ms": t_enq * 1e3,
            "host_wait_after_gpu_ms": dt * 1e3 - gpu_ms,
            "host_read_back_ms": {"launching_k_ctl_publish": lw[0] * 1e-3, "polling_its_sequence_number": lw[1] * 1e-3},
            "gap_us_per_block": gpu_ms * 1e3 / blocks - kern_us if kern_us else None,
            "still_running": int(rc) == lp.capi.MI_RUNNING and k.value == 64 + pivots,
            "kernels": {"lookahead_per_block_of_%d" % bk: la_ev, "sweep_per_block": sw_ev}}
        L.mi355x_tab_destroy(h)
        runs.append(run)
    best = min(runs, key=lambda r: r["ms"])
    best["all_runs"] = [{"wall_ms": r["ms"], "gpu_clock_ms": r["gpu_clock"]["ms"], "host_enqueue_ms": r["host_enqueue_ms"],
                         "host_wait_after_gpu_ms": r["host_wait_after_gpu_ms"], "host_read_back_ms": r["host_read_back_ms"]} for r in runs]
    best["what"] = "the shortest of three runs by the wall clock (three LPs of the shape); all three in all_runs"
    out["cfg3_steady_state"] = best
    torch.cuda.empty_cache()
    return out


def baseline_metric():
    """The metric string exactly as BASELINE.json spells it."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "simplex pivots/sec + achieved HBM GB/s, dense 8192\u00d74096 f64 tableau"


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH doubled per the gfx950 correction, calibrated on a copy of
    the same buffer -- profiles/rNN_cfg3_pmc_traffic.json).  PMC counters cannot be collected
    from inside this process, so the number is the last profiled one for this kernel, or None."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc_traffic.json" % (tag, workload))
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        recs = d.get("kernels", {})
        if kernel in recs:                                   # (exact name first: k_sweepw is not k_sweep)
            return recs[kernel]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        for name, rec in recs.items():
            if name.startswith(kernel + " ") or name.startswith(kernel + "<"):
                return rec["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        if "kernels" not in d and (d.get("kernel", "").startswith(kernel) or kernel.startswith(d.get("kernel", "?"))):
            return d["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def rocprof_avg_us(workload, kernel_prefix):
    """Average launch duration of the kernel whose name starts with `kernel_prefix` in the last committed
    rocprofv3 --kernel-trace --stats summary of this workload (profiles/rNN_<workload>_kernel_stats.csv), next to
    the live HIP-event figure: an event pair brackets the kernel plus the queue's hand-over on either side
    (2-4 us here), rocprofv3 reads the kernel's own begin / end timestamps."""
    import csv
    for tag in ("r06", "r05", "r04"):
        path = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, workload))
        try:
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Name"].replace("void ", "").replace("mi355x::", "")
                    if name.startswith(kernel_prefix):
                        return float(row["AverageNs"]) / 1e3, int(row["Calls"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None, None


def bench_batch(args, lp, rank, local_rank, N, barrier, torch, dist):
    """BASELINE config 4: a batch of independent 512 x 256 LPs per GPU (1024 over 8 GPUs),
    solved to optimality; value = total pivots of all LPs / time.  No collective."""
    import numpy as np
    n, m = 512, 256
    nl = args.batch_lps
    seeds = np.array([lp.synth.seed_for(4, rank * nl + k) for k in range(nl)], dtype=np.uint64)
    L = lp.capi.lib()
    L.mi355x_tune_set_batch_mode(args.batch_mode)
    if args.batch_block:
        L.mi355x_tune_set_batch_block(args.batch_block)
    # untimed warm-up on a throw-away batch of the same size (same kernels, same grid, clocks up:
    # the timed region is ONE launch of 20-40 ms)
    warm = lp.TableauBatch.synthetic(nl, n, m, seeds[::-1].copy(), device=local_rank)
    warm.solve()
    del warm
    batch = lp.TableauBatch.synthetic(nl, n, m, seeds, device=local_rank)
    lp.capi.check(L.mi355x_batch_prepare(batch._h), "mi355x_batch_prepare")   # no allocation inside the timed solve
    L.mi355x_batch_timing_enable(batch._h, 1)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, npv = batch.solve()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    nlch, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    L.mi355x_batch_timing_read(batch._h, ctypes.byref(nlch), ctypes.byref(sm), ctypes.byref(mn))
    assert (st == 0).all(), "not every LP reached optimality"
    tot = torch.tensor([float(npv.sum()), elapsed], dtype=torch.float64,
                       device="cuda" if dist.get_backend() == "nccl" else "cpu") if N > 1 else None
    if N > 1:
        piv = tot[:1].clone()
        dist.all_reduce(piv, op=dist.ReduceOp.SUM)
        tmax = tot[1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total_pivots, elapsed = float(piv.item()), float(tmax.item())
    else:
        total_pivots = float(npv.sum())
    R, C = m + 1, n + m + 1
    return {
        "metric": "simplex pivots/sec, batch of independent 512x256 f64 LPs",
        "value": total_pivots / elapsed, "unit": "pivots/s", "n_gpus": N,
        "steps": int(npv.max()), "warmup": args.warmup,
        "ms_per_step": elapsed / max(int(npv.max()), 1) * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: %d independent LPs per GPU, 512 vars x 256 "
                               "<=-constraints each (257x769 f64 tableau), solved to optimality"
                               % nl, "lps_total": nl * N, "pivots_total": total_pivots,
                   "pivots_per_lp_min_mean_max": [int(npv.min()), float(npv.mean()), int(npv.max())]},
        "aggregate_GBps": 2.0 * R * C * 8 * total_pivots / elapsed / 1e9,
        "update_launches": int(nlch.value), "update_ms_total": sm.value,
        # default path: the resident solve -- every LP in registers, the whole batch ONE launch that
        # reads every stored tableau once and writes it once; per pivot one exchange through L2
        "roofline": {"bound": "latency", "kernel": "k_resident" if nlch.value <= 2 else "k_sweep (batch)",
                     "achieved": 2.0 * R * (n + 1) * 8 * nl / (sm.value * 1e-3) / 1e9 if sm.value else None,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": 2.0 * R * (n + 1) * 8 * nl / (sm.value * 1e-3) / 1e9 / HBM_PEAK_GBPS if sm.value else None,
                     "traffic": pmc_traffic("resident", "k_resident cfg4")[0] if nl == 128 else None,
                     "traffic_source": pmc_traffic("resident", "k_resident cfg4")[1] if nl == 128 else None,
                     "what": "physical bytes of the launch(es) that move the stored tableaux (load + write-back) / their "
                             "duration; the loop itself makes no HBM traffic -- informational, the path is latency-bound",
                     "launches_timed": int(nlch.value),
                     "us_per_pivot_per_lp": (sm.value * 1e3 / max(float(npv.mean()), 1.0)) if sm.value else None},
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    N = args.gpus
    if world != N:
        if world == 1 and N > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (N, N))
        N = world

    import torch
    import torch.distributed as dist
    lp = importlib.import_module("linear-programming_amd")
    if not torch.cuda.is_available() or lp.capi.device_count() < 1:
        sys.exit("bench.py needs a GPU (no CPU fallback exists)")
    # Test hooks (used to exercise the multi-rank code path on a 1-GPU box, never by the driver):
    # BENCH_SHARE_DEVICE=1 puts every rank on cuda:0, BENCH_DIST_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device) for the few scalar reductions bench.py does.
    if os.environ.get("BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if N > 1:
            dist.barrier()

    if args.workload == "cfg4":
        rec = bench_batch(args, lp, rank, local_rank, N, barrier, torch, dist)
        if rank == 0:
            print(json.dumps(rec), flush=True)
        if N > 1:
            dist.destroy_process_group()
        return

    if args.workload == "colpart":
        from importlib import import_module
        colpart = import_module("linear-programming_amd.colpart")
        rec = colpart.bench(args, rank, local_rank, N)
        if rank == 0:
            print(json.dumps(rec), flush=True)
        if N > 1:
            dist.destroy_process_group()
        return

    n, m, cfg = WORKLOADS[args.workload]
    R, C = m + 1, n + m + 1
    bytes_per_pivot = 2 * R * C * 8            # dense tableau: every element read once + written once
    L = lp.capi.lib()
    if args.alternate_sweep:
        L.mi355x_tune_set_alternate_sweep(1)
    if args.force_dense:
        L.mi355x_tune_set_compact(0)
    if args.ld_extra:
        L.mi355x_tune_set_ld_extra(args.ld_extra)
    if args.block:
        L.mi355x_tune_set_block(args.block)
    if args.sweep_tr or args.sweep_nt >= 0:
        L.mi355x_tune_set_sweep_shape(args.sweep_tr, args.sweep_nt)
    if not args.sweepw_ring:
        L.mi355x_tune_set_sweepw_ring(0)
    if args.no_prime:
        L.mi355x_tune_set_prime(0)
    # One LP supports only so many pivots before it is optimal (config 3: 5 700-6 100, config 2:
    # 189-416 with these seeds).  If more timed steps are asked for than one LP safely provides,
    # further LPs of the same shape are generated in HBM BEFORE the timed region and the timed
    # steps simply continue on the next one.
    capacity = {"cfg3": 4500, "cfg2": 150, "cfg5": 20000}[args.workload]
    per_lp = max(capacity - args.warmup, 1)
    n_lps = -(-args.steps // per_lp)
    handles = []
    npv = ctypes.c_int64(0)
    for k in range(n_lps):
        h = ctypes.c_void_p()
        seed = lp.synth.seed_for(cfg, rank + 1000 * k)
        lp.capi.check(L.mi355x_tab_create_synthetic(ctypes.byref(h), n, m, seed, 0, -1, local_rank),
                      "mi355x_tab_create_synthetic")
        # every handle launches on torch's current stream, so chained LPs run back to back
        lp.capi.check(L.mi355x_tab_set_stream(
            h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "set_stream")
        # warm-up (untimed): also moves the handle onto the representation the loop runs on
        lp.capi.check(L.mi355x_tab_solve_async(h, 1, 1024.0, args.warmup, 1), "warmup")
        L.mi355x_tab_sync(h, ctypes.byref(npv))
        if not args.no_events:
            # every k-th launch of a long run (~50 event pairs per kernel class: the event records are
            # host work inside the timed region); a short run (the driver's 20 steps are one full
            # block + 4) records nothing inside the timed region -- its kernel statistics come from
            # the full blocks run right AFTER it (below)
            launches = max(1, args.steps // max(L.mi355x_tab_block_size(h), 1))
            stride = args.event_stride if args.event_stride > 0 else max(1, launches // 50)
            if launches >= 8 or args.event_stride > 0:
                L.mi355x_tab_timing_enable(h, stride)
        handles.append(h)
    seed = lp.synth.seed_for(cfg, rank)
    h = handles[0]

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    left, done = args.steps, 0
    for hk in handles:
        k = min(left, per_lp)
        lp.capi.check(L.mi355x_tab_solve_async(hk, 1, 1024.0, k, 0), "timed steps")
        left -= k
    for hk in handles:
        rc = L.mi355x_tab_sync(hk, ctypes.byref(npv))  # waits for the launch stream
        if rc != lp.capi.MI_RUNNING:
            sys.exit("rank %d: an LP terminated (status %d) inside the timed region" % (rank, rc))
        done += npv.value - args.warmup
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if done != args.steps:
        sys.exit("rank %d: %d of %d timed pivots were performed" % (rank, done, args.steps))

    compact, stored_cols, stored_ld = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0)
    L.mi355x_tab_layout(h, ctypes.byref(compact), ctypes.byref(stored_cols), ctypes.byref(stored_ld))
    # bytes the update kernel has to move per launch in the representation it runs on: every
    # STORED element read once + written once.  Dense: C = n+m+1 columns (SURVEY 8d's figure);
    # compact: only the n non-basic columns + RHS carry information.
    kernel_bytes = 2 * R * stored_cols.value * 8
    # blocked pivoting: one launch of the update kernel applies `block` pivots to every element
    # while it is in registers; the bytes above are then moved once per BLOCK
    block = L.mi355x_tab_block_size(h)
    resident = bool(L.mi355x_tab_resident(h))      # the stored tableau in registers: one launch per request
    if resident:
        block = 1

    def read_events(kind):
        tot_n, tot_ms, mn_ms = 0, 0.0, None
        for hk in handles:
            nlk, sm, mn = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
            L.mi355x_tab_timing_read_kind(hk, kind, ctypes.byref(nlk), ctypes.byref(sm), ctypes.byref(mn))
            tot_n += nlk.value
            tot_ms += sm.value
            if nlk.value:
                mn_ms = mn.value if mn_ms is None else min(mn_ms, mn.value)
        return tot_n, (tot_ms / tot_n if tot_n else None), mn_ms

    upd_n, upd_avg_ms, upd_min_ms = (0, None, None)
    la_n, la_avg_ms, la_min_ms = (0, None, None)
    extra_samples = 0
    if not args.no_events:
        upd_n, upd_avg_ms, upd_min_ms = read_events(0)
        la_n, la_avg_ms, la_min_ms = read_events(1)
        if upd_n < 8 and block > 1:
            # a short run (the driver's 20 steps are ONE block, cut short): the kernel statistics come from
            # WARM full blocks run right AFTER the timed region on the same tableau -- not part of `value`.
            # Two untimed blocks first (the first full blocks behind a short request run a kernel form the
            # request did not, on cold caches: round 5's line read 114 us where rocprofv3 and the steady
            # state read 106-108), then 12 timed ones, and ONLY those are the figure: the one event pair of
            # the timed region brackets a block cut short, a different launch.
            L.mi355x_tab_timing_enable(handles[-1], 0)
            lp.capi.check(L.mi355x_tab_solve_async(handles[-1], 1, 1024.0, 2 * block, 0), "warm blocks")
            L.mi355x_tab_sync(handles[-1], ctypes.byref(npv))
            for hk in handles:                                   # (drop what the timed region recorded)
                L.mi355x_tab_timing_enable(hk, 0)
            # (an event pair around every FOURTH block, as in a long run: a pair around every block puts three
            # marker packets between any two kernels and reads 3-4 % more than rocprofv3 does)
            L.mi355x_tab_timing_enable(handles[-1], 4)
            lp.capi.check(L.mi355x_tab_solve_async(handles[-1], 1, 1024.0, 48 * block, 0), "extra event samples")
            L.mi355x_tab_sync(handles[-1], ctypes.byref(npv))
            n2, a2, m2 = read_events(0)
            l2, la2, lm2 = read_events(1)
            if n2:
                extra_samples = n2
                upd_n, upd_avg_ms, upd_min_ms = n2, a2, m2
            if l2:
                la_n, la_avg_ms, la_min_ms = l2, la2, lm2

    if N > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        value = N * args.steps / elapsed
        roofline = None
        Cs = stored_cols.value
        if upd_avg_ms and resident:
            # the resident solve: the tableau is read once when a launch starts and written once when
            # it ends; inside the loop a pivot is one exchange through L2 and register arithmetic --
            # latency-bound by construction, HBM is idle
            per_launch = max(1, args.steps)
            ach = kernel_bytes / (upd_avg_ms * 1e-3) / 1e9
            roofline = {"bound": "latency", "kernel": "k_resident", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBPS, "traffic": pmc_traffic("resident", "k_resident " + args.workload)[0],
                        "traffic_source": pmc_traffic("resident", "k_resident " + args.workload)[1],
                        "what": "resident solve: the stored tableau lives in registers for the whole launch (load + "
                                "write-back = bytes_moved_per_launch); per pivot one all-to-all exchange through L2 "
                                "and register arithmetic, no HBM traffic -- the HBM figure is informational",
                        "kernel_avg_us": upd_avg_ms * 1e3, "launches_timed": int(upd_n),
                        "pivots_timed": per_launch, "us_per_pivot_in_kernel": upd_avg_ms * upd_n * 1e3 / per_launch,
                        "bytes_moved_per_launch": kernel_bytes,
                        "representation": "compact [non-basic columns | RHS], %d of %d columns stored" % (Cs, C)}
        elif upd_avg_ms:
            # ---- the model the path is held to ----------------------------
            # Blocked pivoting: per block of `block` pivots ONE sweep moves every stored element
            # through HBM once each way (bound: HBM, 2*R*Cs*8 bytes) and applies `block` rank-1
            # updates to it while it is in registers (2*block*R*Cs f64 operations, product and
            # difference rounded separately: the non-FMA vector rate is the second bound), and ONE
            # look-ahead launch selects the block's pivots: per pivot two dependent HBM round trips
            # (the entering column: R scattered doubles, the pivot row: Cs doubles) and two
            # all-to-all exchanges between its workgroups -- latency-bound by construction (a few MB
            # per launch), so its roofline entry is a time share and a per-step latency, not a rate
            # to maximise.  `achieved` is the PHYSICAL rate of the kernel (bytes it has to move /
            # its duration), comparable with the PMC `traffic` and bounded by the peak; the
            # contract's literal formula (per-pivot algorithmic bytes x pivots per launch / duration)
            # is `algorithmic_equivalent` and exceeds the peak by construction (the sweep does not
            # re-stream the tableau per pivot).
            upd_name = "k_sweep16" if block == 16 else (("k_sweepw_ring" if args.sweepw_ring else "k_sweepw") if block > 16 else
                                                        ("k_sweep" if block > 1 else L.mi355x_update_kernel_name().decode()))
            ach = kernel_bytes / (upd_avg_ms * 1e-3) / 1e9
            alg = block * kernel_bytes / (upd_avg_ms * 1e-3) / 1e9
            flops = 2.0 * block * R * Cs
            valu_min_us = flops / (F64_VALU_PEAK_TFLOPS * 1e12) * 1e6
            hbm_min_us = kernel_bytes / (HBM_PEAK_GBPS * 1e9) * 1e6
            traffic, traffic_src = pmc_traffic(args.workload, upd_name)
            per_block_ms = upd_avg_ms + (la_avg_ms or 0.0)
            kernels = [{
                "kernel": upd_name, "role": "tableau update: applies %d pivot(s) to every stored element" % block,
                "avg_us": upd_avg_ms * 1e3, "min_us": upd_min_ms * 1e3, "launches_timed": int(upd_n),
                "time_share": upd_avg_ms / per_block_ms,
                "bound": "hbm", "bytes_moved_per_launch": kernel_bytes, "achieved": ach, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "second_bound": {"what": "f64 vector ALU, non-FMA (2 * pivots * R * stored_cols operations)",
                                 "flops_per_launch": flops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "achieved": flops / (upd_avg_ms * 1e-3) / 1e12,
                                 "frac": flops / (upd_avg_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS,
                                 "min_us": valu_min_us},
                "min_us_by_bound": {"hbm": hbm_min_us, "f64_valu": valu_min_us},
                "binding_bound": "hbm" if hbm_min_us >= valu_min_us else "f64_valu",
            }]
            if la_avg_ms:
                la_bytes = block * (R * 8 * 2 + Cs * 8 * 2 + Cs * 8)      # column + RHS entries, pivot row + objective row, prow
                la_traffic, la_src = pmc_traffic(args.workload, "k_la_block")
                kernels.append({
                    "kernel": "k_la_block", "role": "look-ahead: selects the block's %d pivots (find-entering-column, "
                                                    "find-pivoting-row, row normalisation) ahead of the tableau" % block,
                    "avg_us": la_avg_ms * 1e3, "min_us": la_min_ms * 1e3, "launches_timed": int(la_n),
                    "time_share": la_avg_ms / per_block_ms,
                    "bound": "latency", "us_per_pivot": la_avg_ms * 1e3 / block,
                    "dependent_steps_per_pivot": "2 HBM round trips (entering column, pivot row) + 2 exchanges between the workgroups",
                    "bytes_moved_per_launch": la_bytes, "achieved": la_bytes / (la_avg_ms * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": la_bytes / (la_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "traffic": la_traffic, "traffic_source": la_src})
            kernels.sort(key=lambda kk: -kk["time_share"])
            whole = kernel_bytes / block * value / N / 1e9
            roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                        "kernel": upd_name,
                        # the honest figures next to `frac`, as scalars up front (the nested records below say how
                        # they come about): which kernel takes most of a block, and the whole iteration against the peak
                        "by_rocprofv3": (lambda a: None if a[0] is None else {
                            "kernel_avg_us": a[0], "launches": a[1], "source": a[2],
                            "achieved": kernel_bytes / (a[0] * 1e-6) / 1e9, "frac": kernel_bytes / (a[0] * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                            "what": "the same kernel in the committed rocprofv3 --kernel-trace --stats summary of this command "
                                    "(its own begin / end timestamps; the live figure above is a HIP-event bracket, which "
                                    "also holds the queue's hand-over on either side of the launch)"})(
                            rocprof_avg_us(args.workload, "%s<%d" % (upd_name, block) if block > 1 and upd_name != "k_sweep16" else upd_name)),
                        "dominant_kernel_by_time": kernels[0]["kernel"],
                        "dominant_kernel_time_share": kernels[0]["time_share"],
                        "whole_iteration_frac": whole / HBM_PEAK_GBPS,
                        "whole_iteration_GBps": whole,
                        "what": "the kernel that moves the tableau (>99 % of the HBM bytes of an iteration); "
                                "every kernel above 10 % of the time is in `kernels`, largest first",
                        "kernel_avg_us": upd_avg_ms * 1e3, "kernel_time_share": upd_avg_ms / per_block_ms,
                        "dominant_by_time": {"kernel": kernels[0]["kernel"], "time_share": kernels[0]["time_share"]},
                        "pivots_per_launch": block, "bytes_moved_per_launch": kernel_bytes,
                        "algorithmic_bytes_per_pivot": kernel_bytes,
                        "binding_bound": kernels[0]["binding_bound"] if kernels[0]["kernel"] == upd_name
                                         else [kk for kk in kernels if kk["kernel"] == upd_name][0]["binding_bound"],
                        "algorithmic_equivalent": {
                            "GBps": alg, "x_peak": alg / HBM_PEAK_GBPS,
                            "what": "algorithmic bytes per pivot x pivots per launch / launch duration"},
                        "representation": "compact [non-basic columns | RHS], %d of %d columns stored"
                                          % (Cs, C) if compact.value else "dense",
                        "dense_tableau_bytes_per_pivot": bytes_per_pivot,
                        "launches_timed": int(upd_n),
                        "launches_timed_after_the_timed_region": int(extra_samples),
                        "kernels": kernels,
                        "whole_iteration": {"what": "physical bytes per pivot (stored tableau once each way per block "
                                                    "/ pivots per block) x pivots/s, all kernels and gaps included",
                                            "physical_bytes_per_pivot": kernel_bytes / block, "GBps": whole,
                                            "frac": whole / HBM_PEAK_GBPS,
                                            "us_per_pivot": 1e6 / (value / N)}}
        rec = {
            "metric": baseline_metric() if args.workload == "cfg3"
                      else "simplex pivots/sec (%s)" % args.workload,
            "value": value, "unit": "pivots/s", "n_gpus": N, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE config %d: dense random LP %d vars x %d <=-constraints, "
                                   "%dx%d f64 tableau, one independent LP per GPU" % (cfg, n, m, R, C),
                       "tableau_bytes": R * C * 8, "fp_tolerance": 1024,
                       "parallelism": "independent LPs, %d rank(s), no collective" % N},
            "whole_pivot_GBps": kernel_bytes * value / N / 1e9,
            "dense_equivalent_GBps": bytes_per_pivot * value / N / 1e9,
            "roofline": roofline,
        }
        if N == 1 and args.workload == "cfg3" and not args.no_per_pivot:
            rec["per_pivot_kernel"] = per_pivot_record(lp, L, n, m, seed, local_rank, kernel_bytes, args.block or 0)
        if N == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"], state = cpu_baseline(lp, n, m, seed, args.cpu_pivots)
            rec["parity_in_run"] = parity_in_run(lp, L, n, m, seed, local_rank, state)
    for hk in handles:
        L.mi355x_tab_destroy(hk)
    handles = []
    torch.cuda.empty_cache()
    if rank == 0 and N == 1 and args.workload == "cfg3" and not args.no_other_configs and not args.block:
        rec["other_configs"] = other_configs(lp, L, local_rank, 16)
        ss = rec["other_configs"].get("cfg3_steady_state", {})
        if ss.get("value"):
            rec["steady_state_pivots_per_s"] = ss["value"]
            rec["steady_state_pivots_per_s_gpu_clock"] = ss["gpu_clock"]["pivots_per_s"]
            rec["gap_us_per_block"] = ss["gap_us_per_block"]

    # N > 1 with the default workload: the north star's multi-GPU claim is about ONE large tableau
    # column-partitioned over the GPUs with the pivot column travelling over RCCL / xGMI -- that
    # strong-scaling record is the headline; the independent LPs above (weak scaling, no
    # collective) are attached to it as a secondary field.  The leg runs under a watchdog: should
    # a rank fail or a collective never complete, the weak-scaling record is printed with the
    # reason instead of no line at all.
    if N > 1 and args.workload == "cfg3" and args.multi_gpu == "colpart":
        import threading
        limit = float(os.environ.get("BENCH_COLPART_TIMEOUT", "420"))

        progress = {}

        def give_up():
            if rank == 0:
                why = "column-partition leg did not finish within %.0f s" % limit
                if progress.get("rec"):               # the headline leg was done: only an A/B leg hung
                    out = progress["rec"]
                    out["colpart_error"] = why + " (during the exchange-mode A/B legs; the headline is complete)"
                    out["independent_lps_weak_scaling"] = {"value": rec["value"], "unit": "pivots/s", "scaling": "weak",
                                                           "ms_per_step": rec["ms_per_step"], "roofline": rec["roofline"]}
                    print(json.dumps(out), flush=True)
                else:
                    rec["colpart_error"] = why
                    print(json.dumps(rec), flush=True)
            os._exit(0)

        dog = threading.Timer(limit, give_up)
        dog.daemon = True
        dog.start()
        rec_colpart, err = None, None
        try:
            rec_colpart = importlib.import_module("linear-programming_amd.colpart").bench(args, rank, local_rank, N, progress)
        except BaseException as e:           # SystemExit included: a line is owed whatever happens
            err = "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)   # the ranks agree (a lone failure parks the others in a collective: watchdog)
        dog.cancel()
        if rank == 0:
            if int(flag.item()) == 0 and rec_colpart is not None:
                rec_colpart["independent_lps_weak_scaling"] = {
                    "what": "every rank iterating on its own 8192 x 4096 LP, no collective (BASELINE config 3 per GPU)",
                    "value": rec["value"], "unit": "pivots/s", "ms_per_step": rec["ms_per_step"],
                    "scaling": "weak", "roofline": rec["roofline"]}
                rec = rec_colpart
            else:
                rec["colpart_error"] = err or "the column-partition leg failed on another rank"
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
