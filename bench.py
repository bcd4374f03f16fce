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
    ap.add_argument("--sweep-tr", type=int, default=0, help="tuning: rows per sweep workgroup")
    ap.add_argument("--sweep-nt", type=int, default=-1, help="tuning: non-temporal sweep accesses (0/1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pivots", type=int, default=400,
                    help="cpu_baseline: pivots timed with all host threads (a quarter of it single-threaded)")
    ap.add_argument("--no-events", action="store_true",
                    help="do not bracket the update launches with HIP events (roofline = null)")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket every k-th update launch of the timed region with a HIP event pair")
    return ap.parse_args()


def cpu_baseline(lp, n, m, seed, pivots):
    """The oracle (C restatement of the reference algorithm; the Lisp reference itself cannot
    run here) timed on the host cores on the first `pivots` pivots of the same LP."""
    import oracle
    M, b = lp.synth.tableau(n, m, seed)
    threads = oracle.omp_threads()
    t0 = time.perf_counter()
    st, npiv, _ = oracle.solve(M, b, max_pivots=pivots, omp=True)
    t_omp = time.perf_counter() - t0
    single = max(2, pivots // 4)
    t0 = time.perf_counter()
    st1, npiv1, _ = oracle.solve(M, b, max_pivots=single, omp=False)
    t_one = time.perf_counter() - t0
    return {
        "value": npiv / t_omp, "unit": "pivots/s", "cores": threads, "kind": "port",
        "sample": "first %d pivots of the same %dx%d LP, OpenMP row-parallel C restatement of "
                  "src/simplex.lisp:337-461 (SBCL unavailable in the image); single-thread: "
                  "%.3f pivots/s over the next %d pivots" % (npiv, n, m, npiv1 / t_one, npiv1),
        "single_thread_value": npiv1 / t_one,
        "GBps": 2.0 * (m + 1) * (n + m + 1) * 8 * npiv / t_omp / 1e9,
    }


def baseline_metric():
    """The metric string exactly as BASELINE.json spells it."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "simplex pivots/sec + achieved HBM GB/s, dense 8192\u00d74096 f64 tableau"


def pmc_traffic(workload):
    """HBM bytes per k_update launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH doubled per the gfx950 correction, calibrated on a copy of
    the same buffer -- profiles/r01_cfg3_pmc_traffic.json).  PMC counters cannot be collected
    from inside this process, so the number is the last profiled one for this workload, or None."""
    path = os.path.join(ROOT, "profiles", "r01_%s_pmc_traffic.json" % workload)
    try:
        with open(path) as f:
            d = json.load(f)
        return d["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    except Exception:
        return None, None


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
        L.mi355x_tune_set_sweep_shape(args.sweep_tr or 4, args.sweep_nt)
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
            L.mi355x_tab_timing_enable(h, max(1, args.event_stride))
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

    upd_avg_ms = None
    nl = ctypes.c_int64(0)
    if not args.no_events:
        tot_n, tot_ms = 0, 0.0
        for hk in handles:
            sm, mn = ctypes.c_double(0), ctypes.c_double(0)
            L.mi355x_tab_timing_read(hk, ctypes.byref(nl), ctypes.byref(sm), ctypes.byref(mn))
            tot_n += nl.value
            tot_ms += sm.value
        nl = ctypes.c_int64(tot_n)
        if tot_n > 0:
            upd_avg_ms = tot_ms / tot_n

    if N > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        value = N * args.steps / elapsed
        roofline = None
        if upd_avg_ms:
            # One launch of the dominant kernel moves every STORED element once each way and
            # applies `block` pivots to it (blocked pivoting; block == 1 is the
            # plain k_update).  `achieved` is the PHYSICAL rate of that launch -- bytes it has
            # to move / its duration, directly comparable with the PMC `traffic` and bounded by
            # the HBM peak.  The contract's literal formula (per-pivot algorithmic bytes x
            # pivots per launch / duration) is given next to it as `algorithmic_equivalent`: it
            # exceeds the HBM peak by construction, because the blocked sweep does NOT re-stream
            # the tableau for every pivot -- that is its point.
            ach = kernel_bytes / (upd_avg_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(args.workload)
            alg = block * kernel_bytes / (upd_avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                        "traffic_source": traffic_src,
                        "kernel": "k_sweep" if block > 1 else L.mi355x_update_kernel_name().decode(),
                        "kernel_avg_us": upd_avg_ms * 1e3,
                        "pivots_per_launch": block,
                        "bytes_moved_per_launch": kernel_bytes,
                        "algorithmic_bytes_per_pivot": kernel_bytes,
                        "algorithmic_equivalent": {
                            "GBps": alg, "x_peak": alg / HBM_PEAK_GBPS,
                            "what": "algorithmic bytes per pivot x pivots per launch / launch duration"},
                        "representation": "compact [non-basic columns | RHS], %d of %d columns stored"
                                          % (stored_cols.value, C) if compact.value else "dense",
                        "dense_tableau_bytes_per_pivot": bytes_per_pivot,
                        "launches_timed": int(nl.value)}
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
        if N == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(lp, n, m, seed, args.cpu_pivots)
        print(json.dumps(rec), flush=True)
    for hk in handles:
        L.mi355x_tab_destroy(hk)
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
