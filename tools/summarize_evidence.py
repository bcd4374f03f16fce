"""Copy the judged summaries from gpurun_out/evidence/ (scratch) into profiles/ and derive the
corrected per-launch HBM traffic of every profiled kernel from the two PMC passes.

    python tools/summarize_evidence.py r02

FETCH_SIZE / WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read (MI355X_MICROARCH.md, HBM section): the factor is re-measured here on the
device-to-device copy of the same padded buffer that tools/pmc_probe.py performs first."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV = os.path.join(ROOT, "gpurun_out", "evidence")
PR = os.path.join(ROOT, "profiles")


def one(pattern, required=True):
    g = glob.glob(os.path.join(EV, pattern))
    if not g:
        if required:
            raise SystemExit("missing " + pattern)
        return None
    return max(g, key=os.path.getmtime)       # gpurun merges runs: take the newest


def counter(path, kernel_sub):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel_sub in r["Kernel_Name"]]


def last_json(path):
    with open(path) as f:
        return [l for l in f.read().splitlines() if l.startswith("{")][-1]


def main(tag):
    os.makedirs(PR, exist_ok=True)
    for src, dst in (("kernel_stats/*/*kernel_stats.csv", "_cfg3_kernel_stats.csv"),
                     ("kernel_stats_cfg4/*/*kernel_stats.csv", "_cfg4_kernel_stats.csv"),
                     ("kernel_stats_perpivot/*/*kernel_stats.csv", "_perpivot_cfg3_kernel_stats.csv"),
                     ("pmc_SQ/*/*counter_collection.csv", "_cfg3_pmc_SQ.csv")):
        f = one(src, required=False)
        if f:
            shutil.copy(f, os.path.join(PR, tag + dst))
    for log, dst in (("bench.log", "_bench_cfg3.json"), ("bench_driver_flags.log", "_bench_cfg3_driver_flags.json"),
                     ("bench_cfg4.log", "_bench_cfg4_128lps.json"), ("bench_cfg4_1024.log", "_bench_cfg4_1024lps.json"),
                     ("bench_cfg2.log", "_bench_cfg2.json"), ("bench_colpart_1gpu.log", "_bench_cfg5_colpart_1gpu.json")):
        f = one(log, required=False)
        if f:
            try:
                open(os.path.join(PR, tag + dst), "w").write(last_json(f) + "\n")
            except IndexError:
                print("no JSON line in", log)
    for log, dst in (("shard_step_cost.log", "_shard_step_cost.txt"), ("resident_timing.log", "_resident_timing.txt"),
                     ("resident_ab.log", "_resident_poll_ab.txt"), ("native_end_to_end.log", "_native_end_to_end.log")):
        f = one(log, required=False)
        if f and os.path.getsize(f):
            shutil.copy(f, os.path.join(PR, tag + dst))
    out = {"workload": "cfg3", "kernels": {}}
    dense_ld = (8192 + 4096 + 1 + 15) // 16 * 16
    for variant, kernels in (("", ("k_sweepw_ring<", "k_sweepw<", "k_sweepw_rest", "k_sweep16", "k_la_block")),
                             ("perpivot_", ("k_update", "k_select_gather", "k_select_scale"))):
        fp = one("pmc_%sFETCH_SIZE/*/*counter_collection.csv" % variant, required=False)
        wp = one("pmc_%sWRITE_SIZE/*/*counter_collection.csv" % variant, required=False)
        lg = one("pmc_%sFETCH_SIZE.log" % variant, required=False)
        if not (fp and wp and lg):
            continue
        shutil.copy(fp, os.path.join(PR, "%s_%scfg3_pmc_FETCH_SIZE.csv" % (tag, variant)))
        shutil.copy(wp, os.path.join(PR, "%s_%scfg3_pmc_WRITE_SIZE.csv" % (tag, variant)))
        layout = {}
        for l in open(lg):
            if l.startswith("layout"):
                layout = dict(kv.split("=") for kv in l.split()[1:])
        rows, cols, ld = int(layout["rows"]), int(layout["stored_cols"]), int(layout["stored_ld"])
        cf, cw = max(counter(fp, "copyBuffer")), max(counter(wp, "copyBuffer"))
        copy_kib = rows * dense_ld * 8 / 1024.0           # mi355x_tab_copy copies the DENSE padded buffer
        out["calibration"] = {"what": "__amd_rocclr_copyBuffer (mi355x_tab_copy) of the dense padded tableau: reads "
                                      "and writes rows*ld*8 bytes", "KiB_each_way": copy_kib,
                              "FETCH_SIZE_reported_KiB": cf, "WRITE_SIZE_reported_KiB": cw,
                              "fetch_factor_measured": copy_kib / cf, "write_factor_measured": copy_kib / cw,
                              "correction_applied": "FETCH x2 (gfx950), WRITE x1"}
        for k in kernels:
            f, w = counter(fp, k), counter(wp, k)
            if not f or not w:
                continue
            if k in ("k_sweepw_ring<", "k_sweepw<", "k_sweep16", "k_update"):   # steady state: drop launches that did nothing
                f = [x for x in f if x > 0.5 * max(f)]
                w = [x for x in w if x > 0.5 * max(w)]
            favg, wavg = sum(f) / len(f), sum(w) / len(w)
            rec = {"launches": len(f), "FETCH_SIZE_KiB_avg": favg, "WRITE_SIZE_KiB_avg": wavg,
                   "hbm_bytes_per_launch": (2 * favg + wavg) * 1024,
                   "representation": "compact" if int(layout["compact"]) else "dense",
                   "stored_rows_cols_ld": [rows, cols, ld]}
            if k in ("k_sweepw_ring<", "k_sweepw<", "k_sweep16", "k_update"):
                rec["algorithmic_bytes_per_launch"] = 2 * rows * cols * 8
                rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
            out["kernels"][k.rstrip("<")] = rec
    json.dump(out, open(os.path.join(PR, tag + "_cfg3_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    # wide blocks (round 4): one 8-GPU-sized shard of config 5 as a single tableau (32769 x 8193 stored,
    # 2.15 GB), 24 pivots per sweep: k_sweepw<24> + k_sweepw_rest
    fp = one("pmc_shard_FETCH_SIZE/*/*counter_collection.csv", required=False)
    wp = one("pmc_shard_WRITE_SIZE/*/*counter_collection.csv", required=False)
    lg = one("pmc_shard_FETCH_SIZE.log", required=False)
    if fp and wp and lg:
        shutil.copy(fp, os.path.join(PR, tag + "_cfg5_shard_pmc_FETCH_SIZE.csv"))
        shutil.copy(wp, os.path.join(PR, tag + "_cfg5_shard_pmc_WRITE_SIZE.csv"))
        layout = {}
        for l in open(lg):
            if l.startswith("layout"):
                layout = dict(kv.split("=") for kv in l.split()[1:])
        rows, cols, ld, dld = int(layout["rows"]), int(layout["stored_cols"]), int(layout["stored_ld"]), int(layout["dense_ld"])
        cf, cw = max(counter(fp, "copyBuffer")), max(counter(wp, "copyBuffer"))
        copy_kib = rows * dld * 8 / 1024.0
        res = {"workload": "one 8-GPU-sized column shard of config 5 as a single tableau (%d x %d stored), %s pivots per sweep"
                           % (rows, cols, layout.get("block", "?")),
               "calibration": {"KiB_each_way": copy_kib, "FETCH_SIZE_reported_KiB": cf, "WRITE_SIZE_reported_KiB": cw,
                               "fetch_factor_measured": copy_kib / cf, "write_factor_measured": copy_kib / cw,
                               "correction_applied": "FETCH x2 (gfx950), WRITE x1"}, "kernels": {}}
        for k in ("k_sweepw_ring<", "k_sweepw<", "k_sweepw_rest", "k_la_gather", "k_la_scale"):
            f, w = counter(fp, k), counter(wp, k)
            if not f or not w:
                continue
            if k in ("k_sweepw<", "k_sweepw_ring<"):
                f = [x for x in f if x > 0.5 * max(f)]
                w = [x for x in w if x > 0.5 * max(w)]
            favg, wavg = sum(f) / len(f), sum(w) / len(w)
            rec = {"launches": len(f), "FETCH_SIZE_KiB_avg": favg, "WRITE_SIZE_KiB_avg": wavg,
                   "hbm_bytes_per_launch": (2 * favg + wavg) * 1024, "stored_rows_cols_ld": [rows, cols, ld]}
            if k in ("k_sweepw<", "k_sweepw_ring<"):
                rec["algorithmic_bytes_per_launch"] = 2 * rows * cols * 8
                rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
            res["kernels"][k.rstrip("<")] = rec
        json.dump(res, open(os.path.join(PR, tag + "_cfg5_shard_pmc_traffic.json"), "w"), indent=1)
        print(json.dumps(res, indent=1))
    for src, dst in (("kernel_stats_cfg5/*/*kernel_stats.csv", "_cfg5_kernel_stats.csv"),
                     ("kernel_stats_shard/*/*kernel_stats.csv", "_cfg5_shard_kernel_stats.csv")):
        f = one(src, required=False)
        if f:
            shutil.copy(f, os.path.join(PR, tag + dst))
    f = one("kernel_stats_shard_step/*/*kernel_stats.csv", required=False)
    if f:                                                   # the two-launch step of one 8-GPU-sized shard
        shutil.copy(f, os.path.join(PR, tag + "_shard_step_kernel_stats.csv"))
    f = one("kernel_stats_shard_block/*/*kernel_stats.csv", required=False)
    if f:                                                   # the same shard behind the persistent block launch (k_shard_la_block)
        shutil.copy(f, os.path.join(PR, tag + "_shard_block_kernel_stats.csv"))
    for log, dst in (("shard_la_timing.log", "_shard_la_timing.txt"), ("la_policy_ab.log", "_la_policy_ab.txt"),
                     ("ring_waves_ab.log", "_ring_waves_ab.txt"), ("pytest_gpu_summary.log", "_pytest_gpu_summary.txt")):
        f = one(log, required=False)
        if f and os.path.getsize(f):
            shutil.copy(f, os.path.join(PR, tag + dst))
    f = one("pmc_resident_SQ/*/*counter_collection.csv", required=False)
    if f:                                                   # instruction counters of the resident launches
        rows_ = list(csv.DictReader(open(f)))
        keep = [r for r in rows_ if "k_resident" in r["Kernel_Name"]]
        if keep:
            with open(os.path.join(PR, tag + "_resident_pmc_SQ.csv"), "w", newline="") as fo:
                w_ = csv.DictWriter(fo, fieldnames=list(keep[0].keys()))
                w_.writeheader()
                w_.writerows(keep)
    for log, dst in (("steady_gap.log", "_steady_gap.txt"), ("la_timing.log", "_la_timing.txt"), ("sweep_lds_microbench.log", "_sweep_lds_microbench.txt"),
                     ("ring_ab_kernel_stats.log", "_ring_ab_kernel_stats.txt"), ("wide_block_ab.log", "_wide_block_ab.txt"), ("sweep32_microbench.log", "_sweep32_microbench.txt"),
                     ("resident_lds_ab.log", "_resident_lds_ab.txt"), ("fuzz_totals.log", "_fuzz_totals.txt"),
                     ("la_wide_ab.log", "_la_wide_ab.txt"), ("shard_step_skeleton.log", "_shard_step_skeleton.txt")):
        f = one(log, required=False)
        if f and os.path.getsize(f):
            shutil.copy(f, os.path.join(PR, tag + dst))
    # the resident solve (tools/pmc_probe_resident.py): config 2 as ONE k_resident launch, then a
    # 128-LP config-4 batch as one launch
    fp = one("pmc_resident_FETCH_SIZE/*/*counter_collection.csv", required=False)
    wp = one("pmc_resident_WRITE_SIZE/*/*counter_collection.csv", required=False)
    lg = one("pmc_resident_FETCH_SIZE.log", required=False)
    if fp and wp and lg:
        shutil.copy(fp, os.path.join(PR, tag + "_resident_pmc_FETCH_SIZE.csv"))
        shutil.copy(wp, os.path.join(PR, tag + "_resident_pmc_WRITE_SIZE.csv"))
        layout, piv = {}, {}
        for l in open(lg):
            if l.startswith("layout"):
                layout = dict(kv.split("=") for kv in l.split()[1:])
            if l.startswith("cfg2 rc"):
                piv["cfg2"] = int(l.split()[-1])
            if l.startswith("cfg4 batch"):
                piv["cfg4"] = int(l.split()[3])
        rows, cols, dld = int(layout["rows"]), int(layout["stored_cols"]), int(layout["dense_ld"])
        cf, cw = max(counter(fp, "copyBuffer")), max(counter(wp, "copyBuffer"))
        copy_kib = rows * dld * 8 / 1024.0
        res = {"calibration": {"KiB_each_way": copy_kib, "FETCH_SIZE_reported_KiB": cf, "WRITE_SIZE_reported_KiB": cw,
                               "fetch_factor_measured": copy_kib / cf, "write_factor_measured": copy_kib / cw,
                               "correction_applied": "FETCH x2 (gfx950), WRITE x1"}, "kernels": {}}
        f, w = counter(fp, "k_resident"), counter(wp, "k_resident")
        names = ["cfg2 (513 x 1025 stored, 32 workgroups, whole solve = 1 launch)",
                 "cfg4 (128 LPs of 257 x 513 stored, 8 workgroups each, whole batch = 1 launch)"]
        big = sorted(range(len(f)), key=lambda i: -(2 * f[i] + w[i]))[:2]
        for name, i, key, stored in zip(names, sorted(big), ("cfg2", "cfg4"), (rows * cols * 8, 128 * 257 * 513 * 8)):
            hb = (2 * f[i] + w[i]) * 1024
            res["kernels"]["k_resident " + name] = {
                "FETCH_SIZE_KiB": f[i], "WRITE_SIZE_KiB": w[i], "hbm_bytes_per_launch": hb, "pivots_in_the_launch": piv.get(key),
                "stored_tableau_bytes": stored, "traffic_over_load_plus_writeback": hb / (2.0 * stored),
                "hbm_bytes_per_pivot": hb / piv[key] if piv.get(key) else None}
        json.dump(res, open(os.path.join(PR, tag + "_resident_pmc_traffic.json"), "w"), indent=1)
        print(json.dumps(res, indent=1))
    f = one("kernel_stats_cfg2/*/*kernel_stats.csv", required=False)
    if f:
        shutil.copy(f, os.path.join(PR, tag + "_cfg2_kernel_stats.csv"))
    f = one("native_end_to_end.log", required=False)
    if f:
        shutil.copy(f, os.path.join(PR, tag + "_native_end_to_end.log"))
    for f in sorted(glob.glob(os.path.join(PR, tag + "*kernel_stats.csv"))):
        print(f)
        print(open(f).read()[:900])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
