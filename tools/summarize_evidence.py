"""Copy the judged summaries from gpurun_out/evidence/ (scratch) into profiles/ and derive the
corrected per-launch HBM traffic of the tableau-update kernel (k_sweep / k_update) from the two PMC passes.

    python tools/summarize_evidence.py r01

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


def one(pattern):
    g = glob.glob(os.path.join(EV, pattern))
    if not g:
        raise SystemExit("missing " + pattern)
    return max(g, key=os.path.getmtime)       # gpurun merges runs: take the newest


def counter(path, kernel_sub):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel_sub in r["Kernel_Name"]]
    return v


def main(tag):
    os.makedirs(PR, exist_ok=True)
    shutil.copy(one("kernel_stats/*/*kernel_stats.csv"), os.path.join(PR, tag + "_cfg3_kernel_stats.csv"))
    fpath = os.path.join(PR, tag + "_cfg3_pmc_FETCH_SIZE.csv")
    wpath = os.path.join(PR, tag + "_cfg3_pmc_WRITE_SIZE.csv")
    shutil.copy(one("pmc_FETCH_SIZE/*/*counter_collection.csv"), fpath)
    shutil.copy(one("pmc_WRITE_SIZE/*/*counter_collection.csv"), wpath)
    with open(one("bench.log")) as f:
        bench = [l for l in f.read().splitlines() if l.startswith("{")][-1]
    open(os.path.join(PR, tag + "_bench_cfg3.json"), "w").write(bench + "\n")
    layout = {}
    for l in open(one("pmc_FETCH_SIZE.log")):
        if l.startswith("layout"):
            layout = dict(kv.split("=") for kv in l.split()[1:])
    rows, cols, ld = int(layout["rows"]), int(layout["stored_cols"]), int(layout["stored_ld"])
    kernel = json.loads(bench)["roofline"]["kernel"]          # k_sweep (blocked) or k_update
    f = counter(fpath, kernel)
    w = counter(wpath, kernel)
    cf, cw = max(counter(fpath, "copyBuffer")), max(counter(wpath, "copyBuffer"))
    dense_ld = (8192 + 4096 + 1 + 15) // 16 * 16
    copy_kib = rows * dense_ld * 8 / 1024.0           # mi355x_tab_copy copies the DENSE padded buffer
    favg, wavg = sum(f) / len(f), sum(w) / len(w)
    traffic = (2 * favg + wavg) * 1024
    alg = 2 * rows * cols * 8
    d = {"workload": "cfg3", "kernel": kernel, "launches": len(f),
         "pivots_per_launch": json.loads(bench)["roofline"].get("pivots_per_launch", 1),
         "representation": "compact" if int(layout["compact"]) else "dense",
         "stored_rows_cols_ld": [rows, cols, ld],
         "FETCH_SIZE_KiB_avg": favg, "WRITE_SIZE_KiB_avg": wavg,
         "calibration": {"what": "__amd_rocclr_copyBuffer (mi355x_tab_copy) of the dense padded "
                                 "tableau: reads and writes rows*ld*8 bytes",
                         "KiB_each_way": copy_kib, "FETCH_SIZE_reported_KiB": cf,
                         "WRITE_SIZE_reported_KiB": cw, "fetch_factor_measured": copy_kib / cf,
                         "write_factor_measured": copy_kib / cw,
                         "correction_applied": "FETCH x2 (gfx950), WRITE x1"},
         "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
         "traffic_over_algorithmic": traffic / alg}
    json.dump(d, open(os.path.join(PR, tag + "_cfg3_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(d, indent=1))
    print(open(os.path.join(PR, tag + "_cfg3_kernel_stats.csv")).read()[:1500])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
