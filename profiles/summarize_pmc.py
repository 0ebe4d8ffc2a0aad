#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of profiles/collect.sh (gpurun_out/prof_<tag>/) into
profiles/<tag>_pmc_summary.json + copies of the CSVs.  FETCH_SIZE / WRITE_SIZE are KiB; they are divided
by the reported/true ratio measured on bench/store_calib's known byte counts in the same run
(MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 on gfx950 for coalesced loads)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03_C2"
config = sys.argv[2] if len(sys.argv) > 2 else "C2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", "prof_" + tag)
calib = os.path.join(root, "gpurun_out", "prof_calib")


def newest(pattern):
    # gpurun merges every call's outputs into the same directories: take the latest run's file, not an earlier one's
    return max(glob.glob(pattern), key=os.path.getmtime)


def load(d):
    f = newest(os.path.join(calib if d.startswith("calib") else base, d, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}, f


(fetch, f1), (write, f2), (cf, f3), (cw, f4), (sq, f5) = [load(d) for d in ("fetch", "write", "calib_fetch", "calib_write", "sq")]
calib_bytes = 2 << 30
wcal = [v["WRITE_SIZE"] for k, v in cw.items() if "calib_write" in k][0] * 1024 / calib_bytes
fcal = [v["FETCH_SIZE"] for k, v in cf.items() if "calib_read" in k][0] * 1024 / calib_bytes
out = {"source": "rocprofv3 --pmc (separate passes) on `python3 bench.py --config %s --steps 2 --warmup 1 --no-cpu --no-e2e --slots 1`, MI355X; "
                 "per-launch averages over all launches of the run" % config,
       "workload": config,
       "units": "FETCH_SIZE/WRITE_SIZE are KiB; corrected as MI355X_MICROARCH.md §HBM prescribes and as calibrated here",
       "calibration": {"known_bytes": calib_bytes, "WRITE_SIZE_reported_over_true": wcal,
                       "FETCH_SIZE_reported_over_true": fcal},
       "kernels": {}}
for k in fetch:
    if "fadehip" not in k:
        continue
    f = fetch[k]["FETCH_SIZE"] * 1024 / fcal
    w = write[k]["WRITE_SIZE"] * 1024 / wcal
    e = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
    if k in sq:
        s = sq[k]
        e.update(s)
        cyc = s["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        e["valu_busy_frac"] = s["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)  # quad-cycles; 1024 SIMDs
    out["kernels"][k] = e
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
shutil.copy(newest(os.path.join(base, "stats", "*", "*_kernel_stats.csv")), os.path.join(root, "profiles", tag + "_kernel_stats_1slot.csv"))
shutil.copy(newest(os.path.join(base, "stats2", "*", "*_kernel_stats.csv")), os.path.join(root, "profiles", tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(base, "bench_under_rocprof.json"), os.path.join(root, "profiles", tag + "_bench_under_rocprof.json"))
for d, f in (("fetch", f1), ("write", f2), ("calib_fetch", f3), ("calib_write", f4), ("sq", f5)):
    shutil.copy(f, os.path.join(root, "profiles", "%s_pmc_%s.csv" % (tag, d)))
for k, e in out["kernels"].items():
    print("%-70s hbm %.1f MB  valu_busy %.2f" % (k[:70], e["hbm_bytes_per_launch"] / 1e6, e.get("valu_busy_frac", 0)))
