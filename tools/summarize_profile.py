#!/usr/bin/env python3
"""Turn gpurun_out/prof (tools/collect_profile.sh) into the tracked artefacts under profiles/:
<prefix>_c3_kernel_stats.csv, <prefix>_c3_pmc_summary.csv, <prefix>_c3_bench.json,
<prefix>_c3_bench_under_rocprof.json and hbm_traffic.json (what bench.py reports as
roofline.traffic).  python tools/summarize_profile.py r01_final"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].strip()


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
    stats = glob.glob(os.path.join(SRC, "trace", "**", "*kernel_stats.csv"), recursive=True)
    assert stats, "no kernel_stats.csv under gpurun_out/prof/trace"
    shutil.copy(stats[0], os.path.join(DST, f"{prefix}_c3_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, f"{prefix}_c3_bench.json"))
    shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"),
                os.path.join(DST, f"{prefix}_c3_bench_under_rocprof.json"))
    rows = defaultdict(list)  # (kernel, counter) -> values
    for f in glob.glob(os.path.join(SRC, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(DST, f"{prefix}_c3_pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
        for (k, c), v in rows.items():
            w.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
    step = [k for (k, c) in rows if "gte_kernel<0" in k]
    assert step, "step kernel not found in the counter files"
    k = step[0]
    mean = lambda c: sum(rows[(k, c)]) / len(rows[(k, c)])
    fetch = mean("FETCH_SIZE") * 1024 * 2   # KB -> bytes; gfx950 reports half of wide reads
    write = mean("WRITE_SIZE") * 1024
    hit, miss = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
    path = os.path.join(DST, "hbm_traffic.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update({"c3_bytes_per_launch": fetch + write, "c3_fetch_bytes": fetch, "c3_write_bytes": write,
                "c3_l2_hit_rate": hit / (hit + miss), "kernel": k,
                "how": "tools/collect_profile.sh + tools/summarize_profile.py: rocprofv3 --pmc FETCH_SIZE / "
                       "--pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate passes, python bench.py "
                       "--steps 40 --warmup 10; mean over the step-kernel dispatches; KB->bytes x1024; "
                       "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, "
                       "MI355X_MICROARCH.md HBM section)"})
    json.dump(old, open(path, "w"), indent=1)
    for r in csv.DictReader(open(stats[0])):
        if "gte_kernel<0" in r["Name"]:
            print("rocprof", short(r["Name"]), "calls", r["Calls"], "AverageNs", r["AverageNs"])
    b = json.load(open(os.path.join(SRC, "bench.json")))
    print("bench: ms_per_step", b["ms_per_step"], "kernel_us", b["roofline"]["kernel_us_per_launch"],
          "value", b["value"])
    print("traffic per launch: write %.1f MB + fetch %.1f MB = %.1f MB, L2 hit rate %.2f"
          % (write / 1e6, fetch / 1e6, (write + fetch) / 1e6, hit / (hit + miss)))


if __name__ == "__main__":
    main()
