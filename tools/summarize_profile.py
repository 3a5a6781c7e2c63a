#!/usr/bin/env python3
"""Turn gpurun_out/prof (tools/collect_profile.sh) into the tracked artefacts under profiles/:
<prefix>_<wl>_<envs>_kernel_stats.csv, <prefix>_<wl>_pmc_summary.csv, <prefix>_<wl>_bench.json,
<prefix>_<wl>_<envs>_bench_under_rocprof.json and the entries "<wl>_<envs>" of hbm_traffic.json
(bench.py's fallback for roofline.traffic when it cannot run rocprofv3 itself).
    python tools/summarize_profile.py r02 [c3]"""
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
sys.path.insert(0, ROOT)


def short(name):
    return name.split("(")[0].strip()


def main():
    import bench
    prefix = sys.argv[1] if len(sys.argv) > 1 else "r02"
    wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
    shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, f"{prefix}_{wl}_bench.json"))
    path = os.path.join(DST, "hbm_traffic.json")
    rec = json.load(open(path)) if os.path.exists(path) else {}
    summary = []
    for tdir in sorted(glob.glob(os.path.join(SRC, "trace_*"))):
        if not os.path.isdir(tdir):
            continue
        n = int(os.path.basename(tdir).split("_")[1])
        if n and wl != "c3":  # a stale directory of an earlier c3 collection merged by gpurun
            continue
        envs = n or bench.WORKLOADS[wl]["envs"]
        stats = glob.glob(os.path.join(tdir, "**", "*kernel_stats.csv"), recursive=True)
        assert stats, f"no kernel_stats.csv under {tdir}"
        shutil.copy(stats[0], os.path.join(DST, f"{prefix}_{wl}_{envs}_kernel_stats.csv"))
        shutil.copy(os.path.join(SRC, f"bench_under_rocprof_{n}.json"),
                    os.path.join(DST, f"{prefix}_{wl}_{envs}_bench_under_rocprof.json"))
        rows = defaultdict(list)  # (kernel, counter) -> values in dispatch order
        for f in glob.glob(os.path.join(SRC, f"pmc_{n}_*", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                rows[(short(r["Kernel_Name"]), r["Counter_Name"])].append(
                    (int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        step = sorted({k for (k, c) in rows if "gte_kernel<0" in k})
        assert step, "step kernel not found in the counter files"
        k = step[0]
        # the 40 launches after the de-synchronising prologue and the warm-up
        mean = lambda c: (lambda v: sum(v) / len(v))([x for _, x in sorted(rows[(k, c)])][-40:])
        fetch = mean("FETCH_SIZE") * 1024 * 2   # KB -> bytes; gfx950 reports half of wide reads
        write = mean("WRITE_SIZE") * 1024
        hit, miss = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
        avg_ns = calls = None
        for r in csv.DictReader(open(stats[0])):
            if "gte_kernel<0" in r["Name"]:
                avg_ns, calls = float(r["AverageNs"]), int(r["Calls"])
        rec[f"{wl}_{envs}"] = {
            "bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
            "l2_hit_rate": hit / (hit + miss), "kernel": k, "round": prefix,
            "rocprof_average_ns": avg_ns, "rocprof_calls": calls,
            "regime": "infinity-cache" if envs * 2560 <= (190 << 20) else "hbm",
            "how": "tools/collect_profile.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc "
                   "TCC_HIT_sum TCC_MISS_sum, separate passes over `python3 bench.py --pmc-child "
                   "--steps 40 --warmup 10` (episodes de-synchronised); mean over the last 40 "
                   "step-kernel dispatches; KB->bytes x1024; FETCH_SIZE doubled (gfx950 reports 1/2 "
                   "of wide coalesced reads, MI355X_MICROARCH.md HBM section)"}
        for (kk, c), v in rows.items():
            vals = [x for _, x in v]
            summary.append([envs, kk, c, len(vals), sum(vals) / len(vals), min(vals), max(vals)])
        print(f"{wl} {envs} envs: rocprof {k} calls {calls} AverageNs {avg_ns}; traffic per launch: "
              f"write {write / 1e6:.1f} MB + fetch {fetch / 1e6:.1f} MB = {(write + fetch) / 1e6:.1f} MB "
              f"= {(write + fetch) / (avg_ns * 1e-9) / 1e12:.2f} TB/s, L2 hit rate {hit / (hit + miss):.2f}")
    with open(os.path.join(DST, f"{prefix}_{wl}_pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["envs", "kernel", "counter", "dispatches", "mean", "min", "max"])
        w.writerows(summary)
    json.dump(rec, open(path, "w"), indent=1)
    b = json.load(open(os.path.join(SRC, "bench.json")))
    r = b["roofline"]
    print("bench: ms_per_step", b["ms_per_step"], "kernel_us", r["kernel_us_per_launch"], "value", b["value"],
          "frac", r["frac"], "episodes", b["config"]["episodes_finished"])
    if "hbm_regime" in r:
        h = r["hbm_regime"]
        print("hbm regime:", h["envs"], "envs", h["kernel_us_per_launch"], "us, frac", h["frac"],
              "traffic", h["traffic"])


if __name__ == "__main__":
    main()
