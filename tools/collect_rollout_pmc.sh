#!/bin/bash
# PMC evidence for gte_rollout (through gpurun): bytes per STEP the fused kernels fetch / write,
# window-resident kernel vs the gather-per-step one (kernel_variant 256), configs 3 and 5.
#   bash tools/collect_rollout_pmc.sh   -> gpurun_out/rollout_pmc/summary.txt
set -e
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/rollout_pmc
rm -rf "$OUT" && mkdir -p "$OUT"
export TMPDIR=/tmp
K=64
for wl in c3 c5; do
  for variant in 0 256; do
    for c in FETCH_SIZE WRITE_SIZE; do
      d=/tmp/rpmc_${wl}_${variant}_$c
      rm -rf $d
      (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $d -o pmc -- \
        python3 $ROOT/tools/rollout_bench.py --workload $wl --k $K --reps 2 --variant $variant \
        > /dev/null 2> "$OUT/${wl}_${variant}_$c.err")
      cp $(find $d -name "*counter_collection.csv" | head -1) "$OUT/${wl}_${variant}_$c.csv"
      echo "[rollout pmc] $wl variant $variant $c done"
    done
  done
done
python3 - "$OUT" $K <<'PY'
import csv, glob, os, sys
out, K = sys.argv[1], int(sys.argv[2])
lines = ["gte_rollout with observations of every step, K = %d steps per launch; bytes per STEP of the "
         "whole batch (FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024; rocprofv3 --pmc, one counter per pass)" % K]
for wl, envs in (("c3", 65536), ("c5", 32768)):
    for variant, label in ((0, "window-resident"), (256, "gather-per-step")):
        vals = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(out, f"{wl}_{variant}_{c}.csv")))
                    if "gte_rollout" in r["Kernel_Name"] and "state" not in r["Kernel_Name"]
                    and r["Counter_Name"] == c and int(r["Grid_Size"]) > 0]
            # the launches with per-step observations are the big writers: take the largest
            vals[c] = max(rows) if rows else float("nan")
        fetch = vals["FETCH_SIZE"] * 1024 * 2 / K
        write = vals["WRITE_SIZE"] * 1024 / K
        lines.append(f"{wl} {envs:6d} envs  {label:16s} fetch {fetch / 1e6:8.2f} MB/step ({fetch / envs:7.1f} B/env)"
                     f"   write {write / 1e6:8.2f} MB/step ({write / envs:7.1f} B/env)")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
