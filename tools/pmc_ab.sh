#!/bin/bash
# Hardware counters of the step kernel for several BUILDS of libgte (A/B of kernel variants):
#   bash tools/pmc_ab.sh [workload] libA.so libB.so ...   -> gpurun_out/pmc_ab/summary.txt
# One rocprofv3 --pmc pass per counter group and library (no trace domain next to --pmc).
set -e
cd "$(dirname "$0")/.."
ROOT=$(pwd)
WL=$1; shift
OUT=$ROOT/gpurun_out/pmc_ab
rm -rf "$OUT" && mkdir -p "$OUT"
export TMPDIR=/tmp
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
G3="FETCH_SIZE"
G4="WRITE_SIZE"
G5="TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
G6="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
for lib in "$@"; do
  name=$(basename $lib .so)
  k=0
  for g in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6"; do
    k=$((k+1))
    (cd /tmp && GTE_LIBRARY=$ROOT/gym-trading-env_amd/csrc/$lib rocprofv3 --pmc $g --output-format csv -d /tmp/pmc_ab_${name}_$k -o pmc -- \
        python3 $ROOT/bench.py --workload $WL --pmc-child --steps 24 --warmup 8 > /dev/null 2> "$OUT/${name}_$k.err") || echo "[pmc_ab] group $k failed for $name"
    f=$(find /tmp/pmc_ab_${name}_$k -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/${name}_$k.csv"
    echo "[pmc_ab] $name group $k done"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
vals = defaultdict(dict)
for f in sorted(glob.glob(os.path.join(out, "*.csv"))):
    name = os.path.basename(f).rsplit("_", 1)[0]
    rows = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gte_kernel<0" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in rows.items():
        v = [x for _, x in sorted(v)][-16:]
        vals[c][name] = sum(v) / len(v)
names = sorted({n for d in vals.values() for n in d})
lines = ["counter".ljust(34) + "".join(n[-18:].rjust(20) for n in names)]
for c in sorted(vals):
    lines.append(c.ljust(34) + "".join(f"{vals[c].get(n, float('nan')):20.4g}" for n in names))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
