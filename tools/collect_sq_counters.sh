#!/bin/bash
# One SQ-block counter pass (8 slots) for the step kernel (config 3, episodes out of phase) and the fused rollout
# kernels: where the waves' cycles go.  bash tools/collect_sq_counters.sh -> gpurun_out/sq/summary.txt
set -e
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq
rm -rf "$OUT" && mkdir -p "$OUT"
export TMPDIR=/tmp
C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
(cd /tmp && rocprofv3 --pmc $C --output-format csv -d /tmp/sq_step -o pmc -- python3 $ROOT/bench.py --pmc-child --steps 40 --warmup 10 > /dev/null 2> "$OUT/step.err")
cp $(find /tmp/sq_step -name "*counter_collection.csv" | head -1) "$OUT/step.csv"
(cd /tmp && rocprofv3 --pmc $C --output-format csv -d /tmp/sq_roll -o pmc -- python3 $ROOT/tools/rollout_bench.py --k 64 --reps 2 --desync > /dev/null 2> "$OUT/roll.err")
cp $(find /tmp/sq_roll -name "*counter_collection.csv" | head -1) "$OUT/roll.csv"
python3 - "$OUT" <<'PY'
import csv, sys, os
from collections import defaultdict
out = sys.argv[1]
lines = []
for f, pick in (("step.csv", ["gte_kernel<0"]), ("roll.csv", ["gte_rollout_resident", "gte_rollout_state"])):
    rows = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(os.path.join(out, f))):
        for p in pick:
            if p in r["Kernel_Name"]:
                rows[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in rows.items():
        m = {n: sum(v[-20:]) / len(v[-20:]) for n, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        lines.append(f"{k}\n   waves {m.get('SQ_WAVES', 0):.0f}  wave-cycles {wc:.3e}  waiting (s_waitcnt/barrier) "
                     f"{100 * m.get('SQ_WAIT_ANY', 0) / wc:.1f} %  issue-stalled {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} %  "
                     f"issuing {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} %  VALU insts {m.get('SQ_INSTS_VALU', 0):.3e}  "
                     f"LDS bank-conflict cycles {m.get('SQ_LDS_BANK_CONFLICT', 0):.3e}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
