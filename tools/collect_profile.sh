#!/bin/bash
# Collect the judged artefacts on the MI355X box (run through gpurun):
#   bash tools/collect_profile.sh [workload=c3] [extra bench flags]   -> gpurun_out/prof/*
# then, back in the build container:  python tools/summarize_profile.py r02 [workload]
# Sizes: the workload's own (c3: 65 536 envs, observations resident in the Infinity Cache) and,
# for c3, 262 144 envs (671 MB of observations per launch: the HBM regime).
# Passes are separate, as the MI355X guide prescribes: one --kernel-trace --stats pass, and one
# --pmc pass per counter group with no trace domain next to it; the program itself follows `--`.
set -e
cd "$(dirname "$0")/.."
WL=${1:-c3}
shift || true
OUT=gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --workload $WL "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "[collect] bench done"
SIZES="0"
if [ "$WL" = "c3" ]; then SIZES="0 262144"; fi
for n in $SIZES; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$n" -o trace -- \
    python3 bench.py --workload $WL --envs $n --steps 1000 --warmup 100 --no-cpu-baseline --no-pmc \
    --no-hbm-regime "$@" > "$OUT/bench_under_rocprof_$n.json" 2> "$OUT/trace_$n.err"
  echo "[collect] kernel trace (envs $n) done"
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${n}_$tag" -o pmc -- \
      python3 bench.py --workload $WL --envs $n --pmc-child --steps 40 --warmup 10 "$@" \
      > /dev/null 2> "$OUT/pmc_${n}_$tag.err"
    echo "[collect] pmc $c (envs $n) done"
  done
done
find "$OUT" -name "*.csv" | head -40
