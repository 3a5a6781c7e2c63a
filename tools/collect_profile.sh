#!/bin/bash
# Collect the judged artefacts for the headline config on the MI355X box (run through gpurun):
#   bash tools/collect_profile.sh            -> gpurun_out/prof/*
# then, back in the build container:  python tools/summarize_profile.py r01_final
# Passes are separate, as the MI355X guide prescribes: one --kernel-trace --stats pass, and one
# --pmc pass per counter group with no trace domain next to it.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "[collect] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
echo "[collect] kernel trace done"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$tag" -o pmc -- \
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline > "$OUT/pmc_$tag.json" 2> "$OUT/pmc_$tag.err"
  echo "[collect] pmc $c done"
done
find "$OUT" -name "*.csv" | head -40
