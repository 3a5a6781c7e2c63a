# A/B of kernel_variant bits in one library: us per step, episodes finished.  usage: variant_ab.sh "0 8192" "c3 c4 c5"
VARIANTS=${1:-"0 8192"}; WORKLOADS=${2:-"c3 c4 c5"}
for w in $WORKLOADS; do for r in 1 2; do for v in $VARIANTS; do echo -n "$w variant $v: "; python3 bench.py --workload $w --variant $v --steps 600 --warmup 100 --no-pmc --no-cpu-baseline --no-hbm-regime 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), d['config']['episodes_finished'])"; done; done; done
