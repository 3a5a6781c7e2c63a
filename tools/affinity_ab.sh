# L2-affinity order on (0 = default period) / off (-1) per workload: us per step
WORKLOADS=${1:-"c2"}
for w in $WORKLOADS; do for r in 1 2 3; do for a in 0 -1; do echo -n "$w affinity $a: "; python3 bench.py --workload $w --affinity $a --steps 2000 --warmup 200 --no-pmc --no-cpu-baseline --no-hbm-regime 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_us_per_launch'],2))"; done; done; done
