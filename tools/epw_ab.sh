# envs per wave (launch geometry) A/B on one box: us per step.  usage: epw_ab.sh "0 8 16 32" "c3"
EPWS=${1:-"0 8 16 32"}; WORKLOADS=${2:-"c3"}
for w in $WORKLOADS; do for r in 1 2; do for v in $EPWS; do echo -n "$w epw $v: "; python3 bench.py --workload $w --epw $v --steps 600 --warmup 100 --no-pmc --no-cpu-baseline --no-hbm-regime 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), d['config'].get('launch'))"; done; done; done
