# observation store policy per workload: 0 plain, 1 non-temporal, 2 sc1 (-1 = the library's choice); us per step
WORKLOADS=${1:-"c5"}
for w in $WORKLOADS; do for r in 1 2; do for nt in -1 0 1 2; do echo -n "$w nt $nt: "; python3 bench.py --workload $w --nt $nt --steps 600 --warmup 100 --no-pmc --no-cpu-baseline --no-hbm-regime 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), d['config']['launch']['obs_stores'])"; done; done; done
