# potential of a finer L2-affinity sort, without its cost: episodes in phase (no resets inside the timed
# window), the order built once at reset and never rebuilt (--affinity 100000)
for b in 4096 16384 65536 262144; do echo "bins $b"; GTE_AFFINITY_BINS=$b python3 bench.py --sync-episodes --affinity 100000 --steps 300 --warmup 100 --no-pmc --no-cpu-baseline --no-hbm-regime 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_us_per_launch'],2))"; done
