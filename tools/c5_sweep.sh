# A/B of geometry / store policy / processing order for config 5 (bench.py --workload c5), one box
for cfg in "" "--affinity -1" "--affinity -1 --epw 7" "--affinity -1 --epw 8" "--affinity -1 --epw 9" "--affinity -1 --epw 10" "--affinity -1 --epw 12" "--affinity -1 --epw 8 --nt 1" "--affinity -1 --epw 8 --nt 0" "--affinity -1 --epw 10 --nt 0"; do
  python3 bench.py --workload c5 --no-cpu-baseline --no-pmc --steps 1000 $cfg 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5 $cfg ->', round(d['roofline']['kernel_us_per_launch'],2), 'us', d['config']['launch']['envs_per_wave'], d['config']['launch']['obs_stores'], d['config']['launch']['resident_workgroups_per_cu'])"
done
