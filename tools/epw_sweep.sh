# envs-per-wavefront sweep of the step kernel with episodes out of phase (python3 bench.py), one box
for n in 65536 262144; do
  for e in 0 8 9 10 11 12 13 14 15 16; do
    python3 bench.py --envs $n --epw $e --no-cpu-baseline --no-pmc --no-hbm-regime --steps 600 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); l=d['config']['launch']; print('envs $n epw $e ->', round(d['roofline']['kernel_us_per_launch'],2), 'us', l['envs_per_wave'], l['n_blocks'], l['obs_stores'], l['resident_workgroups_per_cu'])"
  done
done
