#!/bin/bash
# Round 5: the resumable solve.  Parity tests first, then the cold workload (20-step burst as the driver runs it + steady state)
# and the flight workload over budgets x budgeted rounds, one box.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05b; mkdir -p $out
timeout 900 python -m pytest tests/test_mpc_resume_gpu.py tests/test_kd_gpu.py -x -q 2>&1 | tail -15 | tee $out/tests.txt
val() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); f=d.get('flight',{})
print(d['value'], 'steady', d.get('value_steady_state'), 'it/step', c.get('ipm_iters_per_step', f.get('ipm_iters_per_step')))"; }
for cfg in "0 0" "30 0" "24 0" "16 0" "12 0" "8 0" "24 1" "16 1" "16 3" "12 3" "8 4" "6 6"; do
  set -- $cfg
  echo "cold budget $1 rounds $2: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --solve-budget $1 --budget-rounds $2 2>/dev/null | val)"
done | tee $out/budget_cold.txt
for cfg in "0 0" "16 0" "8 0"; do
  set -- $cfg
  echo "flight budget $1 rounds $2: $(python bench.py --workload flight --no-parity --no-cpu-baseline --solve-budget $1 --budget-rounds $2 2>/dev/null | val)"
done | tee $out/budget_flight.txt
