#!/bin/bash
# Round 5, before building the resumable solve: what is there to win?  A solve with iteration cap C is what the MAIN launch of
# a budgeted solve looks like when the stragglers' continuation costs nothing (results are garbage for the capped scenes; the
# lines are diagnostics).  Cold workload (20-step burst + steady state) and the flight workload, caps 100 (shipped) ... 6.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05a; mkdir -p $out
val() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); f=d.get('flight',{})
print(d['value'], 'steady', d.get('value_steady_state'), 'solves/step', c.get('solves_per_step', f.get('solves_per_step')), 'it/step', c.get('ipm_iters_per_step', f.get('ipm_iters_per_step')), 'capped', f.get('capped_solves'), 'submit_ms', c.get('host_submit_ms_per_step'))"; }
for cap in 100 30 16 10 6; do
  echo "cold burst cap $cap: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --ipm-max-iter $cap 2>/dev/null | val)"
done | tee $out/cap_cold.txt
for cap in 100 30 16 10 6; do
  echo "flight cap $cap: $(python bench.py --workload flight --no-parity --no-cpu-baseline --ipm-max-iter $cap 2>/dev/null | val)"
done | tee $out/cap_flight.txt
