#!/bin/bash
# flight workload under iteration caps (the first script's --ipm-max-iter did not reach the flight pipeline), and the split of
# a flight period's wall time: AMK_BENCH_SKIP-style diagnostics are not wired into TASK mode, so the caps are the probe
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05a; mkdir -p $out
val() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('config',{}); f=d.get('flight',{})
print(d['value'], 'solves/step', f.get('solves_per_step'), 'it/step', f.get('ipm_iters_per_step'), 'capped', f.get('capped_solves'), 'submit_ms', c.get('host_submit_ms_per_step'))"; }
for cap in 100 30 12 6 1; do
  echo "flight cap $cap: $(python bench.py --workload flight --no-parity --no-cpu-baseline --ipm-max-iter $cap 2>/dev/null | val)"
done | tee $out/cap_flight.txt
for g in 4 16; do
  echo "flight gang $g cap 100: $(python bench.py --workload flight --gang $g --no-parity --no-cpu-baseline 2>/dev/null | val)"
  echo "flight gang $g cap 6: $(python bench.py --workload flight --gang $g --no-parity --no-cpu-baseline --ipm-max-iter 6 2>/dev/null | val)"
done | tee -a $out/cap_flight.txt
