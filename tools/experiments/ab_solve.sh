#!/bin/bash
# A/B of two compile-time variants on ONE box: saturated solve-only rate (ms_parts.py) and the default bench, alternating
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/ab
for v in A B; do
  if [ $v = A ]; then fl="$FLAGS_A"; else fl="$FLAGS_B"; fi
  AMK_HIPCC_FLAGS="$fl" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  cp avoid_mpc_amd/libavoid_mpc_amd.so /tmp/ab/lib$v.so
done
for r in $(seq 1 ${REPS:-3}); do
  for v in A B; do
    cp /tmp/ab/lib$v.so avoid_mpc_amd/libavoid_mpc_amd.so
    echo "$v: $(python tools/experiments/ms_parts.py 2>/dev/null | grep solve-only | sed 's/.*-> //') | bench $(python bench.py --steps 512 --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
  done
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
