#!/bin/bash
# A/B of two PREBUILT libraries (scratch/ab/libA.so, libB.so: e.g. HEAD vs the working tree) on one box, alternating
cd $GRAFT_REPO_ROOT
cp avoid_mpc_amd/libavoid_mpc_amd.so /tmp/lib_keep.so
for r in $(seq 1 ${REPS:-3}); do
  for v in A B; do
    cp scratch/ab/lib$v.so avoid_mpc_amd/libavoid_mpc_amd.so
    echo "$v: $(python tools/experiments/ms_parts.py 2>/dev/null | grep solve-only | sed 's/.*-> //') | bench $(python bench.py --steps 512 --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
  done
done
cp /tmp/lib_keep.so avoid_mpc_amd/libavoid_mpc_amd.so
