#!/bin/bash
cd $GRAFT_REPO_ROOT
cp avoid_mpc_amd/libavoid_mpc_amd.so /tmp/lib_keep.so
for r in $(seq 1 ${REPS:-3}); do
  for v in A B; do
    cp scratch/ab/lib$v.so avoid_mpc_amd/libavoid_mpc_amd.so
    for st in 20 2048; do
    echo "$v steps $st: $(python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernels_single_stream']['kd_build_kernel']['avg_launch_us'])")"
    done
  done
done
cp /tmp/lib_keep.so avoid_mpc_amd/libavoid_mpc_amd.so
