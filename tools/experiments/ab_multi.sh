#!/bin/bash
# A/B/C/... of PREBUILT libraries scratch/ab/lib<V>.so on one box, alternating: driver-style burst, steady state, lone build launch
cd $GRAFT_REPO_ROOT
cp avoid_mpc_amd/libavoid_mpc_amd.so /tmp/lib_keep.so
for r in $(seq 1 ${REPS:-3}); do
  for v in ${VARIANTS:-A B}; do
    cp scratch/ab/lib$v.so avoid_mpc_amd/libavoid_mpc_amd.so
    echo "$v: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_single_stream']; print('burst', d['value'], 'steady', d['value_steady_state'], 'build_us', k['kd_build_kernel']['avg_launch_us'], 'solve_us', k['mpc_solve_kernel']['avg_launch_us'], 'knn_us', k['step_knn_grid_kernel']['avg_launch_us'])")"
  done
done
cp /tmp/lib_keep.so avoid_mpc_amd/libavoid_mpc_amd.so
