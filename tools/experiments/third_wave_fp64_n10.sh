#!/bin/bash
# What would a third fp64 solve wave per SIMD buy?  At N = 20 the third wave cannot be resident (17.8 KB of LDS per scene: 8 per CU);
# at N = 10 the LDS allows it, so the SAME fp64 source compiled for 2 and for 3 waves per SIMD shows what occupancy is worth in fp64
# (the spills of the 168-register build are part of the price).  Round 6, last session.  Sustained rates (256 x 16 launches).
cd $GRAFT_REPO_ROOT
export AMK_REPS=256
one() {  # $1 waves, $2 scheduler strategy
  AMK_SCHED_STRATEGY=$2 AMK_HIPCC_FLAGS="-DAMK_SOLVE_WAVES=$1" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  for N in 10 20; do
  res=$(python -c "
import json; r=json.load(open('avoid_mpc_amd/kernel_resources.json'))
k=[v for n,v in r.items() if 'mpc_solve_kernelILi${N}E' in n][0]; print('vgprs', k['vgprs'], 'scratch B/lane', k['scratch_bytes_per_lane'])")
  T=$(python -c "print({10: 0.33, 20: 0.66}[$N])")
  echo "fp64 N=$N kernel, $1 waves per SIMD, $2 ($res): $(AMK_T=$T AMK_K=8 python tools/experiments/solve_rate.py 2>/dev/null | grep -o 'solve-only.*status')"
  done
}
one 2 max-ilp
one 3 max-ilp
one 3 iterative-minreg
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
