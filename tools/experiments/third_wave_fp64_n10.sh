#!/bin/bash
# What would a third fp64 solve wave per SIMD buy?  At N = 20 the third wave cannot be resident (17.8 KB of LDS per scene: 8 per CU);
# at N = 10 the LDS allows it, so the SAME fp64 source compiled for 2 and for 3 waves per SIMD shows what occupancy is worth in fp64
# (the spills of the 168-register build are part of the price).  Round 6, last session.
cd $GRAFT_REPO_ROOT
for w in 2 3; do
  AMK_HIPCC_FLAGS="-DAMK_SOLVE_WAVES=$w" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  res=$(python -c "
import json; r=json.load(open('avoid_mpc_amd/kernel_resources.json'))
k=[v for n,v in r.items() if 'mpc_solve_kernelILi10' in n][0]; print('vgprs', k['vgprs'], 'scratch B/lane', k['scratch_bytes_per_lane'], 'occupancy', k['occupancy'])")
  for K in 8 3; do
    echo "fp64 N=10 kernel compiled for $w waves per SIMD ($res): $(AMK_T=0.33 AMK_K=$K python tools/experiments/solve_rate.py 2>/dev/null | grep solve-only)"
  done
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
