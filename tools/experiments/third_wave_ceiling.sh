#!/bin/bash
# Round 5 (review item 5b): what would a third solve wave per SIMD buy?  The fp64 kernel cannot have one (226-230 VGPRs, 19.8 KB of LDS
# per scene: DESIGN.md section 5), but the fp32 build of the SAME algorithm can: half the LDS (16 scenes per CU by LDS) and 171-174
# VGPRs, i.e. 168 with -DAMK_SOLVE_WAVES=3.  Saturated solve-only rate (tools/experiments/ms_parts.py, 16 streams x 256 scenes) of the
# fp32 kernel compiled for 2 / 3 / 4 waves per SIMD = 8 / 12 / 16 scenes per CU: how this algorithm's throughput scales with occupancy
# on this chip when the memories allow it.
cd $GRAFT_REPO_ROOT
for w in 2 3 4; do
  AMK_HIPCC_FLAGS="-DAMK_SOLVE_WAVES=$w" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  res=$(python -c "
import json; r=json.load(open('avoid_mpc_amd/kernel_resources.json'))
k=[v for n,v in r.items() if 'mpc_solve_kernel_f32ILi20' in n][0]; print('vgprs', k['vgprs'], 'scratch B/lane', k['scratch_bytes_per_lane'], 'occupancy', k['occupancy'])")
  echo "fp32 kernel compiled for $w waves per SIMD ($res): $(AMK_PREC=32 python tools/experiments/ms_parts.py 2>/dev/null | grep solve-only)"
done
AMK_HIPCC_FLAGS="-DAMK_SOLVE_WAVES=3" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
res=$(python -c "
import json; r=json.load(open('avoid_mpc_amd/kernel_resources.json'))
k=[v for n,v in r.items() if 'mpc_solve_kernelILi20' in n][0]; print('vgprs', k['vgprs'], 'scratch B/lane', k['scratch_bytes_per_lane'], 'occupancy', k['occupancy'])")
echo "fp64 kernel compiled for 3 waves per SIMD ($res; LDS still allows 8 scenes per CU): $(python tools/experiments/ms_parts.py 2>/dev/null | grep solve-only)"
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
echo "fp64 kernel as shipped: $(python tools/experiments/ms_parts.py 2>/dev/null | grep solve-only)"
