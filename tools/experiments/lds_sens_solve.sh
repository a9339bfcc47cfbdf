#!/bin/bash
# saturated solve-only rate (16 streams x 256 scenes, tools/experiments/ms_parts.py) against the number of co-resident
# solve waves per CU, forced down by inflating the LDS request
cd $GRAFT_REPO_ROOT
for lds in 0 32768 40960 53248 81920; do
  AMK_HIPCC_FLAGS="-DAMK_SOLVE_LDS_MIN=$lds" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  echo "lds_min $lds: $(python tools/experiments/ms_parts.py 2>/dev/null | grep solve-only)"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
