#!/bin/bash
# do builds / searches overlap with the solves of other streams (tools/experiments/ms_overlap.py) for compile-time variants?
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VARS <<< "${FLAGS_LIST:-|-DAMK_BUILD_THREADS=256 -DAMK_SOLVE_LDS_MIN=21000}"
for fl in "${VARS[@]}"; do
  AMK_HIPCC_FLAGS="$fl" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  echo "flags [$fl]"; python tools/experiments/ms_overlap.py 8 2>/dev/null | grep -A1 "solves + builds"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
