#!/bin/bash
# index build alone (tools/experiments/build_only.py) for a list of compile-time variants
cd $GRAFT_REPO_ROOT
for fl in "" "-DAMK_BUILD_DIAG=1" "-DAMK_BUILD_DIAG=2" "-DAMK_BUILD_THREADS=1024" "-DAMK_BUILD_THREADS=1024 -DAMK_GRID_UNROLL=8" $EXTRA_VARIANTS; do
  AMK_HIPCC_FLAGS="$fl" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  echo "flags [$fl]: $(python tools/experiments/build_only.py 2>/dev/null | tail -1)"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
