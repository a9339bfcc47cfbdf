#!/bin/bash
# the default bench line (and the build-only / step-only diagnostics) for a list of compile-time variants: FLAGS_LIST="a|b|c"
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VARS <<< "${FLAGS_LIST:-|-DAMK_GRID_UNROLL=8}"
for fl in "${VARS[@]}"; do
  AMK_HIPCC_FLAGS="$fl" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  r=""
  for sk in "" build step; do
    r="$r $(AMK_BENCH_SKIP=$sk python bench.py --steps 256 --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
  done
  echo "flags [$fl]: ms/step full / no builds / builds only:$r"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
