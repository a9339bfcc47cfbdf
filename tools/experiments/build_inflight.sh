#!/bin/bash
# builds only, 20 in flight (bench.py AMK_BENCH_SKIP=step), for build-kernel ablations: what the scatter costs IN FLIGHT
cd $GRAFT_REPO_ROOT
for fl in "" "-DAMK_BUILD_DIAG=1" "-DAMK_BUILD_DIAG=2" $EXTRA; do
  AMK_HIPCC_FLAGS="$fl" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  echo "flags [$fl]: builds only, ms per step: $(AMK_BENCH_SKIP=step python bench.py --steps 256 --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
