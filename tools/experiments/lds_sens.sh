#!/bin/bash
# sensitivity of the step throughput to the number of co-resident solve waves per CU (LDS request forced up)
cd $GRAFT_REPO_ROOT
for lds in 0 32768 40960 53248 81920; do
  AMK_HIPCC_FLAGS="-DAMK_SOLVE_LDS_MIN=$lds" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  python bench.py --steps 256 --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lds_min $lds value',d['value'],'solve ms',d['roofline']['avg_launch_ms'])"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
