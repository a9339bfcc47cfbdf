#!/bin/bash
# sensitivity of the step throughput to the LDS request of the solve (co-resident solve waves per CU and the headroom left
# for the small kernels of other streams); LDS_LIST="0 21000 ..." bytes
cd $GRAFT_REPO_ROOT
for lds in ${LDS_LIST:-0 21000 23400 27000 32768}; do
  AMK_HIPCC_FLAGS="-DAMK_SOLVE_LDS_MIN=$lds" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  for sk in "" build; do
  AMK_BENCH_SKIP=$sk python bench.py --steps 256 --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lds_min $lds skip=$sk value',d['value'],'solve ms',d['roofline']['avg_launch_ms'])"
  done
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
