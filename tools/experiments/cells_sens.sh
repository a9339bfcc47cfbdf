#!/bin/bash
# grid resolution of the bucketed index (target points per cell, cap on the number of cells) against the step throughput:
# fewer, larger cells make the build's scatter more local (fewer open lines per scene in L2) and the search's candidate
# lists longer.  VARIANTS="ppc:maxcells ..."
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-8:8192 32:2048 64:1024 128:1024 200:512}; do
  ppc=${v%%:*}; c=${v##*:}
  AMK_HIPCC_FLAGS="-DAMK_GRID_MAX_CELLS=$c -DAMK_GRID_PPC=$ppc" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  r=""
  for args in ${BENCH_LIST:-"--steps 256" "--points 200000 --T 1.0 --steps 64 --warmup 4 --streams 8" "--points 5000 --T 0.33 --K 3 --steps 256" "--points 3072 --T 1.0 --K 3 --steps 128"}; do
    r="$r $(python bench.py $args --no-cpu-baseline --no-parity --steady-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1000,1))")"
  done
  echo "ppc $ppc maxcells $c: k steps/s at C2 / C5 / C1 / yaml sizes:$r"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
