#!/bin/bash
# classic (512 threads, 70 KB LDS) against light (256 threads, ~5 KB LDS) index-build block shape, same library, run-time switch
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d.get('value_steady_state') or 0))"; }
for rep in 1 2; do for SH in classic light; do export AMK_BUILD_SHAPE=$SH
  echo "== $SH: $(AMK_REPS=128 python tools/experiments/ms_parts.py 2>/dev/null | grep -E 'build-only|build\+3' | tr '\n' ' ')"
  echo "$SH cold $(run) | burst $(run --steps 20 --warmup 5) | flight $(run --workload flight) | yaml kf100 $(run --workload flight --config yaml --keyframes 100) | kf3 $(run --workload flight --streams 10 --gang 2 --keyframes 3)"
done; done
AMK_BUILD_SHAPE=light timeout 600 python -m pytest tests/test_kd_gpu.py tests/test_step_gpu.py tests/test_kfmap_gpu.py -x -q 2>&1 | tail -2
