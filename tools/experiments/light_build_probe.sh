#!/bin/bash
# Would a build block that FITS beside eight resident solve waves (256 threads = one wave per SIMD at <= 152 registers, <= 18 KB of LDS)
# make progress there?  Probe with the existing source: 256 threads x 16 points (the tile stays 4096 points: same index), 768 staged
# records -- the compiler caps it at 128 registers and spills 240 B/lane, so the build alone is slow; what matters is build + solves.
cd $GRAFT_REPO_ROOT
for cfg in "" "-DAMK_BUILD_THREADS=256 -DAMK_TILE_P=16 -DAMK_STAGE_RECORDS=768"; do
  AMK_HIPCC_FLAGS="$cfg" python -m avoid_mpc_amd.build --force > /dev/null 2>&1
  echo "== build flags: [$cfg]"
  AMK_REPS=128 python tools/experiments/ms_parts.py 2>/dev/null | grep -E "build-only|build\+3"
  python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cold', round(d['value']))"
  python bench.py --workload flight --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flight', round(d['value']))"
done
python -m avoid_mpc_amd.build --force > /dev/null 2>&1
