#!/bin/bash
# same-box A/B of two builds of the library: tools/experiments/ab/lib_old.so vs lib_new.so (both travel with the snapshot)
L=avoid_mpc_amd/libavoid_mpc_amd.so
run() { python bench.py "$@" --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('value_steady_state'))"; }
for rep in 1 2; do
for W in old new; do
  cp tools/experiments/ab/lib_$W.so $L
  echo "$W cold: $(run)"
  echo "$W flight: $(run --workload flight)"
  echo "$W flight yaml kf100: $(run --workload flight --config yaml --keyframes 100)"
  echo "$W flight kf3: $(run --workload flight --streams 10 --gang 2 --keyframes 3)"
done; done
cp tools/experiments/ab/lib_new.so $L
