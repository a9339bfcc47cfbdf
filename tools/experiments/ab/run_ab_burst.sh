#!/bin/bash
L=avoid_mpc_amd/libavoid_mpc_amd.so
run() { python bench.py "$@" --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d.get('value_steady_state') or 0))"; }
for rep in 1 2 3; do for W in old new; do cp tools/experiments/ab/lib_$W.so $L; echo "$W burst: $(run --steps 20 --warmup 5)"; done; done
cp tools/experiments/ab/lib_new.so $L
