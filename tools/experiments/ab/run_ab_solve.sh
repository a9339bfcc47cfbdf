#!/bin/bash
L=avoid_mpc_amd/libavoid_mpc_amd.so
for rep in 1 2; do for W in old new; do cp tools/experiments/ab/lib_$W.so $L; echo "$W: $(AMK_REPS=256 python tools/experiments/solve_rate.py 2>/dev/null | grep -o 'solve-only.*solves/us')"; done; done
cp tools/experiments/ab/lib_new.so $L
