#!/bin/bash
# Does the shader clock drop under the fp64 solve load?  Samples rocm-smi while solve-only launches saturate the chip.
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
echo "--- under load"
( AMK_REPS=12000 python tools/experiments/solve_rate.py 2>/dev/null | grep solve-only ) &
BG=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|power (W)" | head -3; sleep 2; done
wait $BG
