#!/bin/bash
# Is the 20-step burst slow because the shader clock is still ramping?  Same timed region (frames 5 .. 24 of the 40 in flight: the
# bench cycles through its frames, so W = 5 mod 40 times the same scenes) after 5 / 45 / 405 / 2005 warm-up steps; then the same
# burst on other frame sets (W = 0, 10, 20, 30), and the solve-only rate against the length of the measurement.
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d.get('value_steady_state') or 0))"; }
for W in 5 45 405 2005; do echo "burst 20 steps after $W warm-up steps (frames 5..24): $(run --steps 20 --warmup $W) | $(run --steps 20 --warmup $W)"; done
for W in 0 10 20 30; do echo "burst 20 steps after $W warm-up steps (frames $W..): $(run --steps 20 --warmup $W)"; done
for R in 2 8 32 128 512 2048; do echo "solve-only, $R x 16 launches timed: $(AMK_REPS=$R python tools/experiments/solve_rate.py 2>/dev/null | grep -o 'solve-only.*solves/us')"; done
