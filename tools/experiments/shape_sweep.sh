#!/bin/bash
# pipeline shapes re-swept on the 180-register solve (round 6, last session): steady state (2048 steps) and the 20-step burst
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d.get('value_steady_state') or 0))"; }
for sh in "8 4" "10 4" "12 4" "14 4" "10 5" "10 6" "8 6" "12 3"; do set -- $sh
  echo "streams $1 gang $2: steady $(run --streams $1 --gang $2 --steady-steps 0) | burst $(run --streams $1 --gang $2 --steps 20 --warmup 5 --steady-steps 0)"
done
