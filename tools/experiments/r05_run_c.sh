#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05c; mkdir -p $out
python tools/experiments/density_sweep.py 2>&1 | grep -v amdgpu.ids | tee $out/density_sweep.txt
for kfn in 0 3; do
  python bench.py --workload flight --streams 10 --gang 2 --keyframes $kfn 2>$out/flight_kf$kfn.err | tail -1 > $out/flight_kf$kfn.json
  python -c "
import json; d=json.load(open('$out/flight_kf$kfn.json')); print('flight keyframes $kfn (10 x gang 2):', d['value'], d['flight'], d['parity'])"
done 2>&1 | tee $out/flight_keyframes.txt
