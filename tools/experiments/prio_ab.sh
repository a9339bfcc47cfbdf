#!/bin/bash
O=gpurun_out/prio; mkdir -p $O
for rep in 1 2 3; do
for P in 0 1; do
  AMK_PIPELINE_PRIO=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --steady-steps 0 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('burst prio=$P', d['value'])"
done; done
for P in 0 1; do
  AMK_PIPELINE_PRIO=$P python bench.py --no-cpu-baseline --no-parity 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steady prio=$P', d['value'])"
  AMK_PIPELINE_PRIO=$P python bench.py --workload flight --no-cpu-baseline --no-parity 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flight prio=$P', d['value'])"
done
AMK_PIPELINE_PRIO=1 python tools/experiments/burst_timeline.py 10 4 20 > $O/burst_timeline_prio1.txt 2>>$O/err.txt
head -8 $O/burst_timeline_prio1.txt
tail -5 $O/err.txt
