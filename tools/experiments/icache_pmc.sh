#!/bin/bash
# Does the 64 KB solve kernel (65 760 bytes of code, mpc_solve_kernel<20>) miss in the instruction cache?  (round 6, session 4)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/icache; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_TC_INST[A-Z_]*\|SQ_WAVE[A-Z_]*IFETCH[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt
cat $O/names.txt
for G in 1 4; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/g$G -o p -- python bench.py --steps 8 --warmup 2 --streams 1 --gang $G --no-cpu-baseline --no-parity --steady-steps 0 > /dev/null 2>> $O/err.txt
done
python - <<'P'
import csv, glob, collections
for G in (1, 4):
    fs = glob.glob(f'gpurun_out/icache/g{G}/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:40]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQC_ICACHE_REQ': n[k] += 1
    print('gang', G)
    for k, v in acc.items():
        if n[k]: print(' ', k, 'launches', n[k], {c: round(x / n[k]) for c, x in v.items()})
P
tail -3 $O/err.txt
