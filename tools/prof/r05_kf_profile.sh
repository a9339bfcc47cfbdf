#!/bin/bash
# Kernel time of the closed loop WITH the keyframe map (round 5), two regimes:
#   A  the bench's --keyframes 3 line: 50 k-point frames, 10 slots x 2 frames per launch
#   B  the reference's own configuration: 3072-point frames (640 x 480 / 10), N = 30, K = 3, max_frame_count = 100
# Single-stream traces (clean kernel durations) and the in-flight runs.  Usage (through gpurun): tools/prof/r05_kf_profile.sh <tag>
TAG=${1:-r05kf}
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp; cd $ROOTD
OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/err.txt
A="--workload flight --keyframes 3 --no-parity --no-cpu-baseline"
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline"
timeout 600 python bench.py $A --streams 10 --gang 2 > $OUT/A_10x2.json 2>> $OUT/err.txt
timeout 600 python bench.py $B --streams 4 --gang 2 --periods 120 > $OUT/B_4x2.json 2>> $OUT/err.txt
timeout 600 python bench.py $B --streams 8 --gang 4 --periods 120 > $OUT/B_8x4.json 2>> $OUT/err.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/ktA -o kt -- python bench.py $A --streams 1 --gang 2 --periods 24 > /dev/null 2>> $OUT/err.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/ktB -o kt -- python bench.py $B --streams 1 --gang 2 --periods 120 > /dev/null 2>> $OUT/err.txt
for r in A B; do
  db=$(find $OUT/kt$r -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > $OUT/kernel_stats_kf_$r.md
done
find $OUT -name "*.db" -size +6M -delete; find $OUT -name "*.csv" -size +4M -delete
head -30 $OUT/kernel_stats_kf_A.md; head -30 $OUT/kernel_stats_kf_B.md
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f.split("/")[-1], d.get("value"), d["flight"])
    except Exception as e:
        print(f.split("/")[-1], "UNREADABLE", e)
PY
tail -5 $OUT/err.txt
