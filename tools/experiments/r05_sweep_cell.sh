# cell edge of the sweep's hashed grid (in units of th): regime A flights on sensor-like frames + the mark / build kernel times
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05cell; mkdir -p $O; : > $O/err.txt
for c in "$@"; do
  AMK_SWEEP_CELL=$c timeout 900 python bench.py --workload flight --keyframes 3 --streams 10 --gang 2 --no-parity --no-cpu-baseline > $O/A.json 2>> $O/err.txt
  rm -rf $O/kt; AMK_SWEEP_CELL=$c timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --workload flight --keyframes 3 --streams 1 --gang 2 --periods 30 --no-parity --no-cpu-baseline > /dev/null 2>> $O/err.txt
  db=$(find $O/kt -name "*.db" | head -1)
  python - <<PY
import json
d = json.loads([l for l in open("$O/A.json").read().splitlines() if l.startswith("{")][-1])
print("cell $c th: regime A", d["value"])
PY
  python tools/rocprof_summary.py $db | grep "sweep_mark_hash\|hash_build" | cut -c1-110
done
rm -rf $O/kt
