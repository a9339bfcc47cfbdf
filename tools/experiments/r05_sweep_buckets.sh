cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05nb; mkdir -p $O; : > $O/err.txt
for nb in 16384 8192 4096; do
  AMK_SWEEP_NB=$nb timeout 900 python bench.py --workload flight --keyframes 3 --streams 10 --gang 2 --no-parity --no-cpu-baseline > $O/A.json 2>> $O/err.txt
  rm -rf $O/kt; AMK_SWEEP_NB=$nb timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --workload flight --keyframes 3 --streams 1 --gang 2 --periods 30 --no-parity --no-cpu-baseline > /dev/null 2>> $O/err.txt
  db=$(find $O/kt -name "*.db" | head -1)
  python -c "
import json
d = json.loads([l for l in open('$O/A.json').read().splitlines() if l.startswith('{')][-1]); print('buckets $nb: regime A', d['value'])"
  python tools/rocprof_summary.py $db | grep "sweep_mark_hash\|hash_build" | cut -c1-105
done
rm -rf $O/kt
