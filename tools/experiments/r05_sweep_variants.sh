# kSweepTiles x kSweepRecs of the sweep thread's walk (kd_grid.h: grid_outlier_thread), built on the box, regime A flights +
# single-stream kernel time.  Usage: r05_sweep_variants.sh "8 1" "16 1" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05sweepv; mkdir -p $O; : > $O/err.txt
A="--workload flight --keyframes 3 --no-parity --no-cpu-baseline --streams 10 --gang 2"
for v in "$@"; do
  set -- $v; T=$1; R=$2
  AMK_HIPCC_FLAGS="-DAMK_SWEEP_TILES=$T -DAMK_SWEEP_RECS=$R" python -c "from avoid_mpc_amd import build; build.build(force=True)" >> $O/err.txt 2>&1
  python - <<PY
import json
d=json.load(open('avoid_mpc_amd/kernel_resources.json'))
print("tiles $T recs $R:", [v for k,v in d.items() if 'sweep_mark' in k])
PY
  timeout 600 python bench.py $A > $O/A.json 2>> $O/err.txt
  python - <<PY
import json
d = json.loads([l for l in open("$O/A.json").read().splitlines() if l.startswith("{")][-1])
print("tiles $T recs $R regime A", d["value"], d["flight"]["x_final_mean_m"])
PY
  rm -rf $O/kt; timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py $A --streams 1 > /dev/null 2>> $O/err.txt
  db=$(find $O/kt -name "*.db" | head -1); python tools/rocprof_summary.py $db | grep "sweep_mark"
done
rm -rf $O/kt
tail -2 $O/err.txt
