# chunk (records per wavefront) x step (records per trip) of the persistent-lane sweep: the real pair's mark kernel time
# usage: r05_sweep_lanes_variants.sh "256 4" "1024 4" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05lanesv; mkdir -p $O; : > $O/err.txt
for v in "$@"; do
  set -- $v
  AMK_HIPCC_FLAGS="-DAMK_SWEEP_CHUNK=$1 -DAMK_SWEEP_STEP=$2 $3" python -c "from avoid_mpc_amd import build; build.build(force=True)" >> $O/err.txt 2>&1
  rm -rf $O/swc; timeout 600 rocprofv3 --kernel-trace -d $O/swc -o kt -- python tools/experiments/sweep_classes.py > /dev/null 2>> $O/err.txt
  python - <<PY
import json
d=json.load(open('avoid_mpc_amd/kernel_resources.json'))
print("chunk $1 step $2 $3:", [(v['vgprs'], v['occupancy']) for k,v in d.items() if 'sweep_mark_lanes' in k])
PY
  python tools/experiments/sweep_classes.py --read $O/swc | cut -c1-100
done
rm -rf $O/swc; tail -2 $O/err.txt
