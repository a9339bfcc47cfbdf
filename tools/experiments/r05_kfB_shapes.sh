# launch shapes of the yaml-configuration flights with the keyframe map: slots x frames per launch x queue depth
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05kfB_shapes; mkdir -p $O; : > $O/err.txt
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120"
for shape in "12 4 2" "16 4 2" "24 2 2" "12 8 2" "12 4 4" "6 8 2" "20 4 2" "12 2 2"; do
  set -- $shape
  timeout 600 python bench.py $B --streams $1 --gang $2 --queue-depth $3 > $O/s.json 2>> $O/err.txt
  python - <<PY
import json
d = json.loads([l for l in open("$O/s.json").read().splitlines() if l.startswith("{")][-1])
print("slots $1 gang $2 depth $3:", d["value"], "host ms/frame", d["config"]["host_submit_ms_per_step"])
PY
done
tail -2 $O/err.txt
