# what bounds the yaml-configuration flights with the keyframe map?  the same loop with fewer flights per frame (GPU-bound: steps/s
# stay, frames/s double; host-bound: frames/s stay) and with the solve capped at one iteration (GPU work down, host work equal)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05kfB_bound; mkdir -p $O; : > $O/err.txt
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120 --streams 12 --gang 4"
B0="--workload flight --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120 --streams 12 --gang 4"
run() { tag=$1; shift; timeout 600 python bench.py "$@" > $O/s.json 2>> $O/err.txt; python - <<PY
import json
d = json.loads([l for l in open("$O/s.json").read().splitlines() if l.startswith("{")][-1])
print("$tag:", d["value"], "steps/s =", round(d["value"] / d["config"]["scenes_per_gpu"]), "frames/s; host ms/frame", d["config"]["host_submit_ms_per_step"], "iters/step", d["flight"]["ipm_iters_per_step"])
PY
}
run "B  256 flights/frame" $B
run "B  128 flights/frame" $B --scenes 128
run "B   64 flights/frame" $B --scenes 64
run "B  256, solve capped at 1 iteration" $B --ipm-max-iter 1
run "B0 256 flights/frame" $B0
run "B0 128 flights/frame" $B0 --scenes 128
run "B0 256, solve capped at 1 iteration" $B0 --ipm-max-iter 1
tail -2 $O/err.txt
