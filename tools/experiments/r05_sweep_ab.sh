# The keyframe sweep in the closed loop, both regimes (A: 50 k-point frames, --keyframes 3; B: the yaml configuration), and the
# single-stream kernel time of regime A.  Used for the A/Bs of round 5 (profiles/r05_sweep_order_ab.txt, r05_sweep_mlp.txt).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05sweep; mkdir -p $O; : > $O/err.txt
[ -z "$SKIPTESTS" ] && timeout 900 python -m pytest tests/test_keyframe_gpu.py tests/test_kfmap_gpu.py -x -q 2>&1 | tail -4
A="--workload flight --keyframes 3 --no-parity --no-cpu-baseline --streams 10 --gang 2"
B="--workload flight --keyframes 100 --points 3072 --T 1.0 --K 3 --no-parity --no-cpu-baseline --periods 120 --streams 12 --gang 4"
for v in 1 2; do
  timeout 600 python bench.py $A > $O/A.json 2>> $O/err.txt
  timeout 600 python bench.py $B > $O/B.json 2>> $O/err.txt
  python - <<PY
import json
for r in "AB":
    d = json.loads([l for l in open("$O/%s.json" % r).read().splitlines() if l.startswith("{")][-1])
    print("run $v regime", r, d["value"], d["flight"]["x_final_mean_m"], d["flight"]["solves_per_step"])
PY
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py $A --streams 1 > /dev/null 2>> $O/err.txt
db=$(find $O/kt -name "*.db" | head -1); python tools/rocprof_summary.py $db | grep -v "at::native\|Cijk\|rocprim" | head -12 | tee $O/kernel_stats_A.md
find $O -name "*.db" -size +6M -delete; find $O -name "*.csv" -size +4M -delete
tail -3 $O/err.txt
