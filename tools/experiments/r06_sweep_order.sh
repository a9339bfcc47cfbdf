# the hashed sweep with the keyframe's points in last sweep's grid order (AMK_SWEEP_ORDER=1, shipped) against record order (0):
# flags (tests), sensor-like flights of regime A with parity, mark / build kernel time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06order; mkdir -p $O; : > $O/err.txt
[ -z "$SKIPTESTS" ] && timeout 900 python -m pytest tests/test_keyframe_gpu.py tests/test_kfmap_gpu.py -x -q -m gpu 2>&1 | tail -3
for ord in ${ORDERS:-0 1}; do
  export AMK_SWEEP_ORDER=$ord
  timeout 900 python bench.py --workload flight --keyframes 3 --streams 10 --gang 2 > $O/A$ord.json 2>> $O/err.txt
  python - <<PY
import json
d = json.loads([l for l in open("$O/A$ord.json").read().splitlines() if l.startswith("{")][-1])
p = (d.get("parity") or {}).get("flights_vs_cpu_oracle") or {}
print("order $ord regime A", d["value"], d["flight"]["x_final_mean_m"], d["flight"]["solves_per_step"], {k: p.get(k) for k in ("separated", "dpos_max_while_flags_agree_m", "ok")})
PY
  rm -rf $O/kt; timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --workload flight --keyframes 3 --streams 1 --gang 2 --periods 30 --no-parity --no-cpu-baseline > /dev/null 2>> $O/err.txt
  db=$(find $O/kt -name "*.db" | head -1); python tools/rocprof_summary.py $db | grep "sweep_mark\|hash_build\|mpc_solve\|compact" | cut -c1-110
done
rm -rf $O/kt; tail -2 $O/err.txt
