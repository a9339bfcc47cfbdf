# (history: AMK_SWEEP_VARIANT selected the persistent-lane kernel when this ran; it is a patch now -- tools/experiments/patches/
#  r05_sweep_persistent_lanes.patch -- and the variable is ignored; AMK_SWEEP_TARGET=0 still selects the walk over the frame's own index)
# the pool's sweep: target = the current frame's own index (0) or its fine hashed grid (1); flags (tests), sensor-like flights of both
# regimes, single-stream kernel time of regime A
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05lanes; mkdir -p $O; : > $O/err.txt
[ -z "$SKIPTESTS" ] && timeout 900 python -m pytest tests/test_keyframe_gpu.py tests/test_kfmap_gpu.py -x -q 2>&1 | tail -3
for tg in 1 0; do
  AMK_SWEEP_TARGET=$tg AMK_SWEEP_VARIANT=0 timeout 900 python bench.py --workload flight --keyframes 3 --streams 10 --gang 2 > $O/A.json 2>> $O/err.txt
  AMK_SWEEP_TARGET=$tg AMK_SWEEP_VARIANT=0 timeout 900 python bench.py --workload flight --config yaml --keyframes 100 > $O/B.json 2>> $O/err.txt
  python - <<PY
import json
for r in "AB":
    d = json.loads([l for l in open("$O/%s.json" % r).read().splitlines() if l.startswith("{")][-1])
    p = (d.get("parity") or {}).get("flights_vs_cpu_oracle") or {}
    print("target $tg regime", r, "(sensor-like frames)", d["value"], d["flight"]["x_final_mean_m"], d["flight"]["solves_per_step"], {k: p.get(k) for k in ("separated", "dpos_max_while_flags_agree_m", "ok")})
PY
done
rm -rf $O/kt; timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --workload flight --keyframes 3 --streams 1 --gang 2 --no-parity --no-cpu-baseline > /dev/null 2>> $O/err.txt
db=$(find $O/kt -name "*.db" | head -1); python tools/rocprof_summary.py $db | grep -v "at::native\|Cijk\|rocprim" | head -12 | cut -c1-150
rm -rf $O/kt
tail -2 $O/err.txt
