# keyframe flights on sensor-like frames (nothing behind the vehicle) against 360-degree clouds: rate, map size, parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05kfs; mkdir -p $O; : > $O/err.txt
run() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/s.json 2>> $O/err.txt; python - <<PY
import json
d = json.loads([l for l in open("$O/s.json").read().splitlines() if l.startswith("{")][-1])
p = (d.get("parity") or {}).get("flights_vs_cpu_oracle") or {}
print("$tag:", d["value"], "solves/step", d["flight"]["solves_per_step"], "x_final", d["flight"]["x_final_mean_m"], "through", d["flight"]["flights_through_a_cylinder"], "| parity", {k: p.get(k) for k in ("separated", "dpos_max_while_flags_agree_m", "dpos_final_max_of_separated_m", "ok")})
PY
}
run "B sensor-like" --workload flight --config yaml --keyframes 100
run "B 360 cloud  " --workload flight --config yaml --keyframes 100 --frames-behind 6 --no-parity --no-cpu-baseline
run "B sensor 16x4" --workload flight --config yaml --keyframes 100 --streams 16 --no-parity --no-cpu-baseline
run "B0 sensor-like frames, no map" --workload flight --config yaml --frames-behind -1 --no-parity --no-cpu-baseline
run "A sensor-like" --workload flight --keyframes 3 --streams 10 --gang 2
run "A 360 cloud  " --workload flight --keyframes 3 --streams 10 --gang 2 --frames-behind 6 --no-parity --no-cpu-baseline
tail -2 $O/err.txt
