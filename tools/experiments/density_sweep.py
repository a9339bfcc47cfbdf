"""Round 5: how much of "99 % of the flights dip inside the drone radius" is the harness's obstacle density?
Closed-loop flights on the GPU (tests/_flight.py: gpu_flights, TASK mode) over cylinder densities, at BASELINE configs[1]'s
horizon (N = 20, T = 0.66) and at the reference's own (N = 30, T = 1.0, mpc_parameters.yaml:1-2), 3072-point frames (the
reference's frame size), 64 flights x 100 periods each.  python tools/experiments/density_sweep.py"""
import json, os, sys
sys.path.insert(0, '.')
import numpy as np
from tests import _flight

out = {}
for cfg in ("C2", "C5"):
    prm, _ = _flight.make_prm(cfg)
    for dens in (0.1, 0.3, 0.6, 1.0, 1.5):
        seeds = list(range(8000, 8064))
        g = _flight.gpu_flights(seeds, cfg, 100, n_points=3072, world_kw=dict(cyl_per_m=dens), batch=64, mode="task")
        st = _flight.flight_stats(g, prm)
        key = f"{cfg} (N={prm.N}, T={prm.T}) {dens} cyl/m"
        out[key] = dict(flights=st["flights"], inside_drone_radius=st["collided"], through_a_cylinder=st["hit_surface"],
                        min_clearance_median_m=round(st["min_clearance_median"], 3), unsafe_periods=st["unsafe_periods"],
                        solves_per_period=round(st["solves_per_period"], 3), x_final_mean_m=round(st["x_final_mean"], 2))
        print(key, out[key], flush=True)
os.makedirs("gpurun_out/r05c", exist_ok=True)
json.dump(out, open("gpurun_out/r05c/density_sweep.json", "w"), indent=1)
