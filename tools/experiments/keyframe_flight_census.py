"""Round 5: closed-loop parity census of flights WITH the keyframe map (the reference's default regime, FrameKDMap.cpp:29-32), beyond
tests/test_kfmap_gpu.py: more flights, longer, several max_frame_count.  Frames from rendered depth images; GPU = amk_pipeline TASK
mode with the map in the slot (csrc/kfmap.hip), CPU = tests/_kfmap.py + oracle/step_oracle.c: stepo_run_frames.
usage: python tools/experiments/keyframe_flight_census.py [out.json [cfg:flights:periods:batch:gang:max_frames ...]]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import numpy as np
    from tests import _flight
    out = {}
    runs = [("C1", 128, 150, 32, 2, 10), ("C2", 64, 120, 16, 2, 5), ("YAML", 32, 100, 8, 2, 100)]
    if len(sys.argv) > 2:
        runs = [(a.split(":")[0],) + tuple(int(v) for v in a.split(":")[1:]) for a in sys.argv[2:]]
    for cfg, F, P, batch, gang, mf in runs:
        seeds = list(range(30000, 30000 + F))
        kw = dict(cyl_per_m=1.0, x_first=3.0, length=0.33 * P + 45.0)
        kf = dict(max_frame_count=mf, th_dist=0.1, th_count=10)
        t0 = time.time()
        g = _flight.gpu_depth_flights(seeds, cfg, P, world_kw=kw, gang=gang, batch=batch, keyframes=kf)
        t1 = time.time()
        o = _flight.oracle_depth_flights(seeds, cfg, P, world_kw=kw, keyframes=kf)
        t2 = time.time()
        prm, _ = _flight.make_prm(cfg)
        cmp = _flight.compare(g, o, pos_tol=1e-6)
        sep = cmp["separation_period"]
        equal = {}
        for key in ("n_keyframes", "n_query_frames", "outliers", "map_points", "n_cloud"):
            bad = 0
            for f in range(F):
                upto = P if sep[f] < 0 else sep[f]
                bad += int(not np.array_equal(g[key][f, :upto], o[key][f, :upto]))
            equal[key] = bad
        rep = {"config": f"{cfg}: N={prm.N}, K={prm.K}, test sensor 320x240/5, max_frame_count={mf}, gang {gang}", "flights": F, "periods": P,
               "separated": cmp["separated"], "separation_periods": sep[sep >= 0].tolist(),
               "dpos_max_while_flags_agree_m": cmp["dpos_max_while_together"],
               "dpos_final_of_separated_m": np.round(cmp["dpos_final"][sep >= 0], 9).tolist(),
               "flights_with_a_map_statistic_that_differs_before_separation": equal,
               "keyframes_mean_max": [float(o["n_keyframes"].mean()), int(o["n_keyframes"].max())],
               "query_frames_mean_max": [float(o["n_query_frames"].mean()), int(o["n_query_frames"].max())],
               "map_points_mean": float(o["map_points"].mean()), "gpu": _flight.flight_stats(g, prm), "oracle": _flight.flight_stats(o, prm),
               "seconds_gpu_driver_incl_rendering": round(t1 - t0, 1), "seconds_cpu_oracle_all_cores": round(t2 - t1, 1)}
        out[f"{cfg}_mf{mf}"] = rep
        print(cfg, json.dumps(rep), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
