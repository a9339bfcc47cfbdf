"""Closed-loop parity census beyond tests/test_flight_gpu.py: more flights, longer, and all three BASELINE sizes.  GPU
(amk_pipeline, keep_warm_start) against the CPU oracle on the same worlds and frames (tests/_flight.py); writes a JSON report.
usage: python tools/experiments/flight_census.py [out.json [cfg:flights:periods:batch ...]]"""
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
    runs = [("C2", 256, 150, 64), ("C5", 16, 60, 16), ("C1", 128, 150, 128)]
    if len(sys.argv) > 2:   # e.g. C2:1024:150:64
        runs = [(a.split(":")[0],) + tuple(int(v) for v in a.split(":")[1:]) for a in sys.argv[2:]]
    for cfg, F, P, batch in runs:
        seeds = list(range(20000, 20000 + F))
        kw = dict(cyl_per_m=1.5, length=100.0)
        t0 = time.time()
        g = _flight.gpu_flights(seeds, cfg, P, world_kw=kw, batch=batch)
        t1 = time.time()
        o = _flight.oracle_flights(seeds, cfg, P, world_kw=kw)
        t2 = time.time()
        prm, _ = _flight.make_prm(cfg)
        cmp = _flight.compare(g, o, pos_tol=1e-6)
        sep = cmp["separation_period"]
        rep = {"flights": F, "periods": P, "separated": cmp["separated"], "separation_periods": sep[sep >= 0].tolist(),
               "dpos_max_while_flags_agree_m": cmp["dpos_max_while_together"],
               "dpos_final_of_separated_m": np.round(cmp["dpos_final"][sep >= 0], 9).tolist(),
               "du_max_while_together": cmp["du_max_together"], "gpu": _flight.flight_stats(g, prm), "oracle": _flight.flight_stats(o, prm),
               "seconds_gpu_driver_incl_frame_generation": round(t1 - t0, 1), "seconds_cpu_oracle_all_cores": round(t2 - t1, 1)}
        out[cfg] = rep
        print(cfg, json.dumps(rep), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
