"""Generates tests/golden/flight_golden.npz: closed-loop flights at BASELINE configs[1] size (50 k-point frames, N = 20, K = 8)
flown by (a) the CPU oracle (this project's solver, run to convergence) and (b) the IPOPT-shaped emulation stopped after the
reference's 10 iterations (oracle/ipopt_emul.py; AM/src/HighLvlMpc.cpp:17-23), same worlds, same frames, same loop
(tests/_flight.py).  Neither is the reference's output (CasADi / IPOPT are absent: parity with IPOPT itself stays unpinned);
(b) is this project's best estimate of the regime the reference flies in, (a) is what the product returns.

    python tests/golden/make_flight_golden.py          (~2 minutes on 8 cores)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SEEDS = list(range(5000, 5016))     # the first 16 flights of tests/test_flight_gpu.py::test_flights_c2_gpu_equals_oracle
PERIODS = 100
WORLD = dict(cyl_per_m=1.5)

if __name__ == "__main__":
    from tests import _flight
    o = _flight.oracle_flights(SEEDS, "C2", PERIODS, world_kw=WORLD)
    i = _flight.ipopt_flights(SEEDS, "C2", PERIODS, world_kw=WORLD)
    out = dict(seeds=np.array(SEEDS), periods=PERIODS, cyl_per_m=WORLD["cyl_per_m"])
    for tag, log in (("oracle", o), ("ipopt10", i)):
        out[tag + ".pos"] = log["x"][:, :, 0:3].astype(np.float32)      # trajectories (float32: 1e-6 m is far below what is compared)
        out[tag + ".u"] = log["u"].astype(np.float32)
        out[tag + ".flags"] = log["flags"].astype(np.int16)
        out[tag + ".clearance_min"] = log["clearance"].min(axis=1)
    out["ipopt10.status"] = i["ipopt_status"].astype(np.int8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "flight_golden.npz"), **out)
    d = np.abs(o["x"][:, :, 0:3] - i["x"][:, :, 0:3]).max(axis=2)
    print("max |dpos| oracle vs IPOPT-10 per flight:", np.round(d.max(axis=1), 3))
    print("min clearance oracle :", np.round(out["oracle.clearance_min"], 2))
    print("min clearance ipopt10:", np.round(out["ipopt10.clearance_min"], 2))
