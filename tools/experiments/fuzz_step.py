"""Randomised cross-check of the whole control step (amk_step_batch) against the CPU oracle: random horizons (N = 6 .. 31: the
generic-horizon solve kernel as well as the three specialised ones), neighbour counts K = 1 .. 10, cloud sizes from fewer points
than K up to 20 k, empty / one-point / small edge clouds, ragged batches, starts inside obstacles.  Scene by scene: flags equal
and |gpu - oracle| <= 1e-6; scenes whose iteration count differs (a rounding-level branch flip) are counted, as tests/
test_step_gpu.py: compare does.
usage: python tools/experiments/fuzz_step.py [seed] [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from avoid_mpc_amd import synth
from tests.test_step_gpu import run_both

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120.0)
it = scenes_n = flipped = other = bad = 0
worst = 0.0
while time.time() < t_end:
    it += 1
    N = int(rng.integers(6, 32))
    K = int(rng.integers(1, 11))
    prm = synth.MpcParams(T=(N + 0.5) * 0.033, K=K)
    assert prm.N == N, (prm.N, N)
    S = int(rng.integers(1, 7))
    scenes = []
    for s in range(S):
        n = int(rng.choice([max(0, K - 1), K, K + 1, 50, 700, 3000, 20000]))
        sc = synth.make_scene(max(n, 16), int(rng.integers(0, 1 << 30)), prm)
        sc["cloud"] = sc["cloud"][:n]
        ne = int(rng.choice([0, 1, 40, len(sc["edge"])]))
        sc["edge"] = sc["edge"][:ne]
        if rng.random() < 0.2 and n > 0:      # start next to / inside an obstacle: PlanWapionts' edge snap, unsafe scenes
            sc["pos"] = sc["cloud"][rng.integers(0, n)].astype(np.float64) + rng.normal(size=3) * 0.05
            sc["ref_path"] = synth.make_ref_path(sc["pos"], prm)
        scenes.append(sc)
    gpu, cpu = run_both(torch, scenes, prm, n_steps=2)
    diverged = set()
    for t in range(2):
        for s, r in enumerate(cpu[t]):
            if s in diverged:
                continue
            scenes_n += 1
            g = gpu[t]
            du = np.abs(g["u"][s] - r["u"]).max()
            dx = np.abs(g["x0array"][s] - r["x0array"]).max() if r["flags"][1] > 0 else 0.0
            dr = np.abs(g["ref_path"][s] - r["ref_path"]).max()
            d = max(du, dx, dr)
            if np.array_equal(g["flags"][s], r["flags"]):
                worst = max(worst, d)
                if not d <= 1e-6:
                    bad += 1; print("MISMATCH", it, "N", N, "K", K, "scene", s, "step", t, du, dx, dr, r["flags"], flush=True)
            else:
                diverged.add(s)
                if not np.array_equal(g["flags"][s][:3], r["flags"][:3]):
                    bad += 1; print("FLAGS", it, "N", N, "K", K, "scene", s, "step", t, g["flags"][s], r["flags"], flush=True)
                else:
                    flipped += 1
                    other += d > 1e-4
print(f"fuzz batches {it}, scene-steps {scenes_n}: mismatches {bad}, different iteration count {flipped} ({other} of them at another optimum), "
      f"worst |gpu - oracle| elsewhere {worst:.2e}")
