"""EXPERIMENT on the CPU oracle (NOT applied: tools/experiments/patches/r04_stagewise_regularisation_oracle.patch adds the switch
mpco_exp_stagewise to oracle/mpc_oracle.c): regularising a non-positive-definite control block of the Riccati sweep ALONE and in
place against the shipped rule (delta on every block, sweep restarted).  The 64 cold-start fixture scenes per BASELINE size
(tests/golden/mpc_parity_golden.npz): interior-point iterations, backward sweeps (the shipped rule pays one per failed attempt),
convergence, and whether the solve ends in the fixture's optimum.  Result (round 4): C2 sweeps -14.9 % but iterations +8 % (time
-1 %), 23 of 64 scenes end in another optimum; C5 sweeps -7.9 %, iterations +14 % (time +5 %), 2 scenes hit the iteration cap."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import _oracle
from avoid_mpc_amd import synth

G = np.load(os.path.join(ROOT, "tests", "golden", "mpc_parity_golden.npz"))
lib = _oracle.load_oracle()
sw = C.c_int.in_dll(lib, "mpco_exp_stagewise")
dl = C.c_double.in_dll(lib, "mpco_exp_stage_delta_last")
nr = C.c_int.in_dll(lib, "mpco_exp_stage_regs")


def run(cfg, stagewise, max_iter=100):
    c = synth.CONFIGS[cfg]; prm = synth.MpcParams(T=c["T"], K=c["K"]); N, K = prm.N, prm.K
    lbu = [-prm.a_max_xy, -prm.a_max_xy, prm.a_min_z, -prm.a_max_yaw_dot]
    ubu = [prm.a_max_xy, prm.a_max_xy, prm.a_max_z, prm.a_max_yaw_dot]
    rows = []
    for s, ref in enumerate(G[cfg + ".ref"]):
        P = np.concatenate([ref, prm.gain, prm.tau, prm.weights, [prm.radius]])
        sw.value = stagewise; dl.value = 0.0; nr.value = 0
        w, info, st = _oracle.mpco_solve(P, np.zeros(10 + 14 * N), lbu, ubu, N, K, prm.dt, max_iter=max_iter)
        sw.value = 0
        ws = G[cfg + ".wstar"][s]
        rows.append((info[0], info[1], info[2], nr.value, np.abs(w[10:14] - ws[10:14]).max(), st[0]))
    return np.array(rows)


if __name__ == "__main__":
    for cfg in ("C1", "C2", "C5"):
        a, b = run(cfg, 0), run(cfg, 1)
        sweeps_a, sweeps_b = a[:, 1] + a[:, 2], b[:, 1]
        same = (b[:, 4] <= 1e-3)
        lower = (b[:, 5] < a[:, 5] - 1e-6 * np.abs(a[:, 5])) & ~same
        print(f"{cfg}: shipped rule: converged {int((a[:, 0] == 0).sum())}/64, iterations {a[:, 1].mean():.1f}, sweeps {sweeps_a.mean():.1f} "
              f"(failed attempts {a[:, 2].mean():.1f});  stage-wise: converged {int((b[:, 0] == 0).sum())}/64, iterations {b[:, 1].mean():.1f}, "
              f"sweeps {sweeps_b.mean():.1f} (in-place refactorisations {b[:, 3].mean():.1f});  sweeps {100 * (sweeps_b.sum() / sweeps_a.sum() - 1):+.1f} %; "
              f"same optimum as the fixture {int(same.sum())}/64, another one with a lower objective {int(lower.sum())}, iteration cap hit {int((b[:, 0] == 1).sum())}")
