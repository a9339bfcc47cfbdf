"""The reference's own smoke scenario and its re-plan loop (AM/tools/mpc_obstacle_casadi.py:448-536), the "Python caller"
of SURVEY.md section 8(c): cylinder of 100 obstacle points (r = 0.1 at x = 1), start (0, 0, 1), goal (5, 0.1, 1), yaml
parameters (N = 30, K = 3), the script's bounds [-10,-10,-20,-10]..[10,10,20,10], hover warm start, neighbours from
sklearn.neighbors.KDTree exactly as the script takes them; loop = solve -> re-query the K neighbours at the predicted
states -> stop when the watched first-neighbour indices are unchanged, at most mpc_max_iter = 3 passes.

The script asserts nothing and CasADi is absent, so the pins are: (a) every pass ends at a local minimiser confirmed by an
independent optimiser (scipy L-BFGS-B confined around it) -- solver-independent; (b) the survey's own feasibility probe
of this scenario (J = 829.9 -> 794.9 -> 780.0 over the three passes, u0 ~ (7.28, 5.85, 9.85, 0),
the path clears the cylinder at |y| ~ 0.55); (c) the GPU path run through the same loop gives the same passes."""
import math
import os
import sys

import numpy as np
import pytest

from avoid_mpc_amd import synth
from tests import _oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

N, K, DT = 30, 3, 0.033
LBU = np.array([-10.0, -10.0, -20.0, -10.0]); UBU = np.array([10.0, 10.0, 20.0, 10.0])    # :459-465
P_INIT = [0.0, 0.0, 1.0] + [0.0] * 7
P_GOAL = [5.0, 0.1, 1.0] + [0.0] * 7
TAIL = np.concatenate([synth.DEFAULT_GAIN, synth.DEFAULT_TAU, synth.DEFAULT_WEIGHTS, [0.5]])


def run_loop(solve):
    """solve(P, w0) -> w.  Returns the per-pass (P, w) list; restates :448-534 incl. the watched-index bookkeeping."""
    from sklearn.neighbors import KDTree
    obstacles = np.array([[0.1 * math.cos(th) + 1.0, 0.1 * math.sin(th), oz]
                          for oz in np.linspace(0, 3, 10) for th in np.linspace(0, 2 * 3.14, 10)])
    tree = KDTree(obstacles)
    dp = (np.array(P_GOAL) - np.array(P_INIT)) / N
    watched, ref, obs = [], [], []
    w0 = np.zeros(10 + 14 * N)
    for i in range(N):
        pi = np.array(P_INIT) + i * dp
        ref += pi.tolist()
        _, idx = tree.query([pi[0:3]], k=K)
        for j in range(K):
            obs += obstacles[idx[0][j]].tolist()
            watched.append(idx[0][j])
        w0[14 * i + 10:14 * i + 14] = [0.0, 0.0, 9.81, 0.0]
    P = np.array(P_INIT + ref + obs + P_GOAL)
    passes = []
    for _ in range(3):                                  # max_iter_num = mpc_max_iter (yaml :3)
        w = solve(np.concatenate([P, TAIL]), w0)
        passes.append((np.concatenate([P, TAIL]), w.copy()))
        w0 = w
        ref, obs, need_replan = [], [], False
        for i in range(N):
            xi = w[14 * i:14 * i + 10]
            ref += xi.tolist()
            _, idx = tree.query([xi[0:3]], k=K)
            for j in range(K):
                obs += obstacles[idx[0][j]].tolist()
                watched.append(idx[0][j])
            if watched[i] != idx[0][0]:                 # (the script indexes its flat list with the stage number)
                need_replan = True
                watched[i] = idx[0][0]
        P = np.array(P_INIT + ref + obs + P_GOAL)
        if not need_replan:
            break
    return passes


def objective(P, w):
    return _oracle.load_oracle().mpco_nlp_f(np.ascontiguousarray(w), np.ascontiguousarray(P), N, K)


def test_replan_loop_on_the_oracle_against_scipy_and_the_survey_probe():
    import scipy.optimize as so
    from make_mpc_golden import condensed
    passes = run_loop(lambda P, w0: _oracle.mpco_solve(P, w0, LBU, UBU, N, K, DT)[0])
    assert len(passes) == 3
    Js = [objective(P, w) for P, w in passes]
    print("objective per pass:", Js, "u0:", passes[-1][1][10:14])
    for J, Jp in zip(Js, (829.9, 794.9, 780.0)):         # SURVEY.md section 8(c) probe (L-BFGS-B, 36-67 iterations per pass)
        assert abs(J - Jp) <= 2e-3 * Jp, (Js,)
    assert np.abs(passes[-1][1][10:14] - np.array([7.28, 5.85, 9.85, 0.0])).max() <= 0.01          # (after the last pass)
    X = np.stack([passes[-1][1][14 * k:14 * k + 10] for k in range(N + 1)])
    near = np.abs(X[:, 0] - 1.0) < 0.15
    assert near.any() and np.all(np.abs(X[near, 1]) > 0.4) and np.all(np.abs(X[near, 1]) < 0.7)   # clears the cylinder
    lb, ub = np.tile(LBU, N), np.tile(UBU, N)
    for P, w in passes:                                  # every pass: a local minimiser by an independent optimiser
        fg, _ = condensed(P, N, K, DT)
        U = np.stack([w[14 * k + 10:14 * k + 14] for k in range(N)]).reshape(-1)
        r = so.minimize(fg, U, jac=True, method="L-BFGS-B", bounds=list(zip(np.maximum(lb, U - 0.05), np.minimum(ub, U + 0.05))),
                        options=dict(maxiter=500, ftol=1e-16, gtol=1e-10))
        J = objective(P, w)
        assert J - r.fun <= 1e-6 * J and np.abs(r.x[:4] - U[:4]).max() <= 1e-3, (J - r.fun, np.abs(r.x[:4] - U[:4]).max())


@pytest.mark.gpu
def test_replan_loop_on_the_gpu_equals_the_oracle_loop():
    import torch
    from avoid_mpc_amd.host import MpcBatch
    prm = synth.MpcParams(T=N * DT + 1e-9, K=K)
    m = MpcBatch(N * DT + 1e-9, DT, K, 1)
    m.SetupWeights(prm.weights); m.SetupTau(prm.tau); m.SetupGains(prm.gain); m.SetDroneRadius(0.5)
    m.SetDroneAccelLimits(-20.0, 20.0, 10.0, 10.0)       # aMinZ, aMaxZ, aMaxXy, aMaxYawDot of the script
    assert m.N == N

    def solve_gpu(P, w0):
        m.set_warm_start(torch.from_numpy(w0[None].copy()).cuda())
        m.Solve(torch.from_numpy(P[None, :20 + 10 * N + 3 * K * N].copy()).cuda())
        return m.get_warm_start().cpu().numpy()[0]
    gp = run_loop(solve_gpu)
    cp = run_loop(lambda P, w0: _oracle.mpco_solve(P, w0, LBU, UBU, N, K, DT)[0])
    assert len(gp) == len(cp)
    for (Pg, wg), (Pc, wc) in zip(gp, cp):
        assert np.abs(Pg - Pc).max() <= 1e-6 and np.abs(wg - wc).max() <= 1e-6
