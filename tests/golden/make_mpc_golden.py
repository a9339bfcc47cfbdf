"""Generates tests/golden/mpc_golden.npz: substitute pins for the MPC half (PARITY UNPINNED --
CasADi/IPOPT are not available, see oracle/mpc_oracle.c).

Contents (data only):
  smoke.*   the reference's own smoke scenario (AM/tools/mpc_obstacle_casadi.py:448-498: cylinder of 100
            obstacle points, start (0,0,1), goal (5,0.1,1), hover warm start, script bounds) with the
            yaml parameters (AM/config/mpc_parameters.yaml): P, w0, bounds, and the converged optimum
            of the condensed problem from scipy.optimize L-BFGS-B (solver-independent target).
  qp.*      the same scenario with collide_lambda = 0 (an equality-constrained QP): dense KKT solution.
Run:  python tests/golden/make_mpc_golden.py
"""
import math
import os
import sys

import numpy as np
import scipy.optimize as so

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mpc_oracle_np as M  # noqa: E402
from avoid_mpc_amd import synth  # noqa: E402


def smoke_problem(N=30, K=3, dt=0.033, lam=None):
    weights = list(synth.DEFAULT_WEIGHTS)
    if lam is not None:
        weights[24] = lam
    p_init = [0.0, 0.0, 1.0] + [0.0] * 7
    obstacles = []
    for oz in np.linspace(0, 3, 10):
        for th in np.linspace(0, 2 * 3.14, 10):
            obstacles.append([0.1 * math.cos(th) + 1.0, 0.1 * math.sin(th), oz])
    obstacles = np.array(obstacles)
    p_goal = [5.0, 0.1, 1.0] + [0.0] * 7
    dp = (np.array(p_goal) - np.array(p_init)) / N
    ref, obs = [], []
    for i in range(N):
        pi = np.array(p_init) + i * dp
        ref += pi.tolist()
        d = np.sum((obstacles - pi[:3]) ** 2, axis=1)
        obs += obstacles[np.argsort(d, kind="stable")[:K]].reshape(-1).tolist()
    P = np.array(p_init + ref + obs + p_goal + synth.DEFAULT_GAIN + synth.DEFAULT_TAU + weights + [0.5])
    w0 = np.zeros(10 + 14 * N)
    for k in range(N):
        w0[14 * k + 10:14 * k + 14] = [0, 0, 9.81, 0]
    lbu = np.array([-10.0, -10.0, -20.0, -10.0]); ubu = np.array([10.0, 10.0, 20.0, 10.0])
    return P, w0, lbu, ubu, obstacles


def condensed(P, N, K, dt):
    pp = M.split_p(P, N, K)
    A, B, c = M.affine_dynamics(pp["tau"], dt)

    def fg(u):
        U = u.reshape(N, 4)
        X = M.rollout(pp["x_init"], U, A, B, c)
        J, q, Q, r, Rd = M.total_cost(X, U, pp, N, True)
        lam = q[N].copy(); gU = np.zeros_like(U)
        for k in range(N - 1, -1, -1):
            gU[k] = r[k] + B.T @ lam
            if k > 0:
                lam = q[k] + A.T @ lam
        return J, gU.reshape(-1)
    return fg, (A, B, c, pp)


def main():
    N, K, dt = 30, 3, 0.033
    out = {}
    P, w0, lbu, ubu, obstacles = smoke_problem(N, K, dt)
    fg, _ = condensed(P, N, K, dt)
    U0 = np.tile([0, 0, 9.81, 0], N).astype(float)
    bounds = [(lbu[i % 4], ubu[i % 4]) for i in range(4 * N)]
    r = so.minimize(fg, U0, jac=True, method="L-BFGS-B", bounds=bounds,
                    options=dict(maxiter=5000, maxfun=20000, ftol=1e-16, gtol=1e-10))
    out.update({"smoke.P": P, "smoke.w0": w0, "smoke.lbu": lbu, "smoke.ubu": ubu, "smoke.obstacles": obstacles,
                "smoke.NKdt": np.array([N, K, dt]), "smoke.scipy_fun": np.array(r.fun), "smoke.scipy_U": r.x})
    print("smoke: L-BFGS-B f* =", r.fun, "nit", r.nit)
    # lambda = 0: QP.  KKT: [H J'; J 0] [w; l] = [-g0; -c0] on the multiple-shooting variables
    Pq, _, _, _, _ = smoke_problem(N, K, dt, lam=0.0)
    nx = 10 + 14 * N
    z = np.zeros(nx)
    H = M.nlp_hess_f(z, Pq, N, K, majorise_abs=False)
    g0 = M.nlp_grad_f(z, Pq, N, K)
    Jg = M.nlp_jac_g(z, Pq, N, K, dt)
    c0 = M.nlp_g(z, Pq, N, K, dt)
    KKT = np.block([[H, Jg.T], [Jg, np.zeros((Jg.shape[0],) * 2)]])
    sol = np.linalg.solve(KKT, -np.concatenate([g0, c0]))
    wq = sol[:nx]
    out.update({"qp.P": Pq, "qp.w": wq, "qp.f": np.array(M.nlp_f(wq, Pq, N, K))})
    Uq = np.stack([wq[14 * k + 10:14 * k + 14] for k in range(N)])
    print("qp: f* =", out["qp.f"], "|U|max =", np.abs(Uq).max(axis=0), "feas", np.abs(M.nlp_g(wq, Pq, N, K, dt)).max())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpc_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
