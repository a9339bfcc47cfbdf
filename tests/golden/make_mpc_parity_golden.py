"""Generates tests/golden/mpc_parity_golden.npz: the solver-independent targets of the MPC parity gate
(|u - u*|_inf <= 1e-3 against the CONVERGED optimum).

For each BASELINE size (C1: 5k points, N=10, K=3; C2: 50k, N=20, K=8; C5: 200k, N=30, K=8) and each of the first 64
bench scenes (seed 100000 + s, avoid_mpc_amd/synth.py):
  ref    [64][20+10N+3KN]  vecRefStates of the step's FIRST solve (kNN at the reference path through the KD oracle,
                           which is pinned to the reference's nanoflann; independent of any MPC solver)
  wstar  [64][nx]          the local optimum reached from the zero warm start (HighLvlMpc.cpp:26-27,35): the oracle's
                           interior-point method run to the rounding floor (tol 1e-9, 400 iterations)
  Jstar  [64]              its objective (true |s|)
  J_lbfgs, du_lbfgs [64]   cross-check by an independent optimiser: scipy L-BFGS-B on the condensed problem started AT
                           wstar's controls and confined to the box |U - U*|_inf <= 0.05 around them (the problem is
                           non-convex with several basins and the optimum sits on kinks of the |v.n| terms, where the
                           sign-based gradient L-BFGS-B sees is large: unconfined, its first line search jumps into
                           another basin on ~10 % of the scenes) -- the objective it ends with and how far the first
                           control moved.  A local minimiser stays put.
  dJ_probe [64]            min over 400 random feasible perturbations of U* (radii 1e-4 .. 0.3) of J - J*: >= 0 at a
                           local minimiser; uses nothing but nlp_f.
Data only.  Run:  python tests/golden/make_mpc_parity_golden.py      (about 25 minutes)
"""
import os
import sys

import numpy as np
import scipy.optimize as so

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from avoid_mpc_amd import synth  # noqa: E402
from tests import _oracle  # noqa: E402
from make_mpc_golden import condensed  # noqa: E402

NSC = 64


def main():
    out = {}
    for cfg in ("C1", "C2", "C5"):
        c = synth.CONFIGS[cfg]
        prm = synth.MpcParams(T=c["T"], K=c["K"])
        N, K = prm.N, prm.K
        lbu = np.array([-prm.a_max_xy, -prm.a_max_xy, prm.a_min_z, -prm.a_max_yaw_dot])
        ubu = np.array([prm.a_max_xy, prm.a_max_xy, prm.a_max_z, prm.a_max_yaw_dot])
        refs, ws, Js, Jl, dul, its, prb = [], [], [], [], [], [], []
        one = synth.MpcParams(T=c["T"], K=c["K"], max_iter=1)
        for s in range(NSC):
            sc = synth.make_scene(c["n"], 100000 + s, prm)
            kd, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
            mpc = _oracle.MpcOracle(prm.T, prm.dt, prm.K); mpc.configure(prm); mpc.set_solver_options(1e-4, 1)
            r = _oracle.step_oracle(kd, ke, mpc, one, _oracle.scene_state_quads(sc, one), sc["pos"][0],
                                    sc["ref_path"].copy(), want_log=True)
            ref = r["ref_log"][0].copy()
            kd.close(); ke.close(); mpc.close()
            P = np.ascontiguousarray(np.concatenate([ref, prm.gain, prm.tau, prm.weights, [prm.radius]]))
            w, info, st = _oracle.mpco_solve(P, np.zeros(10 + 14 * N), lbu, ubu, N, K, prm.dt, tol=1e-9, max_iter=400)
            fg, _ = condensed(P, N, K, prm.dt)
            U0 = np.stack([w[14 * k + 10:14 * k + 14] for k in range(N)]).reshape(-1)
            lb, ub = np.tile(lbu, N), np.tile(ubu, N)
            bounds = list(zip(np.maximum(lb, U0 - 0.05), np.minimum(ub, U0 + 0.05)))
            rr = so.minimize(fg, U0, jac=True, method="L-BFGS-B", bounds=bounds,
                             options=dict(maxiter=2000, maxfun=8000, ftol=1e-16, gtol=1e-10))
            rng = np.random.default_rng(1000 + s)
            probe = 0.0
            for rad in (1e-4, 1e-3, 1e-2, 0.3):
                for _ in range(100):
                    d = rng.normal(size=U0.size)
                    probe = min(probe, fg(np.clip(U0 + d * (rad / np.linalg.norm(d)), lb, ub))[0] - st[0])
            refs.append(ref); ws.append(w); Js.append(st[0]); Jl.append(rr.fun); dul.append(np.abs(rr.x[:4] - U0[:4]).max())
            its.append(info[1]); prb.append(probe)
            print(cfg, s, "J* %.9f" % st[0], "lbfgs-J* %.2e" % (rr.fun - st[0]), "du %.2e" % dul[-1], "nit", rr.nit,
                  "probe %.2e" % probe, flush=True)
        out[cfg + ".ref"] = np.array(refs); out[cfg + ".wstar"] = np.array(ws); out[cfg + ".Jstar"] = np.array(Js)
        out[cfg + ".J_lbfgs"] = np.array(Jl); out[cfg + ".du_lbfgs"] = np.array(dul); out[cfg + ".dJ_probe"] = np.array(prb)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpc_parity_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
