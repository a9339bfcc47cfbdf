"""Generates tests/golden/mpc_independent_golden.npz: what INDEPENDENT algorithms do on the fixture problems of
tests/golden/mpc_parity_golden.npz (3 BASELINE sizes x 64 bench scenes, the vecRefStates of the step's first solve), from
the reference's zero start (HighLvlMpc.cpp:26-27,35), on the full multiple-shooting NLP (mpc_obstacle_casadi.py:153-224)
with the exact nlp_jac_g / nlp_hess_l of oracle/mpc_oracle_np.py.

  ipopt10.u / .f / .theta / .status / .iters   [64]   oracle/ipopt_emul.py (IPOPT-shaped filter line-search interior point,
        IPOPT's defaults + the reference's options) stopped after the reference's max_iter = 10: the control the reference
        would publish, its objective and constraint violation (sum |g|)
  ipoptc.u / .f / .status / .iters             [64]   the same emulation allowed 300 iterations (status 0 converged,
        3 = the next step needed IPOPT's restoration phase, which is not emulated)
  tc.u / .f / .status / .nit / .viol           [64]   scipy.optimize.minimize(method="trust-constr") -- a trust-region
        interior-point method, nothing in common with this project's solver -- to gtol 1e-8
  ipopt10w.u / .f / .status / .iters           [64]   (run with --warm) the 10-iteration emulation started AT the fixture's
        converged point w* of the same problem: the reference's operating regime is a warm start from the previous solution
        (mNlpW0 = sol, HighLvlMpc.cpp:129), with mu_init still 0.1 -- how far does the truncated solve end from the optimum
        it was started at
Data only (no source text).  Run:  python tests/golden/make_mpc_independent_golden.py   (about 40 minutes on 8 cores)
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "mpc_parity_golden.npz")))   # in memory: the workers are forked


def problem(cfg, s):
    import ipopt_emul as IE
    from avoid_mpc_amd import synth
    c = synth.CONFIGS[cfg]
    prm = synth.MpcParams(T=c["T"], K=c["K"])
    lbu = np.array([-prm.a_max_xy, -prm.a_max_xy, prm.a_min_z, -prm.a_max_yaw_dot])
    ubu = np.array([prm.a_max_xy, prm.a_max_xy, prm.a_max_z, prm.a_max_yaw_dot])
    P = np.concatenate([G[cfg + ".ref"][s], prm.gain, prm.tau, prm.weights, [prm.radius]])
    return IE.ShootingNlp(P, prm.N, prm.K, prm.dt, lbu, ubu)


def one(job):
    cfg, s = job
    import ipopt_emul as IE
    from scipy.optimize import Bounds, LinearConstraint, minimize
    nlp = problem(cfg, s)
    x0 = np.zeros(nlp.n)
    r10 = IE.solve(nlp, x0, max_iter=10)
    rc = IE.solve(nlp, x0, max_iter=300)
    tc = minimize(nlp.f, x0, jac=nlp.grad, hess=nlp.hess, method="trust-constr",
                  constraints=[LinearConstraint(nlp.J, -nlp.g0, -nlp.g0)], bounds=Bounds(nlp.xl, nlp.xu),
                  options=dict(gtol=1e-8, xtol=1e-12, maxiter=3000, initial_barrier_parameter=0.1))
    ws = G[cfg + ".wstar"][s]
    print(cfg, s, "ipopt10 |du| %.2e  ipoptc st %d it %d |du| %.2e  tc st %d nit %d |du| %.2e f-J* %.2e" % (
        np.abs(r10["x"][10:14] - ws[10:14]).max(), rc["status"], rc["iters"], np.abs(rc["x"][10:14] - ws[10:14]).max(),
        tc.status, tc.nit, np.abs(tc.x[10:14] - ws[10:14]).max(), tc.fun - G[cfg + ".Jstar"][s]), flush=True)
    return dict(cfg=cfg, s=s, u10=r10["x"][10:14], f10=r10["f"], th10=r10["theta"], st10=r10["status"], it10=r10["iters"],
                uc=rc["x"][10:14], fc=rc["f"], stc=rc["status"], itc=rc["iters"],
                utc=tc.x[10:14], ftc=tc.fun, sttc=tc.status, nittc=tc.nit, violtc=tc.constr_violation)


def one_warm(job):
    cfg, s = job
    import ipopt_emul as IE
    nlp = problem(cfg, s)
    r = IE.solve(nlp, G[cfg + ".wstar"][s], max_iter=10)
    return dict(u=r["x"][10:14], f=r["f"], st=r["status"], it=r["iters"])


def main():
    path = os.path.join(ROOT, "tests", "golden", "mpc_independent_golden.npz")
    if "--warm" in sys.argv:
        out = dict(np.load(path))
        for cfg in ("C1", "C2", "C5"):
            with ProcessPoolExecutor(int(os.environ.get("NPROC", "8"))) as ex:
                res = list(ex.map(one_warm, [(cfg, s) for s in range(64)], chunksize=1))
            out.update({cfg + ".ipopt10w.u": np.array([r["u"] for r in res]), cfg + ".ipopt10w.f": np.array([r["f"] for r in res]),
                        cfg + ".ipopt10w.status": np.array([r["st"] for r in res]), cfg + ".ipopt10w.iters": np.array([r["it"] for r in res])})
            d = np.abs(out[cfg + ".ipopt10w.u"] - G[cfg + ".wstar"][:, 10:14]).max(1)
            print(cfg, "warm-started ipopt10: |du| median %.3g p90 %.3g max %.3g" % (np.median(d), np.quantile(d, .9), d.max()), flush=True)
        np.savez_compressed(path, **out)
        return
    for cfg in [a for a in sys.argv[1:] if not a.startswith("-")] or ["C1", "C2", "C5"]:   # saved after every size
        jobs = [(cfg, s) for s in range(64)]
        with ProcessPoolExecutor(int(os.environ.get("NPROC", "8"))) as ex:
            res = list(ex.map(one, jobs, chunksize=1))
        out = dict(np.load(path)) if os.path.exists(path) else {}
        rr = [r for r in res if r["cfg"] == cfg]
        st = lambda k: np.array([r[k] for r in rr])
        out.update({cfg + ".ipopt10.u": st("u10"), cfg + ".ipopt10.f": st("f10"), cfg + ".ipopt10.theta": st("th10"),
                    cfg + ".ipopt10.status": st("st10"), cfg + ".ipopt10.iters": st("it10"),
                    cfg + ".ipoptc.u": st("uc"), cfg + ".ipoptc.f": st("fc"), cfg + ".ipoptc.status": st("stc"),
                    cfg + ".ipoptc.iters": st("itc"), cfg + ".tc.u": st("utc"), cfg + ".tc.f": st("ftc"),
                    cfg + ".tc.status": st("sttc"), cfg + ".tc.nit": st("nittc"), cfg + ".tc.viol": st("violtc")})
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
