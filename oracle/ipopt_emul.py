"""IPOPT-SHAPED interior-point emulation on the reference's multiple-shooting NLP.  TEST INFRASTRUCTURE ONLY.

Only tests/, tools/experiments and the fixture generators under tests/golden may import this; the product never does.

PARITY UNPINNED.  The reference solves its NLP with IPOPT (through casadi::nlpsol) and STOPS IT AFTER 10 ITERATIONS without
reading the status (AM/src/HighLvlMpc.cpp:17-23,116-129: tol 1e-4, max_iter 10, warm_start_init_point yes, print_level 0;
AM = roswrapper/ros/src/avoid_mpc).  IPOPT, MUMPS and CasADi are neither under /root/reference nor in this image, so IPOPT's
10th iterate cannot be reproduced.  What this file does instead: it restates the ALGORITHM IPOPT documents -- A. Waechter,
L. T. Biegler, "On the implementation of an interior-point filter line-search algorithm for large-scale nonlinear
programming", Math. Program. 106 (2006), sections 2 and 3, with IPOPT 3.14's default option values as the author recalls
them (SURVEY.md appendix B lists the same hypotheses) -- on the NLP of AM/tools/mpc_obstacle_casadi.py:51-242 as
oracle/mpc_oracle_np.py restates it, with dense linear algebra (LAPACK dsytrf = Bunch-Kaufman LDL', the role MUMPS plays).
Its purpose is an ESTIMATE of the one number a maintainer swapping ObstacleAvoidanceMPC needs: how far the control the
reference actually publishes (IPOPT's iterate when max_iter = 10 strikes) is from the converged optimum this project
returns.  It is not a bit-level model of IPOPT: pivoting, the restoration phase (not emulated: a step that would enter it
is reported), the watchdog and the tiny-step logic can all move individual iterates.

What is restated (paper section / IPOPT option):
  * gradient-based NLP scaling at the user's starting point (nlp_scaling_max_gradient = 100)          sec. 3.8
  * warm_start_init_point = yes: x0 pushed into the bounds by warm_start_bound_push / _frac = 1e-3, bound multipliers
    max(user value = 0, warm_start_mult_bound_push = 1e-3), constraint multipliers = the user's (0: HighLvlMpc.cpp passes
    neither lam_x0 nor lam_g0), mu_init = 0.1                                                           sec. 3.6 / options
  * optimality error E_mu with the s_d, s_c scaling, s_max = 100                                        eq. (5), (6)
  * monotone barrier update, kappa_eps = 10, kappa_mu = 0.2, theta_mu = 1.5, tau = max(0.99, 1 - mu),
    mu >= min(tol, compl_inf_tol) / (barrier_tol_factor + 1)                                             eq. (7), (8)
  * primal-dual step from the augmented system with Sigma = Z/S, exact Hessian (CasADi's SX derivative of fabs: sign,
    no curvature), inertia correction delta_w / delta_c                                                  eq. (11), (13), alg. IC
  * fraction to the boundary for x and z, separate dual step length; the constraint multipliers move with the primal
    step length (alpha_for_y = primal)                                                                   eq. (15)
  * filter line search: theta = ||c||_1, switching condition, Armijo, sufficient decrease, filter augmentation,
    theta_min / theta_max, alpha_min, second-order correction (p_max = 4, kappa_soc = 0.99)               sec. 2.3, 2.4, alg. A
  * bound-multiplier reset with kappa_Sigma = 1e10                                                       eq. (16)
  * termination: scaled E_0 <= tol and the unscaled dual_inf_tol = 1 / constr_viol_tol = 1e-4 / compl_inf_tol = 1e-4,
    or max_iter                                                                                          options
"""
import numpy as np
from scipy.linalg import lapack

import mpc_oracle_np as M


class ShootingNlp:
    """The reference's NLP in the variables w = [X_0, U_0, ..., U_{N-1}, X_N] (mpc_obstacle_casadi.py:153-224)."""

    def __init__(self, P, N, K, dt, lbu, ubu):
        self.P, self.N, self.K, self.dt = np.asarray(P, np.float64), N, K, dt
        self.n = 10 + 14 * N
        self.m = 10 + 10 * N
        self.J = M.nlp_jac_g(np.zeros(self.n), self.P, N, K, dt)     # constant: the dynamics are affine
        self.g0 = M.nlp_g(np.zeros(self.n), self.P, N, K, dt)        # g(w) = J w + g0 exactly
        self.xl = np.full(self.n, -np.inf); self.xu = np.full(self.n, np.inf)      # HighLvlMpc.cpp:70-92
        for k in range(N):
            self.xl[14 * k + 10:14 * k + 14] = lbu
            self.xu[14 * k + 10:14 * k + 14] = ubu

    def f(self, w): return float(M.nlp_f(w, self.P, self.N, self.K))
    def grad(self, w): return M.nlp_grad_f(w, self.P, self.N, self.K)
    def hess(self, w): return M.nlp_hess_f(w, self.P, self.N, self.K, majorise_abs=False)
    def g(self, w): return self.J @ w + self.g0


DEFAULTS = dict(tol=1e-4, max_iter=10, mu_init=0.1, warm_start=True, bound_push=1e-3, bound_frac=1e-3,
                mult_bound_push=1e-3, bound_mult_init_val=1.0, nlp_scaling_max_gradient=100.0, s_max=100.0,
                kappa_eps=10.0, kappa_mu=0.2, theta_mu=1.5, tau_min=0.99, kappa_sigma=1e10, compl_inf_tol=1e-4,
                dual_inf_tol=1.0, constr_viol_tol=1e-4, gamma_theta=1e-5, gamma_phi=1e-5, eta_phi=1e-8,
                delta=1.0, s_theta=1.1, s_phi=2.3, gamma_alpha=0.05, p_max=4, kappa_soc=0.99,
                delta_w_min=1e-20, delta_w_0=1e-4, delta_w_max=1e40, kappa_w_minus=1.0 / 3.0, kappa_w_plus=8.0,
                kappa_w_plus_bar=100.0, delta_c_bar=1e-8, kappa_c=0.25)


def _inertia(ldu, ipiv):
    """(positive, negative, zero) eigenvalue counts of the block-diagonal factor of LAPACK's dsytrf (lower)."""
    n = len(ipiv)
    pos = neg = zero = 0
    i = 0
    while i < n:
        if ipiv[i] > 0:
            d = ldu[i, i]
            pos += d > 0; neg += d < 0; zero += d == 0
            i += 1
        else:  # 2 x 2 pivot: Bunch-Kaufman chooses it with a negative determinant -> one of each sign
            a, b, c = ldu[i, i], ldu[i + 1, i], ldu[i + 1, i + 1]
            det = a * c - b * b
            if det < 0:
                pos += 1; neg += 1
            elif det > 0:
                if a + c > 0: pos += 2
                else: neg += 2
            else:
                zero += 1; pos += (a + c) > 0; neg += (a + c) < 0
            i += 2
    return int(pos), int(neg), int(zero)


def solve(nlp, x0, **kw):
    """-> dict(x, iters, status, mu, trace, restoration).  status: 0 converged, 1 max_iter, 3 a step would have entered the
    restoration phase (not emulated; the iterate before that step is returned), 4 inertia correction overflow."""
    o = dict(DEFAULTS); o.update(kw)
    n, m = nlp.n, nlp.m
    xl, xu, J = nlp.xl, nlp.xu, nlp.J
    bL, bU = np.isfinite(xl), np.isfinite(xu)
    x = np.array(x0, np.float64)
    # ---- NLP scaling (sec. 3.8): at the user's starting point
    g0 = nlp.grad(x)
    gmax = np.abs(g0).max()
    df = min(1.0, o["nlp_scaling_max_gradient"] / gmax) if gmax > o["nlp_scaling_max_gradient"] else 1.0
    # rows of J have max entry 1 <= 100: constraint scaling 1
    # ---- starting point (sec. 3.6; warm start: same push with the warm_start_* values, which equal the defaults 1e-3 here... )
    k1, k2 = o["bound_push"], o["bound_frac"]
    pL = np.where(bL, np.minimum(k1 * np.maximum(1.0, np.abs(np.where(bL, xl, 0.0))), k2 * np.where(bL & bU, xu - xl, np.inf)), 0.0)
    pU = np.where(bU, np.minimum(k1 * np.maximum(1.0, np.abs(np.where(bU, xu, 0.0))), k2 * np.where(bL & bU, xu - xl, np.inf)), 0.0)
    x = np.where(bL, np.maximum(x, xl + pL), x)
    x = np.where(bU, np.minimum(x, xu - pU), x)
    z0 = o["mult_bound_push"] if o["warm_start"] else o["bound_mult_init_val"]
    zL = np.where(bL, z0, 0.0); zU = np.where(bU, z0, 0.0)
    lam = np.zeros(m)
    if not o["warm_start"]:   # least-squares multipliers (eq. (36)); discarded when larger than lambda_max = 1e3
        g = df * nlp.grad(x)
        Kls = np.block([[np.eye(n), J.T], [J, np.zeros((m, m))]])
        sol = np.linalg.solve(Kls, -np.concatenate([g - zL + zU, np.zeros(m)]))
        lam = sol[n:] if np.abs(sol[n:]).max() <= 1e3 else np.zeros(m)
    mu = o["mu_init"]
    tau = max(o["tau_min"], 1.0 - mu)
    mu_floor = min(o["tol"], o["compl_inf_tol"]) / (o["kappa_eps"] + 1.0)

    def slacks(xx):
        return np.where(bL, xx - xl, 1.0), np.where(bU, xu - xx, 1.0)

    def theta(xx): return np.abs(nlp.g(xx)).sum()

    def phi(xx, mu_):
        sL, sU = slacks(xx)
        if (sL[bL] <= 0).any() or (sU[bU] <= 0).any():
            return np.inf
        return df * nlp.f(xx) - mu_ * (np.log(sL[bL]).sum() + np.log(sU[bU]).sum())

    def errors(xx, lam_, zL_, zU_, mu_, grad):
        sL, sU = slacks(xx)
        nb = int(bL.sum() + bU.sum())
        zsum = zL_.sum() + zU_.sum()
        s_d = max(o["s_max"], (np.abs(lam_).sum() + zsum) / (m + nb)) / o["s_max"]
        s_c = max(o["s_max"], zsum / max(nb, 1)) / o["s_max"]
        dual = np.abs(grad + J.T @ lam_ - zL_ + zU_).max()
        prim = np.abs(nlp.g(xx)).max()
        comp = max(np.abs(sL * zL_ - mu_)[bL].max(initial=0.0), np.abs(sU * zU_ - mu_)[bU].max(initial=0.0))
        return max(dual / s_d, prim, comp / s_c), (dual, prim, comp)

    th0 = theta(x)
    theta_max, theta_min = 1e4 * max(1.0, th0), 1e-4 * max(1.0, th0)
    filt = []   # pairs (theta, phi); the region theta >= theta_max is forbidden from the start
    delta_w_last = 0.0
    trace = []
    status, restoration = 1, False
    it = 0
    while True:
        grad = df * nlp.grad(x)
        E0, (dual, prim, comp) = errors(x, lam, zL, zU, 0.0, grad)
        if E0 <= o["tol"] and dual / df <= o["dual_inf_tol"] and prim <= o["constr_viol_tol"] and comp / df <= o["compl_inf_tol"]:
            status = 0
            break
        if it >= o["max_iter"]:
            status = 1
            break
        # ---- barrier update (eq. (7), (8)); the filter is reset
        Emu, _ = errors(x, lam, zL, zU, mu, grad)
        while Emu <= o["kappa_eps"] * mu and mu > mu_floor:
            mu = max(mu_floor, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))
            tau = max(o["tau_min"], 1.0 - mu)
            filt = []
            Emu, _ = errors(x, lam, zL, zU, mu, grad)
        sL, sU = slacks(x)
        Sigma = np.where(bL, zL / sL, 0.0) + np.where(bU, zU / sU, 0.0)
        W = df * nlp.hess(x)
        c = nlp.g(x)
        gphi = grad - np.where(bL, mu / sL, 0.0) + np.where(bU, mu / sU, 0.0)       # gradient of the barrier function
        rhs_x = -(gphi + J.T @ lam)

        # ---- search direction with inertia correction (alg. IC)
        def factor(dw, dc):
            Kmat = np.zeros((n + m, n + m))
            Kmat[:n, :n] = W
            Kmat[:n, :n][np.diag_indices(n)] += Sigma + dw
            Kmat[n:, :n] = J
            Kmat[n:, n:][np.diag_indices(m)] = -dc
            ldu, ipiv, info = lapack.dsytrf(Kmat, lower=1)
            return ldu, ipiv, _inertia(ldu, ipiv) if info == 0 else (0, 0, n + m)

        dw, dc = 0.0, 0.0
        ldu, ipiv, (npos, nneg, nzero) = factor(dw, dc)
        n_fact = 1
        if not (npos == n and nneg == m and nzero == 0):
            if nzero > 0:
                dc = o["delta_c_bar"] * mu ** o["kappa_c"]
            dw = o["delta_w_0"] if delta_w_last == 0.0 else max(o["delta_w_min"], o["kappa_w_minus"] * delta_w_last)
            while True:
                ldu, ipiv, (npos, nneg, nzero) = factor(dw, dc)
                n_fact += 1
                if npos == n and nneg == m and nzero == 0:
                    break
                dw *= o["kappa_w_plus_bar"] if delta_w_last == 0.0 else o["kappa_w_plus"]
                if dw > o["delta_w_max"]:
                    break
            if dw > o["delta_w_max"]:
                status = 4
                break
            delta_w_last = dw

        def kkt_solve(rc):
            sol, info = lapack.dsytrs(ldu, ipiv, np.concatenate([rhs_x, -rc]), lower=1)
            return sol[:n], sol[n:]

        dx, dlam = kkt_solve(c)
        dzL = np.where(bL, mu / sL - zL - zL / sL * dx, 0.0)
        dzU = np.where(bU, mu / sU - zU + zU / sU * dx, 0.0)

        def frac_to_boundary(d):
            a = 1.0
            neg = bL & (d < 0)
            if neg.any(): a = min(a, (-tau * sL[neg] / d[neg]).min())
            pos = bU & (d > 0)
            if pos.any(): a = min(a, (tau * sU[pos] / d[pos]).min())
            return a

        a_max = frac_to_boundary(dx)
        a_z = 1.0
        q = bL & (dzL < 0)
        if q.any(): a_z = min(a_z, (-tau * zL[q] / dzL[q]).min())
        q = bU & (dzU < 0)
        if q.any(): a_z = min(a_z, (-tau * zU[q] / dzU[q]).min())

        # ---- filter line search (alg. A, steps 5 - 9)
        th, ph = np.abs(c).sum(), phi(x, mu)
        dphi = float(gphi @ dx)

        def acceptable_to_filter(tt, pp_):
            if tt >= theta_max:
                return False
            return all(not (tt >= ft and pp_ >= fp) for ft, fp in filt)

        def switching(alpha): return dphi < 0 and alpha * (-dphi) ** o["s_phi"] > o["delta"] * th ** o["s_theta"]

        def accept(tt, pp_, alpha):
            """-> (accepted, f_type)"""
            if not acceptable_to_filter(tt, pp_):
                return False, False
            if th <= theta_min and switching(alpha):
                return pp_ <= ph + o["eta_phi"] * alpha * dphi, True
            return (tt <= (1 - o["gamma_theta"]) * th) or (pp_ <= ph - o["gamma_phi"] * th), False

        if dphi < 0 and th <= theta_min:
            a_min = min(o["gamma_theta"], o["gamma_phi"] * th / (-dphi), o["delta"] * th ** o["s_theta"] / (-dphi) ** o["s_phi"])
        elif dphi < 0:
            a_min = min(o["gamma_theta"], o["gamma_phi"] * th / (-dphi))
        else:
            a_min = o["gamma_theta"]
        a_min *= o["gamma_alpha"]
        alpha, ok, ftype, used_soc, n_ls = a_max, False, False, False, 0
        x_new = None
        while True:
            if alpha < a_min:
                break
            xt = x + alpha * dx
            tt, pt = theta(xt), phi(xt, mu)
            n_ls += 1
            ok, ftype = accept(tt, pt, alpha)
            if ok:
                x_new = xt
                break
            if n_ls == 1 and tt >= th:   # second-order correction (sec. 2.4)
                c_soc = alpha * c + nlp.g(xt)
                th_old = th
                for p in range(o["p_max"]):
                    dxc, dlc = kkt_solve(c_soc)
                    a_soc = frac_to_boundary(dxc)
                    xs = x + a_soc * dxc
                    ts, ps = theta(xs), phi(xs, mu)
                    ok, ftype = accept(ts, ps, alpha)
                    if ok:
                        x_new, dlam, used_soc = xs, dlc, True
                        # the bound multipliers keep the step computed from the uncorrected direction (as IPOPT does)
                        break
                    if ts > o["kappa_soc"] * th_old:
                        break
                    c_soc = a_soc * c_soc + nlp.g(xs)
                    th_old = ts
                if ok:
                    break
            alpha *= 0.5
        if not ok:
            status, restoration = 3, True
            break
        if not ftype:   # h-type step: augment the filter (eq. (22))
            filt.append(((1 - o["gamma_theta"]) * th, ph - o["gamma_phi"] * th))
        trace.append(dict(it=it, f=nlp.f(x), theta=th, E0=E0, mu=mu, dw=dw, dc=dc, alpha=alpha, alpha_z=a_z, n_ls=n_ls,
                          soc=used_soc, n_fact=n_fact))
        x = x_new
        lam = lam + alpha * dlam
        zL = zL + a_z * dzL; zU = zU + a_z * dzU
        sL, sU = slacks(x)   # eq. (16)
        zL = np.where(bL, np.maximum(np.minimum(zL, o["kappa_sigma"] * mu / sL), mu / (o["kappa_sigma"] * sL)), 0.0)
        zU = np.where(bU, np.maximum(np.minimum(zU, o["kappa_sigma"] * mu / sU), mu / (o["kappa_sigma"] * sU)), 0.0)
        it += 1
    return dict(x=x, iters=it, status=status, mu=mu, trace=trace, restoration=restoration, obj_scaling=df,
                theta=theta(x), f=nlp.f(x))
