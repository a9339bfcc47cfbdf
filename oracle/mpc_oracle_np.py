"""NumPy restatement of the Avoid-MPC NLP (values, derivatives, constraint Jacobian).  TEST INFRASTRUCTURE ONLY.

This file is part of the oracle: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product path (HIP kernels behind include/avoid_mpc_amd.h)
never touches it.

PARITY UNPINNED: the reference's MPC arithmetic lives in CasADi 3.6.4 + IPOPT + MUMPS
(README.md:41-43), none of which is in /root/reference or in this image.  What is restated here is
the *specification* of the NLP as written in
    AM/tools/mpc_obstacle_casadi.py   (AM = roswrapper/ros/src/avoid_mpc)
and the packing / bounds / options in AM/src/HighLvlMpc.cpp.  It is pinned by substitutes only:
finite differences, a scipy.optimize converged optimum, and the reference's own smoke scenario
(mpc_obstacle_casadi.py:448-498) -- see tests/test_mpc_oracle.py.

Layouts:
  P  = [x_init(10) | ref_k(10)*N | obstacles(3*K*N) | target(10) | gain(4) | tau(4) | weights(25) | r]
  w  = [X_0, U_0, X_1, U_1, ..., U_{N-1}, X_N]      (nx = 10 + 14 N)
  g  = [X_0 - x_init ; F(X_k,U_k) - X_{k+1}]        (ng = 10 + 10 N)
"""
import numpy as np

S_DIM = 10
U_DIM = 4
GZ = 9.81  # mpc_obstacle_casadi.py:39


# ----------------------------------------------------------------------------------------------
# parameter vector helpers
# ----------------------------------------------------------------------------------------------
def p_len(N, K):
    # mpc_obstacle_casadi.py:76-85
    return S_DIM + S_DIM * N + K * 3 * N + S_DIM + U_DIM * 2 + 25 + 1


def split_p(P, N, K):
    """Slices of P exactly as mpc_obstacle_casadi.py:88-149 takes them."""
    P = np.asarray(P, dtype=np.float64)
    assert P.shape[0] == p_len(N, K)
    o0 = S_DIM + S_DIM * N
    t0 = o0 + 3 * K * N
    out = dict(
        x_init=P[0:S_DIM],
        ref=P[S_DIM:o0].reshape(N, S_DIM),
        obs=P[o0:t0].reshape(N, K, 3),
        target=P[t0:t0 + S_DIM],
        gain=P[-34:-30],
        tau=P[-30:-26],
        weights=P[-26:-1],
        radius=P[-1],
    )
    w = out["weights"]
    out["q_goal"] = w[0:10]
    out["q_pen"] = w[10:20]
    out["q_u"] = w[20:24]
    out["lam"] = w[24]
    return out


# ----------------------------------------------------------------------------------------------
# dynamics  (mpc_obstacle_casadi.py:106-122 ode, :338-357 RK4 x 4)
# ----------------------------------------------------------------------------------------------
DRAG = np.zeros(3)   # v' = a - DRAG * v (mpc_obstacle_casadi.py:95-105 read as matrix products: R (k I) R' v = k v; see mpc_oracle.c)


def ode(x, u, tau):
    return np.array([
        x[4], x[5], x[6],
        u[3],
        x[7] - DRAG[0] * x[4], x[8] - DRAG[1] * x[5], x[9] - DRAG[2] * x[6],
        (u[0] - x[7]) * tau[0],
        (u[1] - x[8]) * tau[1],
        (u[2] - GZ - x[9]) * tau[2],
    ])


def rk4_step(x, u, tau, dt):
    M = 4
    DT = dt / M
    X = np.array(x, dtype=np.float64)
    for _ in range(M):
        k1 = DT * ode(X, u, tau)
        k2 = DT * ode(X + 0.5 * k1, u, tau)
        k3 = DT * ode(X + 0.5 * k2, u, tau)
        k4 = DT * ode(X + k3, u, tau)
        X = X + (k1 + 2 * k2 + 2 * k3 + k4) / 6
    return X


def affine_dynamics(tau, dt):
    """F(x,u) = A x + B u + c  (drag off => exactly affine).  Returned by probing rk4_step."""
    z10, z4 = np.zeros(S_DIM), np.zeros(U_DIM)
    c = rk4_step(z10, z4, tau, dt)
    A = np.zeros((S_DIM, S_DIM))
    B = np.zeros((S_DIM, U_DIM))
    for j in range(S_DIM):
        e = z10.copy(); e[j] = 1.0
        A[:, j] = rk4_step(e, z4, tau, dt) - c
    for j in range(U_DIM):
        e = z4.copy(); e[j] = 1.0
        B[:, j] = rk4_step(z10, e, tau, dt) - c
    return A, B, c


# ----------------------------------------------------------------------------------------------
# objective pieces
# ----------------------------------------------------------------------------------------------
def softplus(x):
    return np.log(1.0 + np.exp(x))  # naive form, mpc_obstacle_casadi.py:250-251


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def rot_of_ref(yaw_ref):
    """mpc_obstacle_casadi.py:174-185: identity with 2x2 blocks on (px,py) and (vx,vy)."""
    c = np.cos(yaw_ref)
    s = np.sin(-yaw_ref)
    R = np.eye(S_DIM)
    for a in (0, 4):
        R[a, a] = c
        R[a, a + 1] = -s
        R[a + 1, a] = s
        R[a + 1, a + 1] = c
    return R


ABS_EPS = 1e-3


def collide_terms(p, v, obs, lam, radius, want_derivs, majorise_abs=True):
    """Sum over the K obstacle points of one stage (mpc_obstacle_casadi.py:186-204).

    returns cost, grad(6: p then v), hess(6x6).  With majorise_abs=False the Hessian is the plain
    second derivative with sign(s) frozen (what CasADi's AD of fabs gives); the solver's model
    Hessian adds the majoriser curvature of the abs term."""
    cost = 0.0
    g6 = np.zeros(6)
    H6 = np.zeros((6, 6))
    I3 = np.eye(3)
    for o in obs:
        d = o - p
        rho = np.sqrt(d @ d)
        n = d / rho
        s = v @ n
        x = -32.0 * (rho - radius)
        g = softplus(x)
        cost += lam * g * abs(s)
        if not want_derivs:
            continue
        sg = sigmoid(x)
        gp = -32.0 * sg
        gpp = 1024.0 * sg * (1.0 - sg)
        sgn = np.sign(s)
        t = v - s * n
        # gradient
        gp_vec = lam * sgn * (gp * (-n) * s + g * (-t / rho))
        gv_vec = lam * sgn * g * n
        g6[0:3] += gp_vec
        g6[3:6] += gv_vec
        # hessian
        Pn = I3 - np.outer(n, n)
        Hpp = (gpp * np.outer(n, n) * s
               + gp * (Pn / rho) * s
               + gp * (np.outer(-n, -t / rho) + np.outer(-t / rho, -n))
               + g * (-(np.outer(t, n) + np.outer(n, t)) / rho ** 2 - s * Pn / rho ** 2))
        Hpv = -gp * np.outer(n, n) - g * Pn / rho
        H6[0:3, 0:3] += lam * sgn * Hpp
        H6[0:3, 3:6] += lam * sgn * Hpv
        H6[3:6, 0:3] += lam * sgn * Hpv.T
        if majorise_abs:
            # curvature of the quadratic majoriser of |s| (see oracle/mpc_oracle.c collide_point)
            gs = np.concatenate([-t / rho, n])
            H6 += lam * g / max(abs(s), ABS_EPS) * np.outer(gs, gs)
    return cost, g6, H6


PV = np.array([0, 1, 2, 4, 5, 6])  # positions then velocities inside the 10-state


def stage_cost(k, N, xk1, uk, pp, want_derivs=True, majorise_abs=True):
    """Cost attached to (U_k, X_{k+1}); returns (cost, q(10), Q(10x10), r(4), Rdiag(4))."""
    u_ref = np.array([0.0, 0.0, GZ, 0.0])
    du = uk - u_ref
    cost = float(du @ (pp["q_u"] * du))                       # :209-210
    r = 2.0 * pp["q_u"] * du
    Rd = 2.0 * pp["q_u"]
    q = np.zeros(S_DIM)
    Q = np.zeros((S_DIM, S_DIM))
    if k >= N - 1:                                             # :168-170 goal stage
        d = xk1 - pp["target"]
        cost += float(d @ (pp["q_goal"] * d))
        q = 2.0 * pp["q_goal"] * d
        Q = np.diag(2.0 * pp["q_goal"])
    else:                                                      # :171-208
        ref = pp["ref"][k]
        R = rot_of_ref(ref[3])
        d = xk1 - ref
        y = R @ d
        cost += float(y @ (pp["q_pen"] * y))
        q = 2.0 * R.T @ (pp["q_pen"] * y)
        Q = 2.0 * R.T @ np.diag(pp["q_pen"]) @ R
        c, g6, H6 = collide_terms(xk1[0:3], xk1[4:7], pp["obs"][k], pp["lam"], pp["radius"],
                                  want_derivs, majorise_abs)
        cost += c
        if want_derivs:
            q = q.copy()
            q[PV] += g6
            Q = Q.copy()
            Q[np.ix_(PV, PV)] += H6
    return cost, q, Q, r, Rd


# ----------------------------------------------------------------------------------------------
# the five plugin functions (multiple-shooting form) -- dense, for testing
# ----------------------------------------------------------------------------------------------
def unpack_w(w, N):
    w = np.asarray(w, dtype=np.float64)
    X = np.zeros((N + 1, S_DIM))
    U = np.zeros((N, U_DIM))
    for k in range(N):
        X[k] = w[14 * k:14 * k + 10]
        U[k] = w[14 * k + 10:14 * k + 14]
    X[N] = w[14 * N:14 * N + 10]
    return X, U


def pack_w(X, U):
    N = U.shape[0]
    w = np.zeros(10 + 14 * N)
    for k in range(N):
        w[14 * k:14 * k + 10] = X[k]
        w[14 * k + 10:14 * k + 14] = U[k]
    w[14 * N:] = X[N]
    return w


def nlp_f(w, P, N, K):
    pp = split_p(P, N, K)
    X, U = unpack_w(w, N)
    return sum(stage_cost(k, N, X[k + 1], U[k], pp, False)[0] for k in range(N))


def nlp_grad_f(w, P, N, K):
    pp = split_p(P, N, K)
    X, U = unpack_w(w, N)
    gX = np.zeros_like(X)
    gU = np.zeros_like(U)
    for k in range(N):
        _, q, _, r, _ = stage_cost(k, N, X[k + 1], U[k], pp)
        gX[k + 1] = q
        gU[k] = r
    return pack_w(gX, gU)


def nlp_hess_f(w, P, N, K, majorise_abs=False):
    """Dense Hessian of f.  majorise_abs=False: plain second derivatives (sign frozen), the thing
    finite differences can check; True: the solver's model Hessian."""
    pp = split_p(P, N, K)
    X, U = unpack_w(w, N)
    nx = 10 + 14 * N
    H = np.zeros((nx, nx))
    for k in range(N):
        _, _, Q, _, Rd = stage_cost(k, N, X[k + 1], U[k], pp, True, majorise_abs)
        ix = slice(14 * (k + 1), 14 * (k + 1) + 10)
        iu = slice(14 * k + 10, 14 * k + 14)
        H[ix, ix] = Q
        H[iu, iu] = np.diag(Rd)
    return H


def nlp_g(w, P, N, K, dt):
    pp = split_p(P, N, K)
    X, U = unpack_w(w, N)
    g = [X[0] - pp["x_init"]]
    for k in range(N):
        g.append(rk4_step(X[k], U[k], pp["tau"], dt) - X[k + 1])
    return np.concatenate(g)


def nlp_jac_g(w, P, N, K, dt):
    pp = split_p(P, N, K)
    A, B, _ = affine_dynamics(pp["tau"], dt)
    nx = 10 + 14 * N
    J = np.zeros((10 + 10 * N, nx))
    J[0:10, 0:10] = np.eye(10)
    for k in range(N):
        rows = slice(10 + 10 * k, 20 + 10 * k)
        J[rows, 14 * k:14 * k + 10] = A
        J[rows, 14 * k + 10:14 * k + 14] = B
        J[rows, 14 * (k + 1):14 * (k + 1) + 10] = -np.eye(10)
    return J


# ----------------------------------------------------------------------------------------------
# condensed form (the shooting defects eliminated: X = rollout(U)), used by the fixture generators to hand the
# problem to scipy.optimize as an independent optimiser
# ----------------------------------------------------------------------------------------------
def rollout(x0, U, A, B, c):
    N = U.shape[0]
    X = np.zeros((N + 1, S_DIM))
    X[0] = x0
    for k in range(N):
        X[k + 1] = A @ X[k] + B @ U[k] + c
    return X


def total_cost(X, U, pp, N, derivs):
    J = 0.0
    q = np.zeros((N + 1, S_DIM)); Q = np.zeros((N + 1, S_DIM, S_DIM))
    r = np.zeros((N, U_DIM)); Rd = np.zeros((N, U_DIM))
    for k in range(N):
        ck, qk, Qk, rk, Rdk = stage_cost(k, N, X[k + 1], U[k], pp, derivs)
        J += ck
        q[k + 1], Q[k + 1], r[k], Rd[k] = qk, Qk, rk, Rdk
    return J, q, Q, r, Rd
