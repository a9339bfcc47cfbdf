"""Round 5 (review item 5a): a parallel-in-time backward sweep for the solve's Newton system?  Prototype + cost count.

The Newton system of one interior-point iteration is an LQR problem over N stages (n = 10 states, m = 4 controls):
    min sum_k 1/2 [x;u]' [[Q_k, 0], [0, R_k]] [x;u] + q_k' x + r_k' u,   x_{k+1} = A x_k + B u_k
and the kernel solves it by a SEQUENTIAL Riccati recursion, 20 dependent stages (csrc/mpc_device_impl.h: riccati_backward).
Sarkka & Garcia-Fernandez ("Temporal parallelization of inference / control", 2021-23) write the same value functions as an
ASSOCIATIVE scan over elements (A, b, C, eta, J): V_{i->j}(x_i, x_j) = max_lambda [ 1/2 x_i' J x_i - x_i' eta - 1/2 lambda' C lambda
- lambda' (x_j - A x_i - b) ], combined by
    A_ik = A_jk (I + C_ij J_jk)^-1 A_ij,           b_ik = A_jk (I + C_ij J_jk)^-1 (b_ij + C_ij eta_jk) + b_jk,
    C_ik = A_jk (I + C_ij J_jk)^-1 C_ij A_jk' + C_jk,
    eta_ik = A_ij' (I + J_jk C_ij)^-1 (eta_jk - J_jk b_ij) + eta_ij,   J_ik = A_ij' (I + J_jk C_ij)^-1 J_jk A_ij + J_ij
so that log2(N) levels replace N stages.  This script (i) builds the elements of a random instance with the solver's structure
(the model's A, B; positive-definite stage Hessians), runs the scan as a reverse suffix scan, and checks the value functions (P_k, p_k)
against the sequential Riccati recursion; (ii) counts the floating-point work and the dependent depth of both, as they would map onto
ONE 64-lane wavefront (the unit a scene owns: LDS per scene is what fixes the occupancy).
python tools/experiments/pit_riccati.py"""
import numpy as np

rng = np.random.default_rng(0)
n, m, N = 10, 4, 20


def model(dt=0.033, tau=(6.1, 6.2, 15.8)):
    """the affine quadrotor model's A, B (p <- v <- a chains per axis, yaw <- yaw_dot): one Euler-ish step is enough for the structure"""
    A = np.eye(n); B = np.zeros((n, m))
    for a in range(3):
        A[a, 4 + a] = dt; A[a, 7 + a] = 0.5 * dt * dt; A[4 + a, 7 + a] = dt
        A[7 + a, 7 + a] = 1 - tau[a] * dt; B[7 + a, a] = tau[a] * dt; B[4 + a, a] = 0.5 * tau[a] * dt * dt; B[a, a] = tau[a] * dt ** 3 / 6
    B[3, 3] = dt
    return A, B


def spd(k, scale=1.0):
    M = rng.normal(size=(k, k)); return scale * (M @ M.T / k + 0.1 * np.eye(k))


A, B = model()
Q = [spd(n) for _ in range(N + 1)]; R = [spd(m, 2.0) for _ in range(N)]
q = [rng.normal(size=n) for _ in range(N + 1)]; r = [rng.normal(size=m) for _ in range(N)]

# ---- sequential Riccati (what the kernel does): P_N = Q_N, p_N = q_N; stage k: K = -(R + B'PB)^-1 B'PA ...
P = [None] * (N + 1); p = [None] * (N + 1)
P[N], p[N] = Q[N], q[N]
for k in range(N - 1, -1, -1):
    H = R[k] + B.T @ P[k + 1] @ B
    G = B.T @ P[k + 1] @ A
    g = r[k] + B.T @ p[k + 1]
    P[k] = Q[k] + A.T @ P[k + 1] @ A - G.T @ np.linalg.solve(H, G)
    p[k] = q[k] + A.T @ p[k + 1] - G.T @ np.linalg.solve(H, g)

# ---- the scan: element k (k < N) describes one stage with the control minimised out; element N the terminal cost
def element(k):
    if k == N:
        return np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), -q[N], Q[N]      # V(x_N) = 1/2 x' Q x + q' x  ->  eta = -q
    Ri = np.linalg.inv(R[k])
    return A.copy(), -B @ Ri @ r[k], B @ Ri @ B.T, -q[k], Q[k]


def combine(e1, e2):   # e1 = i -> j (earlier), e2 = j -> k (later)
    A1, b1, C1, h1, J1 = e1; A2, b2, C2, h2, J2 = e2
    I = np.eye(n)
    M = np.linalg.inv(I + C1 @ J2)
    Mt = np.linalg.inv(I + J2 @ C1)
    return (A2 @ M @ A1, A2 @ M @ (b1 + C1 @ h2) + b2, A2 @ M @ C1 @ A2.T + C2,
            A1.T @ Mt @ (h2 - J2 @ b1) + h1, A1.T @ Mt @ J2 @ A1 + J1)


els = [element(k) for k in range(N + 1)]
# reverse suffix scan by recursive doubling (Hillis-Steele): after ceil(log2(N + 1)) levels els[k] = e_k (+) e_{k+1} (+) ... (+) e_N
suf = list(els); d = 1; levels = 0; combines = 0
while d <= N:
    new = list(suf)
    for k in range(N + 1 - d):
        new[k] = combine(suf[k], suf[k + d]); combines += 1
    suf = new; d *= 2; levels += 1
errP = max(np.abs(suf[k][4] - P[k]).max() / np.abs(P[k]).max() for k in range(N + 1))
errp = max(np.abs(-suf[k][3] - p[k]).max() / (1 + np.abs(p[k]).max()) for k in range(N + 1))
print(f"value functions from the scan against the sequential recursion: rel |dP| {errP:.1e}, |dp| {errp:.1e}  ({levels} levels, {combines} combines)")
assert errP < 1e-9 and errp < 1e-9

# ---- cost as it maps onto ONE wavefront (64 lanes, fp64 FMA: 4 issue cycles, ~32 cycles to a dependent issue on MI355X)
mm = lambda a, b, c: 2 * a * b * c                  # flops of an (a x b)(b x c) product
inv = lambda k: 2 * k ** 3                          # LU-based inverse, flops; dependent depth ~ 3 k (k pivots: rcp, scale, update)
comb_flops = 2 * inv(n) + 2 * mm(n, n, n) + 4 * mm(n, n, n) + 3 * mm(n, n, n) + 4 * mm(n, n, 1)
comb_depth = 3 * n + 4 * 4                          # one inverse (the two run side by side) + four chained products of depth log2(10) ~ 4
seq_stage_flops = 2 * 431 + 2 * (46 + 10) * 12      # the kernel's plan: 431 FMA terms in round A + ~12 flops per owned entry in rounds B / C
scan_flops = combines * comb_flops
scan_depth = levels * comb_depth
seq_depth = N * 17                                  # measured structure: ~5 dependent fp64 ops in round A + ~12 in rounds B / C per stage
print(f"sequential sweep : {N * seq_stage_flops / 1e3:6.1f} k flops, dependent depth ~{seq_depth} fp64 ops  (the kernel: 62 VALU instructions per stage, {N * 62} per sweep)")
print(f"associative scan : {scan_flops / 1e3:6.1f} k flops ({combines} combines x {comb_flops} flops), dependent depth ~{scan_depth} fp64 ops over {levels} levels")
print(f"  on one 64-lane wavefront the scan needs >= {scan_flops / 2 / 64:.0f} FMA instructions per lane against {N * 62} today "
      f"({scan_flops / 2 / 64 / (N * 62):.1f} x), for a dependent chain {seq_depth / scan_depth:.1f} x shorter")
print("  -> with the VALU 42 % busy at saturation a {:.1f} x shorter chain cannot pay for ~{:.0f} x the instructions; the scan pays when a scene "
      "owns SEVERAL wavefronts, which the 19.8 KB of LDS per scene (2 waves per SIMD) does not allow".format(seq_depth / scan_depth, scan_flops / 2 / 64 / (N * 62)))
