// Device side of the MPC solve for gfx950: one wavefront (64 lanes) owns one scene for the whole
// interior-point solve -- no host round trips, no inter-workgroup traffic.
//
// Problem (specification: AM/tools/mpc_obstacle_casadi.py:51-242,338-357; packing/bounds/options:
// AM/src/HighLvlMpc.cpp:5-137; AM = roswrapper/ros/src/avoid_mpc in the reference tree):
//   min  sum_k (U_k-u_ref)'Qu(U_k-u_ref) + sum_{k<N-1} [ (R d_k)'Qpen(R d_k) + collide_k ]
//        + (X_N-target)'Qgoal(X_N-target),       d_k = X_{k+1} - ref_k
//   s.t. X_0 = x_init, X_{k+1} = F(X_k,U_k) (RK4 x 4, affine: F = A x + B u + c), lb <= U_k <= ub
//   collide_k = sum_j lam * softplus(-32(|o_kj - p_{k+1}| - r)) * |v_{k+1}.(o_kj - p_{k+1})/|.||
//
// Algorithm (the contract the CPU oracle restates; DESIGN.md "MPC solve"): feasible-start
// primal-dual interior point on the controls; every Newton system is an LQR problem solved by a
// backward Riccati sweep (stage blocks: 10x10 state, 4x4 control) + forward roll; inertia
// correction by a uniform diagonal shift when a 4x4 block is not positive definite; Armijo
// backtracking on the barrier function.  The stacked shooting Jacobian is block-banded with
// 19-nnz / 10-nnz constant blocks, so there is no dense GEMM here and MFMA does not apply; the
// work is 64-lane fp64 VALU with LDS as the per-scene scratchpad.
#pragma once
#include "amk_common.h"

namespace amk {

constexpr int SD = AMK_S_DIM;  // 10
constexpr int UD = AMK_U_DIM;  // 4
constexpr double kGz = 9.81;   // mpc_obstacle_casadi.py:39
constexpr double kAbsEps = 1e-3;

// parameter block in device memory (doubles)
constexpr int PRM_W = 0;        // weights[25]: Qgoal[10] Qpen[10] Qu[4] lambda
constexpr int PRM_TAU = 25;     // tau[4]
constexpr int PRM_GAIN = 29;    // gains[4]  (parsed but unused by the dynamics, :91-93,114-121)
constexpr int PRM_RADIUS = 33;  // drone radius
constexpr int PRM_LB = 34;      // control lower bounds[4]
constexpr int PRM_UB = 38;      // control upper bounds[4]
constexpr int PRM_A = 42;       // A[10][10] row-major
constexpr int PRM_B = 142;      // B[10][4]
constexpr int PRM_C = 182;      // c[10]
constexpr int PRM_LEN = 192;

struct SolveOpts {
    double tol;       // ipopt.tol            HighLvlMpc.cpp:19
    int max_iter;     // ipopt.max_iter       HighLvlMpc.cpp:20
    int max_ls;       // 12
    double mu_init;   // 0.1  (IPOPT default)
    double bound_push, bound_frac;  // 1e-3 (IPOPT warm_start_bound_push / _frac)
    double kappa_mu;  // 0.2
    double tau_min;   // 0.99
    double eta_phi;   // 1e-8
    double s_max;     // 100
    double kappa_sigma;  // 1e10
};

// LDS carve-up for one scene (offsets in doubles).  All per-scene state of the solve lives here.
struct LdsMap {
    int prm, xinit, target, cy, sy;
    int X, U, zl, zu, dX, dU, dzl, dzu, Xt, Ut;
    int q, r, rb, Rb, gU, H6, rotQ;
    int P, p, lam, M, Hm, G, Atp, Atl, qu, Y, D, Kk, red;
    int total;
    __host__ __device__ explicit LdsMap(int N) {
        int o = 0;
        auto take = [&](int n) { int b = o; o += (n + 1) & ~1; return b; };
        prm = take(PRM_LEN); xinit = take(SD); target = take(SD); cy = take(N); sy = take(N);
        X = take((N + 1) * SD); U = take(N * UD); zl = take(N * UD); zu = take(N * UD);
        dX = take((N + 1) * SD); dU = take(N * UD); dzl = take(N * UD); dzu = take(N * UD);
        Xt = take((N + 1) * SD); Ut = take(N * UD);
        q = take((N + 1) * SD); r = take(N * UD); rb = take(N * UD); Rb = take(N * UD); gU = take(N * UD);
        H6 = take(N * 36); rotQ = take(N * 6);
        P = take(100); p = take(SD); lam = take(SD); M = take(56); Hm = take(10); G = take(40);
        Atp = take(SD); Atl = take(SD); qu = take(UD); Y = take(44); D = take(UD + 6 + 2);
        Kk = take(N * 44); red = take(64);
        total = o;
    }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
// sum over aligned segments of `seg` (power of two) consecutive lanes; every lane gets its segment's sum
__device__ __forceinline__ double seg_sum(double v, int seg) {
    for (int off = seg >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// nonzero rows of column j of A / column a of B (fixed by the model: p <- v <- a chains per axis,
// yaw <- yaw_dot; mpc_obstacle_casadi.py:106-122).  Returns count, rows in r[3].
__device__ __forceinline__ int rows_of_A(int j, int r[3]) {
    if (j < 4) { r[0] = j; return 1; }                     // p_x,p_y,p_z,yaw: identity column
    if (j < 7) { r[0] = j - 4; r[1] = j; return 2; }       // v_a: rows p_a, v_a
    r[0] = j - 7; r[1] = j - 3; r[2] = j; return 3;        // a_a: rows p_a, v_a, a_a
}
__device__ __forceinline__ int rows_of_B(int a, int r[3]) {
    if (a == 3) { r[0] = 3; return 1; }                    // yaw_dot -> yaw
    r[0] = a; r[1] = 4 + a; r[2] = 7 + a; return 3;        // a_cmd -> p_a, v_a, a_a
}

// position/velocity slots of the 10-state, and the inverse map (-1 = not in the 6-block)
__device__ __forceinline__ int pv_of(int i) { return i < 3 ? i : i + 1; }   // 0..5 -> {0,1,2,4,5,6}
__device__ __forceinline__ int pv_inv(int s) { return s < 3 ? s : (s >= 4 && s <= 6 ? s - 1 : -1); }

struct SceneIO {
    const double *ref;     // [N][10] reference states            (P[10 : 10+10N])
    const double *obs;     // [N][K][3] obstacle points           (P[10+10N : ...])
};

// ---- one obstacle term: cost and, when DERIV, gradient (6) + model Hessian (21 unique, row-major
// lower triangle of the (p,v) 6x6 block).  Mirrors oracle collide_point statement by statement.
template <bool DERIV>
__device__ __forceinline__ void collide_point(const double p[3], const double v[3], const double o[3], double lam,
                                              double radius, double &cost, double g6[6], double H[21]) {
    const double d0 = o[0] - p[0], d1 = o[1] - p[1], d2 = o[2] - p[2];
    const double rho = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    const double ir = 1.0 / rho;
    const double n[3] = {d0 * ir, d1 * ir, d2 * ir};
    const double s = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
    const double x = -32.0 * (rho - radius);
    const double ex = exp(x);
    const double g = log(1.0 + ex);  // naive softplus, mpc_obstacle_casadi.py:250-251
    const double as = fabs(s);
    cost = lam * g * as;
    if (!DERIV) return;
    const double sg = 1.0 / (1.0 + exp(-x));
    const double gp = -32.0 * sg;
    const double gpp = 1024.0 * sg * (1.0 - sg);
    const double sgn = (s > 0.0) ? 1.0 : ((s < 0.0) ? -1.0 : 0.0);
    const double t[3] = {v[0] - s * n[0], v[1] - s * n[1], v[2] - s * n[2]};
    const double ls = lam * sgn;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g6[i] = ls * (gp * (-n[i]) * s + g * (-t[i] * ir));
        g6[3 + i] = ls * g * n[i];
    }
    const double gs[6] = {-t[0] * ir, -t[1] * ir, -t[2] * ir, n[0], n[1], n[2]};
    const double wk = lam * g / (as > kAbsEps ? as : kAbsEps);
    // lower triangle, row-major: (i,j), j <= i, index i(i+1)/2 + j
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double h = wk * gs[i] * gs[j];
            if (i < 3) {  // pp block
                const double nn = n[i] * n[j];
                const double Pn = (i == j ? 1.0 : 0.0) - nn;
                h += ls * (gpp * nn * s + gp * Pn * ir * s + gp * (n[i] * t[j] * ir + t[i] * ir * n[j]) +
                           g * (-(t[i] * n[j] + n[i] * t[j]) * ir * ir - s * Pn * ir * ir));
            } else if (j < 3) {  // vp block: H[v_i][p_j] = Hpv[j][i-3] (symmetric expression)
                const int a = i - 3;
                const double nn = n[a] * n[j];
                const double Pn = (a == j ? 1.0 : 0.0) - nn;
                h += ls * (-gp * nn - g * Pn * ir);
            }
            H[i * (i + 1) / 2 + j] = h;
        }
}

// Evaluate the objective on (Xs, Us) held in LDS.  DERIV: also q, r, H6 (36 per stage, full
// symmetric), rotQ.  Returns J (wave-uniform).  One wave; caller syncs before/after.
template <bool DERIV>
__device__ double evaluate(double *sm, const LdsMap &L, const SceneIO &io, int N, int K, int Kpad, const double *Xs,
                           const double *Us) {
    const int lane = threadIdx.x;
    const double *prm = sm + L.prm;
    const double lamw = prm[PRM_W + 24], radius = prm[PRM_RADIUS];
    double Jloc = 0.0;
    // ---- collision terms: lanes = (stage, obstacle) with the obstacle index padded to Kpad
    const int spr = 64 / Kpad;
    for (int k0 = 0; k0 < N - 1; k0 += spr) {
        const int k = k0 + lane / Kpad, j = lane % Kpad;
        const bool act = (k < N - 1) && (j < K);
        double c = 0.0, g6[6] = {0, 0, 0, 0, 0, 0}, H[21];
        if (DERIV) {
#pragma unroll
            for (int e = 0; e < 21; ++e) H[e] = 0.0;
        }
        if (act) {
            const double *xk = Xs + (k + 1) * SD;
            const double p[3] = {xk[0], xk[1], xk[2]}, v[3] = {xk[4], xk[5], xk[6]};
            const double *op = io.obs + ((size_t)k * K + j) * 3;
            const double o[3] = {op[0], op[1], op[2]};
            collide_point<DERIV>(p, v, o, lamw, radius, c, g6, H);
        }
        c = seg_sum(c, Kpad);
        if (DERIV) {
#pragma unroll
            for (int e = 0; e < 6; ++e) g6[e] = seg_sum(g6[e], Kpad);
#pragma unroll
            for (int e = 0; e < 21; ++e) H[e] = seg_sum(H[e], Kpad);
        }
        if (j == 0 && k < N - 1) {
            Jloc += c;
            if (DERIV) {
                double *h6 = sm + L.H6 + k * 36;
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int jj = 0; jj <= i; ++jj) {
                        const double h = H[i * (i + 1) / 2 + jj];
                        h6[i * 6 + jj] = h;
                        h6[jj * 6 + i] = h;
                    }
                double *qk = sm + L.q + (k + 1) * SD;  // collision part; the stage lane adds the rest
#pragma unroll
                for (int i = 0; i < 6; ++i) qk[pv_of(i)] = g6[i];
            }
        }
    }
    if (DERIV) __syncthreads();
    // ---- per-stage quadratic terms: lane = stage
    if (lane < N) {
        const int k = lane;
        const double *uk = Us + k * UD;
        const double *xk = Xs + (k + 1) * SD;
        const double uref[4] = {0.0, 0.0, kGz, 0.0};
#pragma unroll
        for (int i = 0; i < UD; ++i) {
            const double du = uk[i] - uref[i];
            const double w = prm[PRM_W + 20 + i];
            Jloc += du * w * du;  // :209-210
            if (DERIV) sm[L.r + k * UD + i] = 2.0 * w * du;
        }
        if (k >= N - 1) {  // goal stage :168-170
            const double *tg = sm + L.target;
#pragma unroll
            for (int i = 0; i < SD; ++i) {
                const double d = xk[i] - tg[i];
                const double w = prm[PRM_W + i];
                Jloc += d * w * d;
                if (DERIV) sm[L.q + (k + 1) * SD + i] = 2.0 * w * d;
            }
        } else {  // path stage :171-208
            const double *rf = io.ref + k * SD;
            const double cy = sm[L.cy + k], sy = sm[L.sy + k];
            double d[SD], y[SD];
#pragma unroll
            for (int i = 0; i < SD; ++i) { d[i] = xk[i] - rf[i]; y[i] = d[i]; }
            y[0] = cy * d[0] - sy * d[1];
            y[1] = sy * d[0] + cy * d[1];
            y[4] = cy * d[4] - sy * d[5];
            y[5] = sy * d[4] + cy * d[5];
            double wy[SD];
#pragma unroll
            for (int i = 0; i < SD; ++i) {
                const double w = prm[PRM_W + 10 + i];
                Jloc += y[i] * w * y[i];
                wy[i] = 2.0 * w * y[i];
            }
            if (DERIV) {
                double qq[SD];
#pragma unroll
                for (int i = 0; i < SD; ++i) qq[i] = wy[i];
                qq[0] = cy * wy[0] + sy * wy[1];
                qq[1] = -sy * wy[0] + cy * wy[1];
                qq[4] = cy * wy[4] + sy * wy[5];
                qq[5] = -sy * wy[4] + cy * wy[5];
                double *qk = sm + L.q + (k + 1) * SD;
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    const bool inpv = (i < 3) || (i >= 4 && i <= 6);
                    qk[i] = qq[i] + (inpv ? qk[i] : 0.0);
                }
            }
        }
    }
    return wave_sum(Jloc);
}

// Q_k[i][j] for the state X_k, 1 <= k <= N (cost stage k-1 attaches to X_k)
__device__ __forceinline__ double q_elem(const double *sm, const LdsMap &L, int N, int k, int i, int j) {
    const double *prm = sm + L.prm;
    if (k >= N) return i == j ? 2.0 * prm[PRM_W + i] : 0.0;  // goal: diag(2 Qgoal)
    const int st = k - 1;
    double v = 0.0;
    const int bi = (i == 0 || i == 1) ? 0 : ((i == 4 || i == 5) ? 1 : -1);
    const int bj = (j == 0 || j == 1) ? 0 : ((j == 4 || j == 5) ? 1 : -1);
    if (bi >= 0 && bi == bj) {
        const int li = (i == 1 || i == 5) ? 1 : 0;  // local index inside the 2x2 block
        const int lj = (j == 1 || j == 5) ? 1 : 0;
        v = sm[L.rotQ + st * 6 + bi * 3 + (li + lj)];   // [xx, xy, yy]
    } else if (i == j) {
        v = 2.0 * prm[PRM_W + 10 + i];
    }
    const int pi = pv_inv(i), pj = pv_inv(j);
    if (pi >= 0 && pj >= 0) v += sm[L.H6 + st * 36 + pi * 6 + pj];
    return v;
}

// Backward Riccati sweep (+ adjoint sweep for the reduced gradient gU).  Returns false when a
// control block is not positive definite.  Gains go to L.Kk ([k][a*11 + j], column 10 = feed-forward).
__device__ bool riccati_backward(double *sm, const LdsMap &L, int N, double delta) {
    const int lane = threadIdx.x;
    const double *A = sm + L.prm + PRM_A, *B = sm + L.prm + PRM_B;
    double *P = sm + L.P, *pv = sm + L.p, *lam = sm + L.lam;
    // terminal: P = Q_N + delta I, p = lam = q_N
    for (int e = lane; e < 100; e += 64) {
        const int i = e / 10, j = e % 10;
        P[e] = q_elem(sm, L, N, N, i, j) + (i == j ? delta : 0.0);
    }
    if (lane < SD) {
        pv[lane] = sm[L.q + N * SD + lane];
        lam[lane] = pv[lane];
    }
    __syncthreads();
    for (int k = N - 1; k >= 0; --k) {
        // ---- round A: M = A'PA (55 lower), Hm = B'PB + Rb (10 lower), G = B'PA (40), A'p, A'lam, qu, gU
        for (int e = lane; e < 119; e += 64) {
            if (e < 105) {
                int ri[3], rj[3], ni, nj, i, j;
                const double *Li, *Lj;
                int si, sj;  // strides of the left/right factor matrices
                if (e < 55) {  // M(i,j), j <= i
                    i = 0;
                    while ((i + 1) * (i + 2) / 2 <= e) ++i;
                    j = e - i * (i + 1) / 2;
                    ni = rows_of_A(i, ri); nj = rows_of_A(j, rj);
                    Li = A; si = SD; Lj = A; sj = SD;
                } else if (e < 65) {  // Hm(a,b), b <= a
                    const int f = e - 55;
                    i = 0;
                    while ((i + 1) * (i + 2) / 2 <= f) ++i;
                    j = f - i * (i + 1) / 2;
                    ni = rows_of_B(i, ri); nj = rows_of_B(j, rj);
                    Li = B; si = UD; Lj = B; sj = UD;
                } else {  // G(a,j)
                    const int f = e - 65;
                    i = f / 10; j = f % 10;
                    ni = rows_of_B(i, ri); nj = rows_of_A(j, rj);
                    Li = B; si = UD; Lj = A; sj = SD;
                }
                double acc = 0.0;
                for (int a = 0; a < ni; ++a) {
                    const int l = ri[a];
                    double inner = 0.0;
                    for (int b = 0; b < nj; ++b) inner += P[l * 10 + rj[b]] * Lj[rj[b] * sj + j];
                    acc += Li[l * si + i] * inner;
                }
                if (e < 55) sm[L.M + e] = acc;
                else if (e < 65) sm[L.Hm + (e - 55)] = acc + ((i == j) ? sm[L.Rb + k * UD + i] + delta : 0.0);
                else sm[L.G + (e - 65)] = acc;
            } else if (e < 115) {
                const int i = e - 105;
                int ri[3];
                const int ni = rows_of_A(i, ri);
                double a1 = 0.0, a2 = 0.0;
                for (int a = 0; a < ni; ++a) {
                    a1 += A[ri[a] * SD + i] * pv[ri[a]];
                    a2 += A[ri[a] * SD + i] * lam[ri[a]];
                }
                sm[L.Atp + i] = a1;
                sm[L.Atl + i] = a2;
            } else {
                const int a = e - 115;
                int ri[3];
                const int ni = rows_of_B(a, ri);
                double a1 = 0.0, a2 = 0.0;
                for (int b = 0; b < ni; ++b) {
                    a1 += B[ri[b] * UD + a] * pv[ri[b]];
                    a2 += B[ri[b] * UD + a] * lam[ri[b]];
                }
                sm[L.qu + a] = sm[L.rb + k * UD + a] + a1;
                sm[L.gU + k * UD + a] = sm[L.r + k * UD + a] + a2;
            }
        }
        __syncthreads();
        // ---- round B: LDL' of Hm (every lane, wave-uniform), Y = L^-1 [G | qu], gains
        double Lm[6], Dm[4];  // L10 L20 L21 L30 L31 L32
        {
            const double *h = sm + L.Hm;  // lower: h00 h10 h11 h20 h21 h22 h30 h31 h32 h33
            const double d0 = h[0];
            if (!(d0 > 0.0)) return false;
            Lm[0] = h[1] / d0; Lm[1] = h[3] / d0; Lm[3] = h[6] / d0;
            const double d1 = h[2] - Lm[0] * Lm[0] * d0;
            if (!(d1 > 0.0)) return false;
            Lm[2] = (h[4] - Lm[1] * Lm[0] * d0) / d1;
            Lm[4] = (h[7] - Lm[3] * Lm[0] * d0) / d1;
            const double d2 = h[5] - Lm[1] * Lm[1] * d0 - Lm[2] * Lm[2] * d1;
            if (!(d2 > 0.0)) return false;
            Lm[5] = (h[8] - Lm[3] * Lm[1] * d0 - Lm[4] * Lm[2] * d1) / d2;
            const double d3 = h[9] - Lm[3] * Lm[3] * d0 - Lm[4] * Lm[4] * d1 - Lm[5] * Lm[5] * d2;
            if (!(d3 > 0.0)) return false;
            Dm[0] = d0; Dm[1] = d1; Dm[2] = d2; Dm[3] = d3;
        }
        if (lane <= SD) {
            const int j = lane;  // column j of G, or j == 10: qu
            const double g0 = j < SD ? sm[L.G + j] : sm[L.qu + 0];
            const double g1 = j < SD ? sm[L.G + 10 + j] : sm[L.qu + 1];
            const double g2 = j < SD ? sm[L.G + 20 + j] : sm[L.qu + 2];
            const double g3 = j < SD ? sm[L.G + 30 + j] : sm[L.qu + 3];
            const double y0 = g0;
            const double y1 = g1 - Lm[0] * y0;
            const double y2 = g2 - Lm[1] * y0 - Lm[2] * y1;
            const double y3 = g3 - Lm[3] * y0 - Lm[4] * y1 - Lm[5] * y2;
            sm[L.Y + j] = y0; sm[L.Y + 11 + j] = y1; sm[L.Y + 22 + j] = y2; sm[L.Y + 33 + j] = y3;
            // x = L^-T D^-1 y ; gain = -x
            const double x3 = y3 / Dm[3];
            const double x2 = y2 / Dm[2] - Lm[5] * x3;
            const double x1 = y1 / Dm[1] - Lm[2] * x2 - Lm[4] * x3;
            const double x0 = y0 / Dm[0] - Lm[0] * x1 - Lm[1] * x2 - Lm[3] * x3;
            double *kk = sm + L.Kk + k * 44;
            kk[j] = -x0; kk[11 + j] = -x1; kk[22 + j] = -x2; kk[33 + j] = -x3;
        }
        if (lane == 0) {
            sm[L.D + 0] = Dm[0]; sm[L.D + 1] = Dm[1]; sm[L.D + 2] = Dm[2]; sm[L.D + 3] = Dm[3];
        }
        __syncthreads();
        // ---- round C: P_k = Q_k + delta I + M - Y'D^-1 Y ; p_k = q_k + A'p - Y'D^-1 yv ; lam_k = q_k + A'lam
        if (k > 0) {
            const double *Y = sm + L.Y;
            const double i0 = 1.0 / Dm[0], i1 = 1.0 / Dm[1], i2 = 1.0 / Dm[2], i3 = 1.0 / Dm[3];
            for (int e = lane; e < 55 + SD; e += 64) {  // 65 work items: two rounds on 64 lanes
                if (e < 55) {
                    int i = 0;
                    while ((i + 1) * (i + 2) / 2 <= e) ++i;
                    const int j = e - i * (i + 1) / 2;
                    const double ww = Y[i] * Y[j] * i0 + Y[11 + i] * Y[11 + j] * i1 + Y[22 + i] * Y[22 + j] * i2 +
                                      Y[33 + i] * Y[33 + j] * i3;
                    const double val = q_elem(sm, L, N, k, i, j) + (i == j ? delta : 0.0) + sm[L.M + e] - ww;
                    P[i * 10 + j] = val;
                    P[j * 10 + i] = val;
                } else {
                    const int i = e - 55;
                    const double wv = Y[i] * Y[10] * i0 + Y[11 + i] * Y[21] * i1 + Y[22 + i] * Y[32] * i2 +
                                      Y[33 + i] * Y[43] * i3;
                    const double qk = sm[L.q + k * SD + i];
                    pv[i] = qk + sm[L.Atp + i] - wv;
                    lam[i] = qk + sm[L.Atl + i];
                }
            }
        }
        __syncthreads();
    }
    return true;
}

// forward roll of the Newton step: dX_0 = 0, dU_k = K_k dX_k + d_k, dX_{k+1} = A dX_k + B dU_k
__device__ void riccati_forward(double *sm, const LdsMap &L, int N) {
    const int lane = threadIdx.x;
    const double *A = sm + L.prm + PRM_A, *B = sm + L.prm + PRM_B;
    double *dX = sm + L.dX, *dU = sm + L.dU;
    if (lane < SD) dX[lane] = 0.0;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        if (lane < UD) {
            const double *kk = sm + L.Kk + k * 44 + lane * 11;
            double a = kk[10];
#pragma unroll
            for (int j = 0; j < SD; ++j) a += kk[j] * dX[k * SD + j];
            dU[k * UD + lane] = a;
        }
        __syncthreads();
        if (lane < SD) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < SD; ++j) a += A[lane * SD + j] * dX[k * SD + j];
#pragma unroll
            for (int j = 0; j < UD; ++j) a += B[lane * UD + j] * dU[k * UD + j];
            dX[(k + 1) * SD + lane] = a;
        }
        __syncthreads();
    }
}

// The whole solve for one scene.  w0/w_out: decision vector [X_0,U_0,...,U_{N-1},X_N] in global
// memory (warm start in, solution out; may alias).  info[4] as in the C ABI.
__device__ void solve_scene(double *sm, const LdsMap &L, int N, int K, const double *prm_g, const SolveOpts &opt,
                            const double *x_init, const double *target, const SceneIO &io, const double *w0,
                            double *w_out, int *info, double *trace = nullptr) {
    const int lane = threadIdx.x;
    int Kpad = 1;
    while (Kpad < K) Kpad <<= 1;
    for (int e = lane; e < PRM_LEN; e += 64) sm[L.prm + e] = prm_g[e];
    if (lane < SD) {
        sm[L.xinit + lane] = x_init[lane];
        sm[L.target + lane] = target[lane];
    }
    if (lane < N - 1) {  // rot of ref yaw, :174-185
        const double yaw = io.ref[lane * SD + 3];
        sm[L.cy + lane] = cos(yaw);
        sm[L.sy + lane] = sin(-yaw);
    }
    __syncthreads();
    const double *prm = sm + L.prm;
    if (lane < N - 1) {  // constant part of Q on the rotated (px,py) and (vx,vy) blocks
        const double cy = sm[L.cy + lane], sy = sm[L.sy + lane];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double w0q = 2.0 * prm[PRM_W + 10 + 4 * b], w1q = 2.0 * prm[PRM_W + 11 + 4 * b];
            sm[L.rotQ + lane * 6 + b * 3 + 0] = cy * cy * w0q + sy * sy * w1q;
            sm[L.rotQ + lane * 6 + b * 3 + 1] = -cy * sy * w0q + sy * cy * w1q;
            sm[L.rotQ + lane * 6 + b * 3 + 2] = sy * sy * w0q + cy * cy * w1q;
        }
    }
    double mu = opt.mu_init;
    // warm start pushed into the interior; duals on the central path
    for (int e = lane; e < N * UD; e += 64) {
        const int k = e / UD, i = e % UD;
        const double lb = prm[PRM_LB + i], ub = prm[PRM_UB + i];
        const double pl = fmin(opt.bound_push * fmax(1.0, fabs(lb)), opt.bound_frac * (ub - lb));
        const double pu = fmin(opt.bound_push * fmax(1.0, fabs(ub)), opt.bound_frac * (ub - lb));
        double u = w0[14 * k + 10 + i];
        u = fmin(fmax(u, lb + pl), ub - pu);
        sm[L.U + e] = u;
        sm[L.zl + e] = mu / (u - lb);
        sm[L.zu + e] = mu / (ub - u);
    }
    if (lane < SD) sm[L.X + lane] = sm[L.xinit + lane];
    __syncthreads();
    {  // rollout X_{k+1} = A X_k + B U_k + c
        const double *A = prm + PRM_A, *B = prm + PRM_B, *c = prm + PRM_C;
        for (int k = 0; k < N; ++k) {
            if (lane < SD) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < SD; ++j) a += A[lane * SD + j] * sm[L.X + k * SD + j];
#pragma unroll
                for (int j = 0; j < UD; ++j) a += B[lane * UD + j] * sm[L.U + k * UD + j];
                sm[L.X + (k + 1) * SD + lane] = a + c[lane];
            }
            __syncthreads();
        }
    }
    const double mu_min = opt.tol / 10.0;
    double delta_last = 0.0, a_last = 0.0;
    int status = 1, n_reg = 0, ls_fail = 0, it = 0;
    const int nvar = UD * N;
    for (it = 0; it < opt.max_iter; ++it) {
        const double J = evaluate<true>(sm, L, io, N, K, Kpad, sm + L.X, sm + L.U);
        __syncthreads();
        if (it > 0 && a_last >= 0.5) mu = fmax(mu_min, opt.kappa_mu * mu);
        const double tau = fmax(opt.tau_min, 1.0 - mu);
        for (int e = lane; e < nvar; e += 64) {
            const int i = e % UD;
            const double u = sm[L.U + e];
            const double sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
            sm[L.rb + e] = sm[L.r + e] - mu / sl + mu / su;
            sm[L.Rb + e] = 2.0 * prm[PRM_W + 20 + i] + sm[L.zl + e] / sl + sm[L.zu + e] / su;
        }
        __syncthreads();
        double delta = 0.0;
        int reg_now = 0;
        bool ok = riccati_backward(sm, L, N, delta);
        while (!ok) {
            __syncthreads();
            if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
            else delta *= (delta_last == 0.0) ? 100.0 : 8.0;
            ++reg_now;
            if (delta > 1e40) break;
            ok = riccati_backward(sm, L, N, delta);
        }
        if (!ok) { status = 2; break; }
        // KKT error E_0 at the current iterate (gU from the adjoint sweep inside the backward pass)
        {
            double zs = 0.0, ed = 0.0, ec = 0.0;
            for (int e = lane; e < nvar; e += 64) {
                const int i = e % UD;
                const double u = sm[L.U + e], zl = sm[L.zl + e], zu = sm[L.zu + e];
                const double sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
                zs += zl + zu;
                ed = fmax(ed, fabs(sm[L.gU + e] - zl + zu));
                ec = fmax(ec, fmax(fabs(sl * zl), fabs(su * zu)));
            }
            zs = wave_sum(zs); ed = wave_max(ed); ec = wave_max(ec);
            const double s_d = fmax(opt.s_max, zs / (2.0 * nvar)) / opt.s_max;
            if (trace && lane == 0) {
                trace[8 * it + 0] = J; trace[8 * it + 1] = fmax(ed, ec) / s_d; trace[8 * it + 2] = mu;
                trace[8 * it + 3] = delta;
            }
            if (fmax(ed, ec) / s_d <= opt.tol) { status = 0; break; }
        }
        n_reg += reg_now;
        if (delta > 0.0) delta_last = delta;
        riccati_forward(sm, L, N);
        // dual steps, fraction to the boundary, directional derivative, barrier value
        double a_pr = 1.0, a_du = 1.0, dphi = 0.0, phi0 = 0.0;
        for (int e = lane; e < nvar; e += 64) {
            const int i = e % UD;
            const double u = sm[L.U + e], zl = sm[L.zl + e], zu = sm[L.zu + e], du = sm[L.dU + e];
            const double sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
            const double dzl = mu / sl - zl - (zl / sl) * du;
            const double dzu = mu / su - zu + (zu / su) * du;
            sm[L.dzl + e] = dzl;
            sm[L.dzu + e] = dzu;
            if (du < 0.0) a_pr = fmin(a_pr, -tau * sl / du);
            if (du > 0.0) a_pr = fmin(a_pr, tau * su / du);
            if (dzl < 0.0) a_du = fmin(a_du, -tau * zl / dzl);
            if (dzu < 0.0) a_du = fmin(a_du, -tau * zu / dzu);
            dphi += (sm[L.gU + e] - mu / sl + mu / su) * du;
            phi0 -= mu * (log(sl) + log(su));
        }
        a_pr = wave_min(a_pr); a_du = wave_min(a_du); dphi = wave_sum(dphi); phi0 = J + wave_sum(phi0);
        // backtracking Armijo line search on the barrier function
        double a = a_pr;
        bool accepted = false;
        for (int ls = 0; ls < opt.max_ls; ++ls) {
            __syncthreads();
            for (int e = lane; e < nvar; e += 64) sm[L.Ut + e] = sm[L.U + e] + a * sm[L.dU + e];
            for (int e = lane; e < (N + 1) * SD; e += 64) sm[L.Xt + e] = sm[L.X + e] + a * sm[L.dX + e];
            __syncthreads();
            double phi = evaluate<false>(sm, L, io, N, K, Kpad, sm + L.Xt, sm + L.Ut);
            double lg = 0.0;
            for (int e = lane; e < nvar; e += 64) {
                const int i = e % UD;
                const double u = sm[L.Ut + e];
                lg -= mu * (log(u - prm[PRM_LB + i]) + log(prm[PRM_UB + i] - u));
            }
            phi += wave_sum(lg);
            if (phi <= phi0 + opt.eta_phi * a * dphi) { accepted = true; break; }
            if (ls + 1 < opt.max_ls) a *= 0.5;
        }
        if (!accepted) ++ls_fail;
        a_last = accepted ? a : 0.0;
        if (trace && lane == 0) {
            trace[8 * it + 4] = a; trace[8 * it + 5] = a_pr; trace[8 * it + 6] = a_du; trace[8 * it + 7] = dphi;
        }
        __syncthreads();
        for (int e = lane; e < (N + 1) * SD; e += 64) sm[L.X + e] = sm[L.Xt + e];
        for (int e = lane; e < nvar; e += 64) {
            const int i = e % UD;
            const double u = sm[L.Ut + e];
            sm[L.U + e] = u;
            const double sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
            double zl = sm[L.zl + e] + a_du * sm[L.dzl + e], zu = sm[L.zu + e] + a_du * sm[L.dzu + e];
            zl = fmax(fmin(zl, opt.kappa_sigma * mu / sl), mu / (opt.kappa_sigma * sl));
            zu = fmax(fmin(zu, opt.kappa_sigma * mu / su), mu / (opt.kappa_sigma * su));
            sm[L.zl + e] = zl;
            sm[L.zu + e] = zu;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int e = lane; e < (N + 1) * SD; e += 64) w_out[14 * (e / SD) + (e % SD)] = sm[L.X + e];
    for (int e = lane; e < nvar; e += 64) w_out[14 * (e / UD) + 10 + (e % UD)] = sm[L.U + e];
    if (lane == 0) {  // kept for the control-step bookkeeping of the calling kernel
        sm[L.red + 0] = (double)status;
        sm[L.red + 1] = (double)it;
    }
    if (info && lane == 0) {
        info[0] = status;
        info[1] = it;
        info[2] = n_reg;
        info[3] = ls_fail;
    }
}

__device__ __forceinline__ int sm_status(const double *sm, const LdsMap &L) { return (int)sm[L.red + 0]; }
__device__ __forceinline__ int sm_iters(const double *sm, const LdsMap &L) { return (int)sm[L.red + 1]; }

}  // namespace amk
