// The NLP functions of the reference's generated solver plugin, evaluated on the device at caller-supplied points.
//
// The reference's plugin (AM/tools/mpc_obstacle_casadi.py:290-300 -> so/mpc_obstacle_v2.so, loaded at
// AM/src/HighLvlMpc.cpp:50,52) exports nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l of the multiple-shooting NLP
//   x = [X_0, U_0, X_1, ..., U_{N-1}, X_N]                                  mpc_obstacle_casadi.py:158,164,217,224
//   f = objective                                                            :153-214
//   g = [X_0 - P[0:10] ; F(X_k, U_k) - X_{k+1}]                              :156-160,219
// amk_mpc_eval computes all of them for a batch of scenes; casadi_plugin.hip wraps
// it in the plugin's own C ABI.  One wavefront per scene: the decision vector is loaded into LDS and the objective
// pass of the solver (mpc_device_impl.h evaluate<true, EXACT>) runs on it in the plugin's form -- |s| itself, its
// derivative sign(s), no curvature from the abs() (CasADi's SX rule for fabs).  The constraints are linear in x
// (F = A x + B u + c with the drag term off, mpc_parameters.yaml:4), so jac_g is constant and hess_l = lam_f hess f.
#include "mpc_handle.h"

#include <cstring>
#include <vector>

using namespace amk;

namespace {

// order of the 25 structural non-zeros of the upper triangle of the Hessian block of X_k (1 <= k <= N-1), column-major,
// rows ascending inside a column (CCS): the 6x6 (p,v) block is dense, yaw and the accelerations only have a diagonal
__device__ __constant__ signed char kHessRow[25] = {0, 0, 1, 0, 1, 2, 3, 0, 1, 2, 4, 0, 1, 2, 4, 5, 0, 1, 2, 4, 5, 6, 7, 8, 9};
__device__ __constant__ signed char kHessCol[25] = {0, 1, 1, 2, 2, 2, 3, 4, 4, 4, 4, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 7, 8, 9};
const int kHessRowH[25] = {0, 0, 1, 0, 1, 2, 3, 0, 1, 2, 4, 0, 1, 2, 4, 5, 0, 1, 2, 4, 5, 6, 7, 8, 9};
const int kHessColH[25] = {0, 1, 1, 2, 2, 2, 3, 4, 4, 4, 4, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 7, 8, 9};

template <int NT>
__global__ __launch_bounds__(64) void mpc_eval_kernel(int Nrt, int K, int nref, int nx, const double *__restrict__ prm,
                                                      const double *__restrict__ w_all, const double *__restrict__ ref_states,
                                                      const double *__restrict__ lam_f, double *__restrict__ f_out,
                                                      double *__restrict__ grad_out, double *__restrict__ g_out,
                                                      double *__restrict__ jac_out, double *__restrict__ hess_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    double *sm = reinterpret_cast<double *>(sm_raw);
    const int s = blockIdx.x, lane = threadIdx.x;
    const int N = NT > 0 ? NT : Nrt;
    const LdsMap L(N, PRM_LEN);
    const double *P = ref_states + (size_t)s * nref;
    const double *w = w_all + (size_t)s * nx;
    SceneIO io;
    io.ref = P + SD;
    io.obs = P + SD + SD * N;
    const double *target = P + SD + SD * N + 3 * K * N;
    for (int e = lane; e < PRM_LEN; e += 64) sm[L.prm + e] = prm[e];
    if (lane < SD) {
        sm[L.xinit + lane] = P[lane];
        sm[L.target + lane] = target[lane];
    }
    if (lane < N - 1) {  // rot of ref yaw, :174-185
        const double yaw = io.ref[lane * SD + 3];
        sm[L.cy + lane] = cos(yaw);
        sm[L.sy + lane] = sin(-yaw);
    }
    for (int e = lane; e < (N + 1) * SD; e += 64) sm[L.X + e] = w[14 * (e / SD) + e % SD];
    for (int e = lane; e < N * UD; e += 64) sm[L.U + e] = w[14 * (e / UD) + 10 + e % UD];
    __syncthreads();
    const double *pr = sm + L.prm;
    const double f = evaluate<true, true>(sm, L, io, N, K, sm + L.X, sm + L.U, 0.0, 0.0, 0.0, nullptr, nullptr);
    __syncthreads();
    if (f_out && lane == 0) f_out[s] = f;
    if (grad_out) {  // zero on X_0 (no cost is attached to it), q on X_1..X_N, r on U_k
        double *go = grad_out + (size_t)s * nx;
        for (int e = lane; e < (N + 1) * SD; e += 64) go[14 * (e / SD) + e % SD] = e < SD ? 0.0 : sm[L.q + e];
        for (int e = lane; e < N * UD; e += 64) go[14 * (e / UD) + 10 + e % UD] = sm[L.r + e];
    }
    if (g_out) {
        double *go = g_out + (size_t)s * (10 + 10 * N);
        const double *A = pr + PRM_A, *B = pr + PRM_B, *c = pr + PRM_C;
        if (lane < SD) go[lane] = sm[L.X + lane] - sm[L.xinit + lane];  // :160
        for (int e = lane; e < N * SD; e += 64) {                      // F(X_k, U_k) - X_{k+1}, :219
            const int k = e / SD, i = e % SD;
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < SD; ++j) a += A[i * SD + j] * sm[L.X + k * SD + j];
#pragma unroll
            for (int j = 0; j < UD; ++j) a += B[i * UD + j] * sm[L.U + k * UD + j];
            go[10 + e] = (a + c[i]) - sm[L.X + (k + 1) * SD + i];
        }
    }
    if (jac_out) {  // constant; column-major, rows ascending (see amk_mpc_jac_sparsity)
        double *jo = jac_out + (size_t)s * (10 + 39 * N);
        const double *A = pr + PRM_A, *B = pr + PRM_B;
        for (int e = lane; e < (N + 1) * SD + N * UD; e += 64) {  // one decision variable = one column per lane-step
            const bool is_u = e >= (N + 1) * SD;
            const int k = is_u ? (e - (N + 1) * SD) / UD : e / SD, i = is_u ? (e - (N + 1) * SD) % UD : e % SD;
            // entries before the column: every earlier stage holds 29 (X) + 10 (U)
            int r[3];
            if (is_u) {
                int pos = 39 * k + 29;  // after X_k's column block
                for (int a = 0; a < i; ++a) pos += rows_of_B(a, r);
                const int nr = rows_of_B(i, r);
                for (int t = 0; t < nr; ++t) jo[pos + t] = B[r[t] * UD + i];
            } else {
                int pos = 39 * k;
                for (int j = 0; j < i; ++j) pos += 1 + (k < N ? rows_of_A(j, r) : 0);
                jo[pos] = k == 0 ? 1.0 : -1.0;  // d(X_0 - x_init)/dX_0 or d(-X_k)/dX_k of the previous defect
                if (k < N) {
                    const int nr = rows_of_A(i, r);
                    for (int t = 0; t < nr; ++t) jo[pos + 1 + t] = A[r[t] * SD + i];
                }
            }
        }
    }
    if (hess_out) {  // lam_f * upper triangle, column-major (see amk_mpc_hess_sparsity)
        double *ho = hess_out + (size_t)s * (25 * (N - 1) + 10 + 4 * N);
        const double sig = lam_f ? lam_f[s] : 1.0;
        for (int e = lane; e < 25 * (N - 1); e += 64) {  // X_1 .. X_{N-1}
            const int k = e / 25, t = e % 25;             // Hessian block of X_{k+1}, cost stage k
            const int i = kHessRow[t], j = kHessCol[t];
            double h = 0.0;
            const int bi = (i == 0 || i == 1) ? 0 : ((i == 4 || i == 5) ? 1 : -1);
            const int bj = (j == 0 || j == 1) ? 0 : ((j == 4 || j == 5) ? 1 : -1);
            if (bi >= 0 && bi == bj) {  // rotated 2x2 blocks of 2 R'Q_pen R, :174-185,206-208
                const double cy = sm[L.cy + k], sy = sm[L.sy + k];
                const double w0q = 2.0 * pr[PRM_W + 10 + 4 * bi], w1q = 2.0 * pr[PRM_W + 11 + 4 * bi];
                const int which = ((i == 1 || i == 5) ? 1 : 0) + ((j == 1 || j == 5) ? 1 : 0);
                h = which == 0 ? cy * cy * w0q + sy * sy * w1q : which == 1 ? -cy * sy * w0q + sy * cy * w1q : sy * sy * w0q + cy * cy * w1q;
            } else if (i == j) {
                h = 2.0 * pr[PRM_W + 10 + i];
            }
            const int pi = pv_inv(i), pj = pv_inv(j);
            if (pi >= 0 && pj >= 0) h += sm[L.H6 + k * 21 + pj * (pj + 1) / 2 + pi];  // i <= j: lower-triangle index (pj, pi)
            ho[4 * (k + 1) + 25 * k + t] = sig * h;
        }
        for (int e = lane; e < SD; e += 64) ho[4 * N + 25 * (N - 1) + e] = sig * 2.0 * pr[PRM_W + e];  // X_N: 2 Q_goal
        for (int e = lane; e < N * UD; e += 64) {                                                      // U_k: 2 Q_u
            const int k = e / UD, a = e % UD;
            ho[(k == 0 ? 0 : 4 * k + 25 * k) + a] = sig * 2.0 * pr[PRM_W + 20 + a];
        }
    }
}

// ---- nlp_grad (mpc_obstacle_casadi.py:294-303 emits it with the other dependencies; CasADi's nlpsol evaluates it after a solve
// for lam_p): gamma = lam_f f + lam_g' g;  outputs grad_x gamma = lam_f grad f + J' lam_g and grad_p gamma over the WHOLE
// parameter vector p = [x_init | ref | obstacles | target | gain(4) | tau(4) | weights(25) | radius].  dtau: d(A, B, c)/d tau_a
// for a = 0, 1, 2 ([3][150] doubles: A 100, B 40, c 10), from the host's dual-number RK4 probe (dynamics_dtau).
template <int NT>
__global__ __launch_bounds__(64) void mpc_eval_gamma_kernel(int Nrt, int K, int nref, int nx, const double *__restrict__ prm,
                                                            const double *__restrict__ dtau, const double *__restrict__ w_all,
                                                            const double *__restrict__ ref_states,
                                                            const double *__restrict__ lam_f_all,
                                                            const double *__restrict__ lam_g_all, double *__restrict__ gx_out,
                                                            double *__restrict__ gp_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    double *sm = reinterpret_cast<double *>(sm_raw);
    const int s = blockIdx.x, lane = threadIdx.x;
    const int N = NT > 0 ? NT : Nrt;
    const LdsMap L(N, PRM_LEN);
    const int ng = 10 + 10 * N, np = nref + 34;
    const double *P = ref_states + (size_t)s * nref;
    const double *w = w_all + (size_t)s * nx;
    const double *lg = lam_g_all ? lam_g_all + (size_t)s * ng : nullptr;
    const double sig = lam_f_all ? lam_f_all[s] : 1.0;
    SceneIO io;
    io.ref = P + SD;
    io.obs = P + SD + SD * N;
    const double *target = P + SD + SD * N + 3 * K * N;
    for (int e = lane; e < PRM_LEN; e += 64) sm[L.prm + e] = prm[e];
    if (lane < SD) {
        sm[L.xinit + lane] = P[lane];
        sm[L.target + lane] = target[lane];
    }
    if (lane < N - 1) {
        const double yaw = io.ref[lane * SD + 3];
        sm[L.cy + lane] = cos(yaw);
        sm[L.sy + lane] = sin(-yaw);
    }
    for (int e = lane; e < (N + 1) * SD; e += 64) sm[L.X + e] = w[14 * (e / SD) + e % SD];
    for (int e = lane; e < N * UD; e += 64) sm[L.U + e] = w[14 * (e / UD) + 10 + e % UD];
    __syncthreads();
    const double *pr = sm + L.prm;
    evaluate<true, true>(sm, L, io, N, K, sm + L.X, sm + L.U, 0.0, 0.0, 0.0, nullptr, nullptr);
    __syncthreads();
    auto lam = [&](int row) { return lg ? lg[row] : 0.0; };
    if (gx_out) {
        double *go = gx_out + (size_t)s * nx;
        const double *A = pr + PRM_A, *B = pr + PRM_B;
        int r[3];
        for (int e = lane; e < (N + 1) * SD; e += 64) {
            const int k = e / SD, i = e % SD;
            double a = sig * (k == 0 ? 0.0 : sm[L.q + e]);
            a += k == 0 ? lam(i) : -lam(10 + 10 * (k - 1) + i);
            if (k < N) {
                const int nr = rows_of_A(i, r);
                for (int t = 0; t < nr; ++t) a += A[r[t] * SD + i] * lam(10 + 10 * k + r[t]);
            }
            go[14 * k + i] = a;
        }
        for (int e = lane; e < N * UD; e += 64) {
            const int k = e / UD, c = e % UD;
            double a = sig * sm[L.r + e];
            const int nr = rows_of_B(c, r);
            for (int t = 0; t < nr; ++t) a += B[r[t] * UD + c] * lam(10 + 10 * k + r[t]);
            go[14 * k + 10 + c] = a;
        }
    }
    if (!gp_out) return;
    double *gp = gp_out + (size_t)s * np;
    const double lamw = pr[PRM_W + 24], radius = pr[PRM_RADIUS];
    if (lane < SD) gp[lane] = -lam(lane);                                     // x_init: g[0:10] = X_0 - x_init
    // ---- reference states and the Q_pen part of the weights: lane = path stage
    double wq[SD];
#pragma unroll
    for (int i = 0; i < SD; ++i) wq[i] = 0.0;
    for (int k = lane; k < N; k += 64) {
        double *o = gp + SD + k * SD;
        if (k >= N - 1) {  // the last reference state is not used by the objective (:168-171)
            for (int i = 0; i < SD; ++i) o[i] = 0.0;
            continue;
        }
        const double *xk = sm + L.X + (k + 1) * SD, *rf = io.ref + k * SD;
        const double cy = sm[L.cy + k], sy = sm[L.sy + k];
        double d[SD], y[SD];
        for (int i = 0; i < SD; ++i) { d[i] = xk[i] - rf[i]; y[i] = d[i]; }
        y[0] = cy * d[0] - sy * d[1]; y[1] = sy * d[0] + cy * d[1];
        y[4] = cy * d[4] - sy * d[5]; y[5] = sy * d[4] + cy * d[5];
        double wy[SD];
        for (int i = 0; i < SD; ++i) { wy[i] = 2.0 * pr[PRM_W + 10 + i] * y[i]; wq[i] += y[i] * y[i]; }
        double qq[SD];
        for (int i = 0; i < SD; ++i) qq[i] = wy[i];
        qq[0] = cy * wy[0] + sy * wy[1]; qq[1] = -sy * wy[0] + cy * wy[1];
        qq[4] = cy * wy[4] + sy * wy[5]; qq[5] = -sy * wy[4] + cy * wy[5];
        for (int i = 0; i < SD; ++i) o[i] = -sig * qq[i];
        // yaw_ref also turns the rotation: dy0 = y1, dy1 = -y0 (and 4, 5 alike) per unit of yaw_ref
        o[3] = sig * (-wy[3] + (wy[0] * y[1] - wy[1] * y[0]) + (wy[4] * y[5] - wy[5] * y[4]));
    }
    // ---- obstacle points, d/d lambda, d/d radius: lane = collision term
    double dl = 0.0, dr = 0.0;
    for (int e = lane; e < N * K; e += 64) {
        const int k = e / K;
        double *o = gp + SD + SD * N + 3 * e;
        if (k >= N - 1) { o[0] = o[1] = o[2] = 0.0; continue; }
        const double *xk = sm + L.X + (k + 1) * SD;
        const double *op = io.obs + 3 * (size_t)e;
        const double d0 = op[0] - xk[0], d1 = op[1] - xk[1], d2 = op[2] - xk[2];
        const double rho = sqrt(d0 * d0 + d1 * d1 + d2 * d2), ir = 1.0 / rho;
        const double x = -32.0 * (rho - radius), ex = exp(x), g = log(1.0 + ex), sg = ex / (1.0 + ex), gpr = -32.0 * sg;
        const double n[3] = {d0 * ir, d1 * ir, d2 * ir};
        const double sv = xk[4] * n[0] + xk[5] * n[1] + xk[6] * n[2];
        const double sgn = sv > 0.0 ? 1.0 : (sv < 0.0 ? -1.0 : 0.0);
        const double t[3] = {xk[4] - sv * n[0], xk[5] - sv * n[1], xk[6] - sv * n[2]};
        for (int i = 0; i < 3; ++i) o[i] = sig * lamw * sgn * (gpr * n[i] * sv + g * t[i] * ir);  // = -d/dp of the term
        dl += g * fabs(sv);
        dr += lamw * 32.0 * sg * fabs(sv);
    }
    dl = wave_sum(dl); dr = wave_sum(dr);
    double wsum[SD];
#pragma unroll
    for (int i = 0; i < SD; ++i) wsum[i] = wave_sum(wq[i]);
    // ---- control weights, d/d tau
    double wu[UD] = {0.0, 0.0, 0.0, 0.0}, dt3[3] = {0.0, 0.0, 0.0};
    for (int k = lane; k < N; k += 64) {
        const double *uk = sm + L.U + k * UD, *xk = sm + L.X + k * SD;
        const double uref[4] = {0.0, 0.0, 9.81, 0.0};
        for (int i = 0; i < UD; ++i) wu[i] += (uk[i] - uref[i]) * (uk[i] - uref[i]);
        for (int a = 0; a < 3; ++a) {
            const double *dA = dtau + a * 150, *dB = dA + 100, *dc = dA + 140;
            double acc = 0.0;
            for (int i = 0; i < SD; ++i) {
                double fi = dc[i];
                for (int j = 0; j < SD; ++j) fi += dA[i * SD + j] * xk[j];
                for (int j = 0; j < UD; ++j) fi += dB[i * UD + j] * uk[j];
                acc += lam(10 + 10 * k + i) * fi;
            }
            dt3[a] += acc;
        }
    }
    for (int i = 0; i < UD; ++i) wu[i] = wave_sum(wu[i]);
    for (int a = 0; a < 3; ++a) dt3[a] = wave_sum(dt3[a]);
    double *tg = gp + SD + SD * N + 3 * K * N;
    if (lane < SD) {
        const double dgl = sm[L.X + N * SD + lane] - sm[L.target + lane];
        tg[lane] = -sig * 2.0 * pr[PRM_W + lane] * dgl;                        // target
        tg[SD + 8 + lane] = sig * dgl * dgl;                                   // Q_goal
    }
    if (lane == 0) {
        double *tail = tg + SD;   // gain(4) tau(4) weights(25) radius
        for (int i = 0; i < 4; ++i) tail[i] = 0.0;                            // gain: unused by the model (:114-121)
        for (int a = 0; a < 3; ++a) tail[4 + a] = dt3[a];
        tail[7] = 0.0;                                                        // tau_yaw: unused
        for (int i = 0; i < SD; ++i) tail[8 + 10 + i] = sig * wsum[i];        // Q_pen
        for (int i = 0; i < UD; ++i) tail[8 + 20 + i] = sig * wu[i];          // Q_u
        tail[8 + 24] = sig * dl;                                              // lambda
        tail[33] = sig * dr;                                                  // radius
    }
}

// d(A, B, c)/d tau_a, a = 0..2, by running the RK4 x 4 probe of mpc_solve.hip's refresh_dynamics on dual numbers
struct Dual { double v, d; };
inline Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
inline Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
inline Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
inline Dual operator*(double a, Dual b) { return {a * b.v, a * b.d}; }
void ode_dual(const Dual *x, const Dual *u, const Dual *tau, const double *drag, Dual *xd) {  // mpc_obstacle_casadi.py:95-122
    xd[0] = x[4]; xd[1] = x[5]; xd[2] = x[6];
    xd[3] = u[3];
    xd[4] = x[7] - drag[0] * x[4]; xd[5] = x[8] - drag[1] * x[5]; xd[6] = x[9] - drag[2] * x[6];   // (amk_mpc_set_drag_coefficient)
    xd[7] = (u[0] - x[7]) * tau[0];
    xd[8] = (u[1] - x[8]) * tau[1];
    xd[9] = (u[2] - Dual{9.81, 0.0} - x[9]) * tau[2];
}
void rk4_dual(const Dual *x, const Dual *u, const Dual *tau, const double *drag, double dt, Dual *xn) {  // :338-357
    const double DT = dt / 4;
    Dual X[SD], k1[SD], k2[SD], k3[SD], k4[SD], t[SD];
    for (int i = 0; i < SD; ++i) X[i] = x[i];
    for (int m = 0; m < 4; ++m) {
        ode_dual(X, u, tau, drag, k1);
        for (int i = 0; i < SD; ++i) { k1[i] = DT * k1[i]; t[i] = X[i] + 0.5 * k1[i]; }
        ode_dual(t, u, tau, drag, k2);
        for (int i = 0; i < SD; ++i) { k2[i] = DT * k2[i]; t[i] = X[i] + 0.5 * k2[i]; }
        ode_dual(t, u, tau, drag, k3);
        for (int i = 0; i < SD; ++i) { k3[i] = DT * k3[i]; t[i] = X[i] + k3[i]; }
        ode_dual(t, u, tau, drag, k4);
        for (int i = 0; i < SD; ++i) { k4[i] = DT * k4[i]; X[i] = X[i] + (1.0 / 6) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]); }
    }
    for (int i = 0; i < SD; ++i) xn[i] = X[i];
}
void dynamics_dtau(const amk_mpc *m, double *out /* [3][150] */) {
    for (int a = 0; a < 3; ++a) {
        Dual tau[4];
        for (int i = 0; i < 4; ++i) tau[i] = {m->h_prm[PRM_TAU + i], i == a ? 1.0 : 0.0};
        Dual z10[SD], z4[UD], e[SD], c[SD], f[SD];
        for (int i = 0; i < SD; ++i) z10[i] = {0.0, 0.0};
        for (int i = 0; i < UD; ++i) z4[i] = {0.0, 0.0};
        double *dA = out + a * 150, *dB = dA + 100, *dc = dA + 140;
        rk4_dual(z10, z4, tau, m->drag, m->dt, c);
        for (int i = 0; i < SD; ++i) dc[i] = c[i].d;
        for (int j = 0; j < SD; ++j) {
            for (int i = 0; i < SD; ++i) e[i] = {i == j ? 1.0 : 0.0, 0.0};
            rk4_dual(e, z4, tau, m->drag, m->dt, f);
            for (int i = 0; i < SD; ++i) dA[i * SD + j] = f[i].d - c[i].d;
        }
        for (int j = 0; j < UD; ++j) {
            Dual e4[UD];
            for (int i = 0; i < UD; ++i) e4[i] = {i == j ? 1.0 : 0.0, 0.0};
            rk4_dual(z10, e4, tau, m->drag, m->dt, f);
            for (int i = 0; i < SD; ++i) dB[i * UD + j] = f[i].d - c[i].d;
        }
    }
}

int eval_nnz_jac(const amk_mpc *m) { return 10 + 39 * m->N; }
int eval_nnz_hess(const amk_mpc *m) { return 25 * (m->N - 1) + 10 + 4 * m->N; }

void rows_of_A_h(int j, std::vector<int> &r) {
    r.clear();
    if (j < 4) r = {j};
    else if (j < 7) r = {j - 4, j};
    else r = {j - 7, j - 3, j};
}
void rows_of_B_h(int a, std::vector<int> &r) {
    r.clear();
    if (a == 3) r = {3};
    else r = {a, 4 + a, 7 + a};
}

}  // namespace

extern "C" {

int amk_mpc_ng(const amk_mpc *m) { return m ? 10 + 10 * m->N : -1; }
int amk_mpc_jac_nnz(const amk_mpc *m) { return m ? eval_nnz_jac(m) : -1; }
int amk_mpc_hess_nnz(const amk_mpc *m) { return m ? eval_nnz_hess(m) : -1; }

int amk_mpc_jac_sparsity(const amk_mpc *m, int *colind, int *row) {
    if (!m || !colind || !row) return AMK_ERR_INVALID_ARG;
    const int N = m->N;
    int pos = 0, col = 0;
    std::vector<int> r;
    for (int k = 0; k <= N; ++k) {
        for (int i = 0; i < SD; ++i) {  // column of X_k[i]
            colind[col++] = pos;
            row[pos++] = k == 0 ? i : 10 + 10 * (k - 1) + i;
            if (k < N) {
                rows_of_A_h(i, r);
                for (int t : r) row[pos++] = 10 + 10 * k + t;
            }
        }
        if (k < N)
            for (int a = 0; a < UD; ++a) {  // column of U_k[a]
                colind[col++] = pos;
                rows_of_B_h(a, r);
                for (int t : r) row[pos++] = 10 + 10 * k + t;
            }
    }
    colind[col] = pos;
    return pos == eval_nnz_jac(m) && col == m->nx ? AMK_OK : AMK_ERR_UNSUPPORTED;
}

int amk_mpc_hess_sparsity(const amk_mpc *m, int *colind, int *row) {
    if (!m || !colind || !row) return AMK_ERR_INVALID_ARG;
    const int N = m->N;
    int pos = 0, col = 0;
    for (int k = 0; k <= N; ++k) {
        for (int j = 0; j < SD; ++j) {  // column of X_k[j]
            colind[col++] = pos;
            if (k == 0) continue;       // no cost on X_0
            if (k == N) { row[pos++] = 14 * k + j; continue; }
            for (int t = 0; t < 25; ++t)
                if (kHessColH[t] == j) row[pos++] = 14 * k + kHessRowH[t];
        }
        if (k < N)
            for (int a = 0; a < UD; ++a) {
                colind[col++] = pos;
                row[pos++] = 14 * k + 10 + a;
            }
    }
    colind[col] = pos;
    return pos == eval_nnz_hess(m) && col == m->nx ? AMK_OK : AMK_ERR_UNSUPPORTED;
}

int amk_mpc_eval(amk_mpc *m, const double *d_w, const double *d_ref_states, const double *d_lam_f, double *d_f,
                 double *d_grad_f, double *d_g, double *d_jac_g, double *d_hess_l, void *stream) {
    if (!m || !d_w || !d_ref_states) return AMK_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    // the kernel lays its scratchpad out with LdsMap(N, PRM_LEN) -- A and B in LDS, 140 doubles more than the solver's map
    const size_t lds = sizeof(double) * (size_t)LdsMap(m->N, PRM_LEN).total;
#define AMK_LAUNCH_EVAL(NT)                                                                                            \
    hipLaunchKernelGGL(mpc_eval_kernel<NT>, dim3(m->S), dim3(64), lds, st, m->N, m->K, m->nref, m->nx, m->prm.p,          \
                       d_w, d_ref_states, d_lam_f, d_f, d_grad_f, d_g, d_jac_g, d_hess_l)
    switch (m->N) {
        case 10: AMK_LAUNCH_EVAL(10); break;
        case 20: AMK_LAUNCH_EVAL(20); break;
        case 30: AMK_LAUNCH_EVAL(30); break;
        default: AMK_LAUNCH_EVAL(0); break;
    }
#undef AMK_LAUNCH_EVAL
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_mpc_np(const amk_mpc *m) { return m ? m->nref + 34 : -1; }

int amk_mpc_eval_gamma(amk_mpc *m, const double *d_w, const double *d_ref_states, const double *d_lam_f, const double *d_lam_g,
                       double *d_grad_x, double *d_grad_p, void *stream) {
    if (!m || !d_w || !d_ref_states) return AMK_ERR_INVALID_ARG;
    if (!m->ev_dtau.p) AMK_HIP(m->ev_dtau.alloc(3 * 150));
    if (!m->ev_dtau_valid || std::memcmp(m->ev_dtau_tau, m->h_prm + PRM_TAU, sizeof m->ev_dtau_tau) != 0) {
        double h[3 * 150];
        dynamics_dtau(m, h);
        AMK_HIP(hipMemcpy(m->ev_dtau.p, h, sizeof h, hipMemcpyHostToDevice));
        std::memcpy(m->ev_dtau_tau, m->h_prm + PRM_TAU, sizeof m->ev_dtau_tau);
        m->ev_dtau_valid = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = sizeof(double) * (size_t)LdsMap(m->N, PRM_LEN).total;
#define AMK_LAUNCH_GAMMA(NT)                                                                                           \
    hipLaunchKernelGGL(mpc_eval_gamma_kernel<NT>, dim3(m->S), dim3(64), lds, st, m->N, m->K, m->nref, m->nx, m->prm.p,    \
                       m->ev_dtau.p, d_w, d_ref_states, d_lam_f, d_lam_g, d_grad_x, d_grad_p)
    switch (m->N) {
        case 10: AMK_LAUNCH_GAMMA(10); break;
        case 20: AMK_LAUNCH_GAMMA(20); break;
        case 30: AMK_LAUNCH_GAMMA(30); break;
        default: AMK_LAUNCH_GAMMA(0); break;
    }
#undef AMK_LAUNCH_GAMMA
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_mpc_eval_gamma_host(amk_mpc *m, const double *h_w, const double *h_ref_states, const double *h_lam_f,
                            const double *h_lam_g, double *h_grad_x, double *h_grad_p) {
    if (!m || !h_w || !h_ref_states) return AMK_ERR_INVALID_ARG;
    const size_t S = m->S, nx = m->nx, ng = 10 + 10 * m->N, np = m->nref + 34;
    if (!m->ev_w.p) {
        AMK_HIP(m->ev_w.alloc(S * nx));
        AMK_HIP(m->ev_ref.alloc(S * m->nref));
        AMK_HIP(m->ev_out.alloc(S * (2 + nx + ng + eval_nnz_jac(m) + eval_nnz_hess(m))));
    }
    if (!m->ev_gam.p) AMK_HIP(m->ev_gam.alloc(S * (1 + ng + nx + np)));
    double *lf = m->ev_gam.p, *lgm = lf + S, *gx = lgm + S * ng, *gp = gx + S * nx;
    AMK_HIP(hipMemcpy(m->ev_w.p, h_w, sizeof(double) * S * nx, hipMemcpyHostToDevice));
    AMK_HIP(hipMemcpy(m->ev_ref.p, h_ref_states, sizeof(double) * S * m->nref, hipMemcpyHostToDevice));
    if (h_lam_f) AMK_HIP(hipMemcpy(lf, h_lam_f, sizeof(double) * S, hipMemcpyHostToDevice));
    if (h_lam_g) AMK_HIP(hipMemcpy(lgm, h_lam_g, sizeof(double) * S * ng, hipMemcpyHostToDevice));
    int st = amk_mpc_eval_gamma(m, m->ev_w.p, m->ev_ref.p, h_lam_f ? lf : nullptr, h_lam_g ? lgm : nullptr,
                                h_grad_x ? gx : nullptr, h_grad_p ? gp : nullptr, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipStreamSynchronize(nullptr));
    if (h_grad_x) AMK_HIP(hipMemcpy(h_grad_x, gx, sizeof(double) * S * nx, hipMemcpyDeviceToHost));
    if (h_grad_p) AMK_HIP(hipMemcpy(h_grad_p, gp, sizeof(double) * S * np, hipMemcpyDeviceToHost));
    return AMK_OK;
}

int amk_mpc_eval_host(amk_mpc *m, const double *h_w, const double *h_ref_states, const double *h_lam_f, double *h_f,
                      double *h_grad_f, double *h_g, double *h_jac_g, double *h_hess_l) {
    if (!m || !h_w || !h_ref_states) return AMK_ERR_INVALID_ARG;
    const size_t S = m->S, nx = m->nx, ng = 10 + 10 * m->N, nj = eval_nnz_jac(m), nh = eval_nnz_hess(m);
    if (!m->ev_w.p) {
        AMK_HIP(m->ev_w.alloc(S * nx));
        AMK_HIP(m->ev_ref.alloc(S * m->nref));
        AMK_HIP(m->ev_out.alloc(S * (2 + nx + ng + nj + nh)));
    }
    double *lam = m->ev_out.p, *f = lam + S, *gr = f + S, *g = gr + S * nx, *jac = g + S * ng, *hs = jac + S * nj;
    AMK_HIP(hipMemcpy(m->ev_w.p, h_w, sizeof(double) * S * nx, hipMemcpyHostToDevice));
    AMK_HIP(hipMemcpy(m->ev_ref.p, h_ref_states, sizeof(double) * S * m->nref, hipMemcpyHostToDevice));
    if (h_lam_f) AMK_HIP(hipMemcpy(lam, h_lam_f, sizeof(double) * S, hipMemcpyHostToDevice));
    int st = amk_mpc_eval(m, m->ev_w.p, m->ev_ref.p, h_lam_f ? lam : nullptr, h_f ? f : nullptr, h_grad_f ? gr : nullptr,
                          h_g ? g : nullptr, h_jac_g ? jac : nullptr, h_hess_l ? hs : nullptr, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipStreamSynchronize(nullptr));
    if (h_f) AMK_HIP(hipMemcpy(h_f, f, sizeof(double) * S, hipMemcpyDeviceToHost));
    if (h_grad_f) AMK_HIP(hipMemcpy(h_grad_f, gr, sizeof(double) * S * nx, hipMemcpyDeviceToHost));
    if (h_g) AMK_HIP(hipMemcpy(h_g, g, sizeof(double) * S * ng, hipMemcpyDeviceToHost));
    if (h_jac_g) AMK_HIP(hipMemcpy(h_jac_g, jac, sizeof(double) * S * nj, hipMemcpyDeviceToHost));
    if (h_hess_l) AMK_HIP(hipMemcpy(h_hess_l, hs, sizeof(double) * S * nh, hipMemcpyDeviceToHost));
    return AMK_OK;
}

}  // extern "C"
