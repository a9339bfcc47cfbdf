// MPC solve of the Avoid-MPC hot path for gfx950: batch of ObstacleAvoidanceMPC objects.
//
// Replaces AM/include/HighLvlMpc.h:4-33 / AM/src/HighLvlMpc.cpp:5-137 (class
// ObstacleAvoidanceMPC: constructor defaults, setters, Solve, warm start) and the generated plugin it
// loads through casadi::nlpsol (AM/tools/mpc_obstacle_casadi.py).  Device algorithm: mpc_device.h.
#include "mpc_handle.h"

#include <cmath>
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

using namespace amk;

namespace {

// ---- host-side model: RK4 x 4 of the quadrotor ODE (mpc_obstacle_casadi.py:106-122, 338-357) probed for
// its affine form F(x,u) = A x + B u + c (exactly affine with the drag term off, the repo default
// mpc_parameters.yaml:4).
// Drag (mpc_obstacle_casadi.py:95-105, yaml use_drag_coefficient): `rotmat * diag(k, k, k) * rotmat.T * v` read as matrix products is
// k v whatever the attitude (R (k I) R' = k I) -- linear in the state, so F stays affine with the same sparsity (amk_mpc_set_drag_coefficient).
void ode_host(const double *x, const double *u, const double *tau, const double *drag, double *xd) {
    xd[0] = x[4]; xd[1] = x[5]; xd[2] = x[6];
    xd[3] = u[3];
    xd[4] = x[7] - drag[0] * x[4]; xd[5] = x[8] - drag[1] * x[5]; xd[6] = x[9] - drag[2] * x[6];
    xd[7] = (u[0] - x[7]) * tau[0];
    xd[8] = (u[1] - x[8]) * tau[1];
    xd[9] = (u[2] - kGz - x[9]) * tau[2];
}

void rk4_host(const double *x, const double *u, const double *tau, const double *drag, double dt, double *xn) {
    const int M = 4;
    const double DT = dt / M;
    double X[SD], k1[SD], k2[SD], k3[SD], k4[SD], t[SD];
    std::memcpy(X, x, sizeof X);
    for (int m = 0; m < M; ++m) {
        ode_host(X, u, tau, drag, k1);
        for (int i = 0; i < SD; ++i) { k1[i] *= DT; t[i] = X[i] + 0.5 * k1[i]; }
        ode_host(t, u, tau, drag, k2);
        for (int i = 0; i < SD; ++i) { k2[i] *= DT; t[i] = X[i] + 0.5 * k2[i]; }
        ode_host(t, u, tau, drag, k3);
        for (int i = 0; i < SD; ++i) { k3[i] *= DT; t[i] = X[i] + k3[i]; }
        ode_host(t, u, tau, drag, k4);
        for (int i = 0; i < SD; ++i) { k4[i] *= DT; X[i] = X[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6; }
    }
    std::memcpy(xn, X, sizeof X);
}

void refresh_dynamics(amk_mpc *m) {
    double *A = m->h_prm + PRM_A, *B = m->h_prm + PRM_B, *c = m->h_prm + PRM_C;
    const double *tau = m->h_prm + PRM_TAU;
    double z10[SD] = {0}, z4[UD] = {0}, e10[SD], e4[UD], f[SD];
    rk4_host(z10, z4, tau, m->drag, m->dt, c);
    for (int j = 0; j < SD; ++j) {
        std::memset(e10, 0, sizeof e10);
        e10[j] = 1.0;
        rk4_host(e10, z4, tau, m->drag, m->dt, f);
        for (int i = 0; i < SD; ++i) A[i * SD + j] = f[i] - c[i];
    }
    for (int j = 0; j < UD; ++j) {
        std::memset(e4, 0, sizeof e4);
        e4[j] = 1.0;
        rk4_host(z10, e4, tau, m->drag, m->dt, f);
        for (int i = 0; i < SD; ++i) B[i * UD + j] = f[i] - c[i];
    }
}

// Enumerates, for every entry of A'PA, B'PB(+R), B'PA, A'p, A'lam, B'p(+r_bar), B'lam(+r), the <= 9 source cells
// and coefficients (mpc_device.h "Riccati plan").  Generic in the non-zeros of A and B except that the yaw chain
// must be decoupled from the others (checked; it is for every tau/gain of the model), so P[yaw][other] == 0 for every stage and those terms drop out.
int build_plan(amk_mpc *m, std::vector<double> &coef, std::vector<int> &meta) {
    const double *A = m->h_prm + PRM_A, *B = m->h_prm + PRM_B;
    const LdsMap L(m->N);
    bool yaw_free = true;
    for (int i = 0; i < SD; ++i)
        if (i != 3 && (A[3 * SD + i] != 0.0 || A[i * SD + 3] != 0.0)) yaw_free = false;
    for (int a = 0; a < 3; ++a)
        if (B[3 * UD + a] != 0.0) yaw_free = false;
    for (int i = 0; i < SD; ++i)
        if (i != 3 && B[i * UD + 3] != 0.0) yaw_free = false;
    if (!yaw_free) return AMK_ERR_UNSUPPORTED;  // the control block is inverted as blkdiag(3x3, 1x1) (riccati_backward)
    // the sweeps assume the chain structure p_a <- v_a <- a_a <- u_a, yaw <- u_3 (rows_of_A / rows_of_B / cols_of_A_row)
    auto axis = [](int i) { return i < 3 ? i : (i == 3 ? 3 : (i < 7 ? i - 4 : i - 7)); };
    auto level = [](int i) { return i < 3 ? 0 : (i == 3 ? 0 : (i < 7 ? 1 : 2)); };
    for (int i = 0; i < SD; ++i) {
        for (int j = 0; j < SD; ++j)
            if (A[i * SD + j] != 0.0 && (axis(i) != axis(j) || level(j) < level(i))) return AMK_ERR_UNSUPPORTED;
        for (int a = 0; a < UD; ++a)
            if (B[i * UD + a] != 0.0 && axis(i) != a) return AMK_ERR_UNSUPPORTED;
    }
    auto zeroP = [&](int l, int mm) { return (l == 3) != (mm == 3); };
    const int ZERO = L.red + 10, DUMMY = L.red + 8;  // a cell that always holds 0.0 / a write-only cell
    struct Item { std::vector<std::pair<int, double>> t; int out, ks, aux, aux_ks; double dflag; };
    std::vector<Item> items;
    auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };
    auto mat_terms = [&](const double *Lf, int ls, int ci, const double *Rf, int rs, int cj) {
        std::vector<std::pair<int, double>> t;
        for (int l = 0; l < SD; ++l) {
            if (Lf[l * ls + ci] == 0.0) continue;
            for (int mm = 0; mm < SD; ++mm) {
                if (Rf[mm * rs + cj] == 0.0 || zeroP(l, mm)) continue;
                t.push_back({L.P + (l >= mm ? l * 10 + mm : mm * 10 + l), Lf[l * ls + ci] * Rf[mm * rs + cj]});  // P is symmetric: only its lower triangle is kept current
            }
        }
        return t;
    };
    auto vec_terms = [&](const double *Lf, int ls, int ci, int base) {
        std::vector<std::pair<int, double>> t;
        for (int l = 0; l < SD; ++l)
            if (Lf[l * ls + ci] != 0.0) t.push_back({base + l, Lf[l * ls + ci]});
        return t;
    };
    for (int i = 0; i < SD; ++i)
        for (int j = 0; j <= i; ++j) {
            auto t = mat_terms(A, SD, i, A, SD, j);
            if (!t.empty()) items.push_back({t, L.M + tri(i, j), 0, ZERO, 0, 0.0});
        }
    for (int a = 0; a < UD; ++a)
        for (int b = 0; b <= a; ++b) {
            auto t = mat_terms(B, UD, a, B, UD, b);
            if (a == b) items.push_back({t, L.Hm + tri(a, b), 0, L.Rb + a, UD, 1.0});  // + R_bar_k + delta
            else if (!t.empty()) items.push_back({t, L.Hm + tri(a, b), 0, ZERO, 0, 0.0});
        }
    for (int a = 0; a < UD; ++a)
        for (int j = 0; j < SD; ++j) {
            auto t = mat_terms(B, UD, a, A, SD, j);
            if (!t.empty()) items.push_back({t, L.G + a * 10 + j, 0, ZERO, 0, 0.0});
        }
    // q_k + A'p: the stage gradient rides in as the per-stage addend (the reduced gradient gU = r + B'lam is the adjoint
    // scans' job, mpc_device_impl.h: adjoint_sweep)
    for (int i = 0; i < SD; ++i) items.push_back({vec_terms(A, SD, i, L.p), L.Atp + i, 0, L.q + i, SD, 0.0});
    for (int a = 0; a < UD; ++a) items.push_back({vec_terms(B, UD, a, L.p), L.qu + a, 0, L.rb + a, UD, 0.0});
    if ((int)items.size() > PLAN_ITEMS) return AMK_ERR_UNSUPPORTED;
    // heavy items first: lane l executes items l (<= 9 terms) and 64 + l (<= PLAN_TERMS_LIGHT terms)
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.t.size() > b.t.size(); });
    for (size_t e = 64; e < items.size(); ++e)
        if ((int)items[e].t.size() > PLAN_TERMS_LIGHT) return AMK_ERR_UNSUPPORTED;
    coef.assign((size_t)PLAN_ITEMS * (PLAN_TERMS + 1) + 64 * 2, 0.0);
    meta.assign((size_t)PLAN_ITEMS * (PLAN_TERMS + 4) + 64 * LANE_META_INTS, 0);
    for (int e = 0; e < PLAN_ITEMS; ++e) {
        int *mt = meta.data() + (size_t)e * (PLAN_TERMS + 4);
        for (int t = 0; t < PLAN_TERMS; ++t) mt[t] = ZERO;
        mt[PLAN_TERMS + 0] = DUMMY;
        mt[PLAN_TERMS + 1] = 0;
        mt[PLAN_TERMS + 2] = ZERO;
        mt[PLAN_TERMS + 3] = 0;
        if (e >= (int)items.size()) continue;
        const Item &it = items[e];
        if ((int)it.t.size() > PLAN_TERMS) return AMK_ERR_UNSUPPORTED;
        for (size_t t = 0; t < it.t.size(); ++t) {
            mt[t] = it.t[t].first;
            coef[(size_t)e * (PLAN_TERMS + 1) + t] = it.t[t].second;
        }
        coef[(size_t)e * (PLAN_TERMS + 1) + PLAN_TERMS] = it.dflag;
        mt[PLAN_TERMS + 0] = it.out; mt[PLAN_TERMS + 1] = it.ks; mt[PLAN_TERMS + 2] = it.aux; mt[PLAN_TERMS + 3] = it.aux_ks;
    }
    // ---- round B/C roles (mpc_device.h LaneRole): 46 lanes own the entries of P that are not structurally
    // zero, 10 own p, the rest idle; the lanes on the diagonal of P / on p_0 also store a column of the gains
    int *lm = meta.data() + (size_t)PLAN_ITEMS * (PLAN_TERMS + 4);
    double *lc = coef.data() + (size_t)PLAN_ITEMS * (PLAN_TERMS + 1);
    std::vector<int> free_lanes;
    auto pv_inv_h = [](int s) { return s < 3 ? s : (s >= 4 && s <= 6 ? s - 1 : -1); };
    for (int lane = 0; lane < 64; ++lane) {
        int *r = lm + lane * LANE_META_INTS;
        // defaults: idle lane (zero columns, zero base, writes to the dummy cell)
        r[0] = ZERO; r[1] = 0; r[2] = ZERO; r[3] = 0;
        for (int t = 0; t < 3; ++t) { r[4 + t] = ZERO; r[7 + t] = 0; }
        r[10] = DUMMY; r[11] = DUMMY; r[12] = -1; r[13] = ZERO; r[14] = DUMMY;
        if (lane >= 55) { free_lanes.push_back(lane); continue; }
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= lane) ++i;
        const int j = lane - i * (i + 1) / 2;
        if (zeroP(i, j)) { free_lanes.push_back(lane); continue; }
        r[0] = L.G + i; r[1] = 10; r[2] = L.G + j; r[3] = 10;
        int nb = 0;
        r[4 + nb] = L.M + lane; r[7 + nb] = 0; ++nb;
        const int bi = (i == 0 || i == 1) ? 0 : ((i == 4 || i == 5) ? 1 : -1);
        const int bj = (j == 0 || j == 1) ? 0 : ((j == 4 || j == 5) ? 1 : -1);
        if (bi >= 0 && bi == bj) {  // rotated 2x2 blocks of Q_k, stage k-1: rotQ[(k-1)*6 + .]
            r[4 + nb] = L.rotQ - 6 + bi * 3 + ((i == 1 || i == 5) ? 1 : 0) + ((j == 1 || j == 5) ? 1 : 0); r[7 + nb] = 6; ++nb;
        } else if (i == j) {
            lc[lane * 2 + 0] = 2.0 * m->h_prm[PRM_W + 10 + i];
        }
        const int pi = pv_inv_h(i), pj = pv_inv_h(j);
        if (pi >= 0 && pj >= 0) {   // collision Hessian, stage k-1: H6[(k-1)*21 + .]
            r[4 + nb] = L.H6 - 21 + (pi >= pj ? pi * (pi + 1) / 2 + pj : pj * (pj + 1) / 2 + pi); r[7 + nb] = 21; ++nb;
        }
        lc[lane * 2 + 1] = (i == j) ? 1.0 : 0.0;  // + delta on the diagonal
        r[10] = L.P + i * 10 + j; r[11] = DUMMY;   // lower triangle (i >= j)
        if (i == j) r[12] = i;
    }
    if ((int)free_lanes.size() < SD) return AMK_ERR_UNSUPPORTED;
    for (int i = 0; i < SD; ++i) {  // p_k[i] = (q_k + A'p)[i] - G(:,i)' Hm^-1 qu
        const int lane = free_lanes[i];
        int *r = lm + lane * LANE_META_INTS;
        r[0] = L.G + i; r[1] = 10; r[2] = L.qu; r[3] = 1;
        r[4] = L.Atp + i; r[7] = 0;
        r[10] = L.p + i; r[11] = L.p + i;
        if (i == 0) r[12] = SD;  // feed-forward column
    }
    return AMK_OK;
}

int upload_params(amk_mpc *m) {
    static_assert(sizeof(PlanItemMeta) == sizeof(int) * (PLAN_TERMS + 4), "PlanItemMeta layout");
    static_assert(sizeof(LaneRole) == sizeof(int) * LANE_META_INTS, "LaneRole layout");
    AMK_HIP(hipMemcpy(m->prm.p, m->h_prm, sizeof(double) * PRM_LEN, hipMemcpyHostToDevice));
    std::vector<double> coef;
    std::vector<int> meta;
    int st = build_plan(m, coef, meta);
    if (st != AMK_OK) return st;
    AMK_HIP(hipMemcpy(m->plan_coef.p, coef.data(), sizeof(double) * coef.size(), hipMemcpyHostToDevice));
    AMK_HIP(hipMemcpy(m->plan_meta.p, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice));
    return AMK_OK;
}

}  // namespace

// Internal (tests/test_fast_math_gpu.py): fast_math.h on n doubles -- out = [exp(x), log(|x|), 1/x] planes
__global__ void fast_math_probe_kernel(const double *__restrict__ x, double *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = amk::fast_exp(x[i]);
    out[n + i] = amk::fast_log(fabs(x[i]));
    out[2 * n + i] = amk::rcp_f64(x[i]);
}
extern "C" int amk__fast_math_probe(const double *d_x, double *d_out, int n, void *stream) {
    hipLaunchKernelGGL(fast_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_x, d_out, n);
    return hipGetLastError() == hipSuccess ? AMK_OK : AMK_ERR_HIP;
}

// Internal debugging aid (not part of the C ABI): when set, scene 0 of mpc_solve_kernel writes 8 doubles per
// interior-point iteration {J, kkt error, mu, delta, alpha, alpha_pr, alpha_du, dphi}.
static double *g_trace = nullptr;
extern "C" void amk__debug_trace(double *d_buf) { g_trace = d_buf; }

#ifndef AMK_SOLVE_PRIO
#define AMK_SOLVE_PRIO 3
#endif
#ifndef AMK_SOLVE_WAVES
#define AMK_SOLVE_WAVES 2  // waves per SIMD the register budget is set for (256 VGPRs each)
#endif
#ifndef AMK_SOLVE_LDS_MIN
#define AMK_SOLVE_LDS_MIN 0
#endif
// grid = S blocks of one wavefront; dynamic LDS = LdsMap(N).total reals.  R = double: the product default;
// R = float: the fp32 variant of BASELINE config C5 (amk_mpc_set_precision), same algorithm, same interfaces.
template <class R, int NT>  // NT > 0: horizon baked in (LDS offsets become immediates, as the reference bakes N into its plugin)
__device__ __forceinline__ void solve_kernel_body(int Nrt, int K, int nref, int nx, const double *__restrict__ prm,
                                                  const SolveOpts &opt, const double *__restrict__ ref_states,
                                                  double *__restrict__ w0, double *__restrict__ u_out,
                                                  double *__restrict__ x0array, int *__restrict__ info, double *trace,
                                                  const int *__restrict__ done, double *__restrict__ ref_path,
                                                  int *__restrict__ step_flags, const double *__restrict__ plan_coef,
                                                  const int *__restrict__ plan_meta, double *__restrict__ ybuf,
                                                  double *__restrict__ gains, const SolveSched sched) {
    using namespace amk32;  // solve_scene / sm_status / sm_iters overload on the scratchpad type
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    R *sm = reinterpret_cast<R *>(sm_raw);
    const int s = blockIdx.x;
    // the solves go first where a SIMD has to choose: the search / plan waves that share the CUs are fillers for the issue
    // slots these latency-bound waves leave empty (same-box A/B, 3 x alternating: lone solve launch 2070-2076 us against
    // 2092-2100, steady state 554.9-555.9 k against 549.5-554.5 k steps/s; profiles/r05_ab_build_prefetch.txt, variant D)
    __builtin_amdgcn_s_setprio(AMK_SOLVE_PRIO);
    // control step: done[s] = 1: this scene left the re-plan loop already; 2: its solve was paused by an earlier launch's
    // iteration budget (it skipped this round's queries and packing: P still holds the problem it is solving)
    const int dstate = sched.done_rw ? sched.done_rw[s] : (done ? done[s] : 0);
    if (dstate == 1) return;
    const bool resume = dstate == 2;
    const int N = NT > 0 ? NT : Nrt;
    const LdsMap L(N);
    const double *P = ref_states + (size_t)s * nref;
    SceneIO io;
    io.ref = P + SD;
    io.obs = P + SD + SD * N;
    const double *target = P + SD + SD * N + 3 * K * N;
    double *w = w0 + (size_t)s * nx;
    solve_scene(sm, L, N, K, prm, opt, P, target, io, w, w, info ? info + 4 * s : nullptr, plan_coef, plan_meta,
                ybuf + (size_t)s * N * (K > 0 ? K : 1) * kTermRecord, gains + (size_t)s * N * GAIN_STAGE, s == 0 ? trace : nullptr,
                resume, sched.budget, sched.rec ? sched.rec + (size_t)s * (8 * N + 8) : nullptr);
    __syncthreads();
    const int lane = threadIdx.x;
    if (sm_paused(sm, L)) {  // out of budget: the scene keeps its place in the re-plan loop and continues in a later launch
        if (lane == 0) sched.done_rw[s] = 2;
        return;
    }
    if (lane < UD) u_out[4 * s + lane] = sm[L.U + lane];  // sol[10..13]  HighLvlMpc.cpp:124-128
    if (x0array)                                           // rows [X_k,U_k], k < N  :130-136
        for (int e = lane; e < 14 * N; e += 64) {
            const int k = e / 14, i = e % 14;
            x0array[(size_t)s * 14 * N + e] = i < SD ? sm[L.X + k * SD + i] : sm[L.U + k * UD + (i - SD)];
        }
    if (ref_path)  // mRefPath[i] = x0Array[i][0:10], AvoidanceStateMachine.cpp:338-342
        for (int e = lane; e < SD * N; e += 64) ref_path[(size_t)s * SD * N + e] = sm[L.X + e];
    if (step_flags && lane == 0) {
        const int passes = step_flags[4 * s + 1] + 1;
        step_flags[4 * s + 1] = passes;
        step_flags[4 * s + 2] = max(step_flags[4 * s + 2], sm_status(sm, L));  // worst status over the step's solves
        step_flags[4 * s + 3] += sm_iters(sm, L);
        // per-scene scheduling (amk_step_batch with an iteration budget): the scene is free for its next pass, or has made them all
        if (sched.done_rw) sched.done_rw[s] = passes >= sched.max_passes ? 1 : 0;
    }
}

#define AMK_SOLVE_ARGS                                                                                                   \
    int Nrt, int K, int nref, int nx, const double *__restrict__ prm, SolveOpts opt,                                     \
        const double *__restrict__ ref_states, double *__restrict__ w0, double *__restrict__ u_out,                      \
        double *__restrict__ x0array, int *__restrict__ info, double *trace, const int *__restrict__ done,               \
        double *__restrict__ ref_path, int *__restrict__ step_flags, const double *__restrict__ plan_coef,               \
        const int *__restrict__ plan_meta, double *__restrict__ ybuf, double *__restrict__ gains, const SolveSched sched
#define AMK_SOLVE_PASS Nrt, K, nref, nx, prm, opt, ref_states, w0, u_out, x0array, info, trace, done, ref_path, step_flags, plan_coef, plan_meta, ybuf, gains, sched

template <int NT>
__global__ __launch_bounds__(64, AMK_SOLVE_WAVES) void mpc_solve_kernel(AMK_SOLVE_ARGS) {
    solve_kernel_body<double, NT>(AMK_SOLVE_PASS);
}
template <int NT>
__global__ __launch_bounds__(64, AMK_SOLVE_WAVES) void mpc_solve_kernel_f32(AMK_SOLVE_ARGS) {
    solve_kernel_body<float, NT>(AMK_SOLVE_PASS);
}

namespace amk {
int launch_solve(amk_mpc *m, const double *d_ref_states, double *d_u, double *d_x0array, int *d_info, const int *d_done,
                 double *d_ref_path, int *d_step_flags, hipStream_t stream, int budget, int max_passes) {
    SolveSched sched{0, 0, nullptr, nullptr};
    if (max_passes > 0) {   // per-scene scheduling of amk_step_batch: done[] is read AND written (0 free / 1 left the loop / 2 paused)
        if (!d_done || !d_step_flags) return AMK_ERR_INVALID_ARG;
        if (!m->resume_rec.p) AMK_HIP(m->resume_rec.alloc((size_t)m->S * (8 * m->N + 8)));
        sched = SolveSched{budget > 0 ? budget : 0, max_passes, m->resume_rec.p, const_cast<int *>(d_done)};
    }
    TimedLaunch tl(KC_SOLVE, stream);
#define AMK_LAUNCH_SOLVE(KERNEL, NT, LDS)                                                                            \
    hipLaunchKernelGGL(KERNEL<NT>, dim3(m->launch_scenes()), dim3(64), LDS, stream, m->N, m->K, m->nref, m->nx, m->prm.p, m->opt,      \
                       d_ref_states, m->w0.p, d_u, d_x0array, d_info, g_trace, d_done, d_ref_path, d_step_flags,         \
                       m->plan_coef.p, m->plan_meta.p, m->ybuf.p, m->gains.p, sched)
    if (m->precision == 32) {  // fp32 arithmetic, half the scratchpad
        const size_t lds = m->lds_bytes / 2;
        switch (m->N) {
            case 10: AMK_LAUNCH_SOLVE(mpc_solve_kernel_f32, 10, lds); break;
            case 20: AMK_LAUNCH_SOLVE(mpc_solve_kernel_f32, 20, lds); break;
            case 30: AMK_LAUNCH_SOLVE(mpc_solve_kernel_f32, 30, lds); break;
            default: AMK_LAUNCH_SOLVE(mpc_solve_kernel_f32, 0, lds); break;
        }
    } else {
        switch (m->N) {  // the BASELINE horizons get their own instantiation; anything else runs the generic one
            case 10: AMK_LAUNCH_SOLVE(mpc_solve_kernel, 10, m->lds_bytes); break;
            case 20: AMK_LAUNCH_SOLVE(mpc_solve_kernel, 20, m->lds_bytes); break;
            case 30: AMK_LAUNCH_SOLVE(mpc_solve_kernel, 30, m->lds_bytes); break;
            default: AMK_LAUNCH_SOLVE(mpc_solve_kernel, 0, m->lds_bytes); break;
        }
    }
#undef AMK_LAUNCH_SOLVE
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}
}  // namespace amk

// internal (diagnostics): host copy of the Riccati plan's LDS indices, [PLAN_ITEMS][PLAN_TERMS + 4] ints + lane roles
extern "C" int amk__plan_dump(amk_mpc *m, int *h_meta, int n_ints) {
    std::vector<double> coef;
    std::vector<int> meta;
    int st = build_plan(m, coef, meta);
    if (st != AMK_OK) return st;
    const LdsMap L(m->N);
    const int hdr[8] = {L.P, L.p, L.lam, L.M, L.Hm, L.G, L.q, L.total};
    for (int i = 0; i < 8 && i < n_ints; ++i) h_meta[i] = hdr[i];
    for (int i = 0; i + 8 < n_ints && i < (int)meta.size(); ++i) h_meta[8 + i] = meta[i];
    return (int)meta.size() + 8;
}

// internal: resident solve blocks per CU as the runtime computes it (diagnostics)
extern "C" int amk__solve_occupancy(amk_mpc *m) {
    int nb = -1;
    const void *f = m->N == 10 ? (const void *)mpc_solve_kernel<10> : m->N == 20 ? (const void *)mpc_solve_kernel<20>
                  : m->N == 30 ? (const void *)mpc_solve_kernel<30> : (const void *)mpc_solve_kernel<0>;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, f) == hipSuccess)
        printf("solve kernel N=%d: numRegs %d, sharedSizeBytes %zu, localSizeBytes %zu, maxDynamicShared %d, lds request %zu\n", m->N,
               fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxDynamicSharedSizeBytes, m->lds_bytes);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 64, m->lds_bytes) != hipSuccess) return -2;
    return nb;
}

extern "C" {

int amk_mpc_create(double T, double dt, int nearest_point_num, int n_scenes, amk_mpc **out) {
    if (!out || !(T > 0) || !(dt > 0) || nearest_point_num < 0 || n_scenes <= 0) return AMK_ERR_INVALID_ARG;
    *out = nullptr;
    const int N = (int)(T / dt);  // HighLvlMpc.cpp:9 (mN = T / dt, truncating)
    if (N < 2 || N > AMK_MAX_HORIZON || nearest_point_num > AMK_MAX_K) return AMK_ERR_UNSUPPORTED;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    amk_mpc *m = new amk_mpc();
    m->T = T; m->dt = dt; m->N = N; m->K = nearest_point_num; m->S = n_scenes;
    m->nx = 10 + 14 * N;
    m->nref = 20 + 10 * N + 3 * m->K * N;
    std::memset(m->h_prm, 0, sizeof m->h_prm);
    // constructor defaults, HighLvlMpc.cpp:13-16,53-56
    const double wdef[25] = {100, 100, 100, 300, 1, 1, 1, 0., 0., 0., 0.0, 10, 10, 30, 0, 1, 1, 0., 0., 0., 1., 1., 1., 1., 1.};
    std::memcpy(m->h_prm + PRM_W, wdef, sizeof wdef);
    const double tdef[4] = {0.01, 0.01, 0.01, 0};
    std::memcpy(m->h_prm + PRM_TAU, tdef, sizeof tdef);
    for (int i = 0; i < 4; ++i) m->h_prm[PRM_GAIN + i] = 1.0;
    const double lb[4] = {-10., -10., 1., -10.}, ub[4] = {10., 10., 20., 10.};
    std::memcpy(m->h_prm + PRM_LB, lb, sizeof lb);
    std::memcpy(m->h_prm + PRM_UB, ub, sizeof ub);
    refresh_dynamics(m);
    // solver options.  tol is ipopt.tol (HighLvlMpc.cpp:19); max_iter counts the
    // iterations of this project's method, not IPOPT's 10 (HighLvlMpc.cpp:20)
    m->opt.tol = 1e-4; m->opt.max_iter = AMK_MPC_DEFAULT_MAX_ITER; m->opt.max_ls = 12; m->opt.mu_init = 0.1;
    m->opt.bound_push = 1e-3; m->opt.bound_frac = 1e-3; m->opt.kappa_mu = 0.2; m->opt.kappa_eps = 100.0;
    m->opt.mu_min_fac = 1e-2; m->opt.maj = 1.0; m->opt.tau_min = 0.99;
    m->opt.eta_phi = 1e-8; m->opt.s_max = 100.0; m->opt.kappa_sigma = 1e10;
    m->lds_bytes = sizeof(double) * (size_t)LdsMap(N).total;
    if (m->lds_bytes < (size_t)AMK_SOLVE_LDS_MIN) m->lds_bytes = AMK_SOLVE_LDS_MIN;
    hipError_t e;
    if ((e = m->prm.alloc(PRM_LEN)) != hipSuccess ||
        (e = m->plan_coef.alloc((size_t)PLAN_ITEMS * (PLAN_TERMS + 1) + 64 * 2)) != hipSuccess ||
        (e = m->plan_meta.alloc((size_t)PLAN_ITEMS * (PLAN_TERMS + 4) + 64 * LANE_META_INTS)) != hipSuccess || (e = m->w0.alloc((size_t)n_scenes * m->nx)) != hipSuccess ||
        (e = m->ybuf.alloc((size_t)n_scenes * N * (m->K > 0 ? m->K : 1) * kTermRecord)) != hipSuccess ||
        (e = m->gains.alloc((size_t)n_scenes * N * GAIN_STAGE)) != hipSuccess ||
        (e = hipMemset(m->gains.p, 0, sizeof(double) * (size_t)n_scenes * N * GAIN_STAGE)) != hipSuccess ||
        (e = hipMemset(m->w0.p, 0, sizeof(double) * (size_t)n_scenes * m->nx)) != hipSuccess ||
        (e = hipFuncSetAttribute(N == 10   ? (const void *)mpc_solve_kernel<10>
                                 : N == 20 ? (const void *)mpc_solve_kernel<20>
                                 : N == 30 ? (const void *)mpc_solve_kernel<30>
                                           : (const void *)mpc_solve_kernel<0>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes)) != hipSuccess) {
        delete m;
        return amk::hip_fail(e);
    }
    int st = upload_params(m);
    if (st != AMK_OK) {
        delete m;
        return st;
    }
    *out = m;
    return AMK_OK;
}

int amk_mpc_destroy(amk_mpc *m) {
    if (!m) return AMK_ERR_INVALID_ARG;
    delete m;
    return AMK_OK;
}

int amk_mpc_horizon(const amk_mpc *m) { return m ? m->N : -1; }
int amk_mpc_nx(const amk_mpc *m) { return m ? m->nx : -1; }
int amk_mpc_ref_len(const amk_mpc *m) { return m ? m->nref : -1; }

int amk_mpc_setup_weights(amk_mpc *m, const double *w) {
    if (!m || !w) return AMK_ERR_INVALID_ARG;
    std::memcpy(m->h_prm + PRM_W, w, sizeof(double) * 25);
    return upload_params(m);
}
int amk_mpc_setup_tau(amk_mpc *m, const double *tau) {
    if (!m || !tau) return AMK_ERR_INVALID_ARG;
    std::memcpy(m->h_prm + PRM_TAU, tau, sizeof(double) * 4);
    refresh_dynamics(m);
    return upload_params(m);
}
// The reference's use_drag_coefficient switch in the one reading under which it is well defined (see ode_host): v' = a - k .* v.
// Changes A (the v <- v and p <- v entries) and nothing of its sparsity; the Riccati plan is rebuilt from the new values.
int amk_mpc_set_drag_coefficient(amk_mpc *m, double kx, double ky, double kz) {
    if (!m || !(kx >= 0.0 && ky >= 0.0 && kz >= 0.0) || !(kx < 1e3 && ky < 1e3 && kz < 1e3)) return AMK_ERR_INVALID_ARG;
    m->drag[0] = kx; m->drag[1] = ky; m->drag[2] = kz;
    m->ev_dtau_valid = false;   // d(A, B, c)/d tau depends on the drag too
    refresh_dynamics(m);
    return upload_params(m);
}
int amk_mpc_setup_gains(amk_mpc *m, const double *g) {
    if (!m || !g) return AMK_ERR_INVALID_ARG;
    std::memcpy(m->h_prm + PRM_GAIN, g, sizeof(double) * 4);
    return upload_params(m);
}
int amk_mpc_set_drone_radius(amk_mpc *m, double r) {
    if (!m) return AMK_ERR_INVALID_ARG;
    m->h_prm[PRM_RADIUS] = r;
    return upload_params(m);
}
int amk_mpc_set_drone_accel_limits(amk_mpc *m, double aMinZ, double aMaxZ, double aMaxXy, double aMaxYawDot) {
    if (!m) return AMK_ERR_INVALID_ARG;
    const double lb[4] = {-aMaxXy, -aMaxXy, aMinZ, -aMaxYawDot}, ub[4] = {aMaxXy, aMaxXy, aMaxZ, aMaxYawDot};
    for (int i = 0; i < 4; ++i)
        if (!(ub[i] > lb[i])) return AMK_ERR_INVALID_ARG;
    std::memcpy(m->h_prm + PRM_LB, lb, sizeof lb);
    std::memcpy(m->h_prm + PRM_UB, ub, sizeof ub);
    return upload_params(m);
}
int amk_mpc_set_solver_options(amk_mpc *m, double tol, int max_iter) {
    if (!m || !(tol > 0) || max_iter < 0) return AMK_ERR_INVALID_ARG;
    m->opt.tol = tol;
    m->opt.max_iter = max_iter;
    return AMK_OK;
}

int amk_mpc_set_solve_budget(amk_mpc *m, int budget, int budget_rounds) {
    if (!m || budget < 0 || budget_rounds < 0 || budget_rounds > AMK_MAX_BUDGET_ROUNDS) return AMK_ERR_INVALID_ARG;
    m->solve_budget = budget;
    m->budget_rounds = budget_rounds;
    return AMK_OK;
}

int amk_mpc_set_precision(amk_mpc *mpc, int bits) {
    if (!mpc) return AMK_ERR_INVALID_ARG;
    if (bits != 32 && bits != 64) return AMK_ERR_UNSUPPORTED;
    mpc->precision = bits;
    return AMK_OK;
}

int amk_mpc_solve(amk_mpc *m, const double *d_ref_states, double *d_u, double *d_x0array, int *d_info, int faster,
                  void *stream) {
    (void)faster;  // mSolver and mSolverFaster carry identical options, HighLvlMpc.cpp:50-52
    if (!m || !d_ref_states || !d_u) return AMK_ERR_INVALID_ARG;
    return amk::launch_solve(m, d_ref_states, d_u, d_x0array, d_info, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int amk_mpc_get_warm_start(amk_mpc *m, double *d_w, void *stream) {
    if (!m || !d_w) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipMemcpyAsync(d_w, m->w0.p, sizeof(double) * (size_t)m->S * m->nx, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AMK_OK;
}
int amk_mpc_set_warm_start(amk_mpc *m, const double *d_w, void *stream) {
    if (!m || !d_w) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipMemcpyAsync(m->w0.p, d_w, sizeof(double) * (size_t)m->S * m->nx, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AMK_OK;
}
int amk_mpc_reset_warm_start(amk_mpc *m, void *stream) {
    if (!m) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipMemsetAsync(m->w0.p, 0, sizeof(double) * (size_t)m->S * m->nx, (hipStream_t)stream));
    return AMK_OK;
}

int amk_mpc_solve_host(amk_mpc *m, const double *h_ref_states, double *h_u, double *h_x0array, int *h_info, int faster) {
    if (!m || !h_ref_states || !h_u) return AMK_ERR_INVALID_ARG;
    const size_t S = m->S;
    if (!m->st_ref.p) {
        AMK_HIP(m->st_ref.alloc(S * m->nref));
        AMK_HIP(m->st_u.alloc(S * 4));
        AMK_HIP(m->st_x0.alloc(S * 14 * m->N));
        AMK_HIP(m->st_info.alloc(S * 4));
    }
    AMK_HIP(hipMemcpy(m->st_ref.p, h_ref_states, sizeof(double) * S * m->nref, hipMemcpyHostToDevice));
    int st = amk_mpc_solve(m, m->st_ref.p, m->st_u.p, m->st_x0.p, m->st_info.p, faster, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipDeviceSynchronize());
    AMK_HIP(hipMemcpy(h_u, m->st_u.p, sizeof(double) * S * 4, hipMemcpyDeviceToHost));
    if (h_x0array) AMK_HIP(hipMemcpy(h_x0array, m->st_x0.p, sizeof(double) * S * 14 * m->N, hipMemcpyDeviceToHost));
    if (h_info) AMK_HIP(hipMemcpy(h_info, m->st_info.p, sizeof(int) * S * 4, hipMemcpyDeviceToHost));
    return AMK_OK;
}

}  // extern "C"
