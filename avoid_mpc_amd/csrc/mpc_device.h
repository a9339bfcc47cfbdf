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
// primal-dual interior point on the controls, the non-smooth |v.n| of every collision term carried
// as an l1 term (epigraph variable at its closed-form optimum, two multipliers per term, same
// barrier parameter); every Newton system is an LQR problem solved by a backward Riccati sweep
// (stage blocks: 10x10 state, 4x4 control) + forward roll; inertia correction by a uniform
// diagonal shift when a 4x4 block is not positive definite; Armijo backtracking on the (smooth)
// barrier function; monotone barrier update; stop when the last barrier problem is solved to tol.  The stacked shooting Jacobian is block-banded with
// 19-nnz / 10-nnz constant blocks, so there is no dense GEMM here and MFMA does not apply; the
// work is 64-lane fp64 VALU with LDS as the per-scene scratchpad.
#pragma once
#include "amk_common.h"
#include "fast_math.h"

namespace amk {

// Per-phase clocks / per-iteration trace of the solve (scratch/dbg2.py); compiled in only with
// -DAMK_SOLVE_TRACE (AMK_SOLVE_TRACE=1 python -m avoid_mpc_amd.build --force).
#ifdef AMK_SOLVE_TRACE
#define AMK_CLK() wall_clock64()
constexpr bool kTrace = true;
#else
#define AMK_CLK() 0LL
constexpr bool kTrace = false;
#endif

constexpr int SD = AMK_S_DIM;  // 10
constexpr int UD = AMK_U_DIM;  // 4

// parameter block in device memory (doubles)
constexpr int PRM_W = 0;        // weights[25]: Qgoal[10] Qpen[10] Qu[4] lambda
constexpr int PRM_TAU = 25;     // tau[4]
constexpr int PRM_GAIN = 29;    // gains[4]  (parsed but unused by the dynamics, :91-93,114-121)
constexpr int PRM_RADIUS = 33;  // drone radius
constexpr int PRM_LB = 34;      // control lower bounds[4]
constexpr int PRM_UB = 38;      // control upper bounds[4]
constexpr int PRM_C = 42;       // c[10]
constexpr int PRM_LDS_LEN = 52; // what the solve keeps in its scratchpad: everything above (A, B are read from global
                                // memory into registers by the sweeps that need them: 140 doubles of LDS per scene)
constexpr int PRM_A = 52;       // A[10][10] row-major
constexpr int PRM_B = 152;      // B[10][4]
constexpr int PRM_LEN = 192;

struct SolveOpts {     // DESIGN.md section 5; the CPU restatement used by the tests carries the same defaults
    double tol;       // ipopt.tol            HighLvlMpc.cpp:19
    int max_iter;     // iteration cap of THIS method (default AMK_MPC_DEFAULT_MAX_ITER); the reference's ipopt.max_iter = 10 counts IPOPT's
    int max_ls;       // 12
    double mu_init;   // 0.1  (IPOPT default)
    double bound_push, bound_frac;  // 1e-3 (IPOPT warm_start_bound_push / _frac)
    double kappa_mu;  // 0.2: mu <- max(mu_min, min(kappa_mu mu, mu^1.5))
    double kappa_eps; // 100: barrier update when E_mu <= kappa_eps mu
    double mu_min_fac;  // 1e-2: mu_min = mu_min_fac * tol
    double maj;       // 1: weight of the majoriser curvature of |s| at mu_init (fades with mu / mu_init)
    double tau_min;   // 0.99
    double eta_phi;   // 1e-8
    double s_max;     // 100
    double kappa_sigma;  // 1e10
};

// LDS carve-up for one scene (offsets in doubles).  All per-scene state of the solve lives here.
// The feedback gains of the Riccati sweep (4 x 10 + feed-forward per stage) live in a global, L2-resident scratch, not
// in LDS: written once by the backward sweep, read once by the forward roll, 7.7 KB of the former 26.3 KB per scene --
// with them in LDS a CU held 6 scenes, without them 8 (two per SIMD).  Only the entries that are not structurally zero are
// stored (round 4; rounds 2-3 kept [k][4][16] = 64 doubles per stage): the yaw chain is decoupled (build_plan checks it), so
// K_k[a][yaw] = 0 for the three acceleration rows and K_k[yaw_dot][j] = 0 for every state but yaw -- 3 x (9 states +
// feed-forward) + (yaw, feed-forward) = 32 doubles = two cache lines per stage instead of four.  gain_offset(a, c): where
// K_k[a][c] lives (c = 10: feed-forward), -1 for a structural zero.
constexpr int kTermRecord = 4;  // doubles per collision term in the global scratch (mpc_device_impl.h: YB)
constexpr int GAIN_STAGE = 32;
__host__ __device__ __forceinline__ int gain_offset(int a, int c) {
    if (a < 3) return c == 3 ? -1 : a * 10 + (c < 3 ? c : c - 1);
    return c == 3 ? 30 : (c == 10 ? 31 : -1);
}

struct LdsMap {
    int prm, xinit, target, cy, sy;
    int X, U, zl, zu, dX, dU, dzl, dzu, Xt, Ut;
    int q, r, rb, Rb, gU, H6, rotQ;
    int P, p, lam, M, Hm, G, Atp, Atl, qu, Y, Z, D, red;
    int total;
    __host__ __device__ explicit LdsMap(int N, int prm_len = PRM_LDS_LEN) {
        int o = 0;
        auto take = [&](int n) { int b = o; o += (n + 1) & ~1; return b; };
        prm = take(prm_len); xinit = take(SD); target = take(SD); cy = take(N); sy = take(N);
        X = take((N + 1) * SD); U = take(N * UD); zl = take(N * UD); zu = take(N * UD);
        dX = take((N + 1) * SD); dU = take(N * UD);
        q = take((N + 1) * SD); r = take(N * UD); rb = take(N * UD); Rb = take(N * UD); gU = take(N * UD);
        H6 = take(N * 21); rotQ = take(N * 6);
        // the backward sweep's temporaries (P .. qu: 250 reals, born and dead inside riccati_backward) live in the trial
        // states' storage (Xt: written and read by the line search only): 2 KB less per scene, which at N = 30 is the
        // difference between 5 and 6 scenes per CU (28.2 -> 26.2 KB of the 160).  M / Hm / G hold structural zeros the sweep
        // relies on: solve_scene clears them before every iteration's first sweep.
        constexpr int kRic = 100 + 3 * SD + 56 + 10 + 40 + SD + UD;
        Xt = take((N + 1) * SD > kRic ? (N + 1) * SD : kRic);
        P = Xt; p = P + 100; lam = p + SD; M = lam + SD; Hm = M + 56; G = Hm + 10;
        Atp = G + 40; Atl = Atp + SD; qu = Atl + SD; Y = 0; Z = 0; D = 0;
        Ut = take(N * UD); red = take(16);
        // buffers with disjoint lifetimes share storage: the barrier-shifted control gradient / Hessian (rb, Rb) are
        // dead once the backward sweep has run; the dual steps are born after the forward roll and consumed by the
        // update that follows the line search.  (Not on q / r: the first trial point is evaluated WITH derivatives,
        // into q, r, H6, so that an accepted full step needs no second evaluation.)
        dzl = rb; dzu = Rb;
        total = o;
    }
};

constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8 lanes
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i inside each row of 16

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

// ---- Riccati plan: which LDS cells each lane combines in "round A" of a backward stage.  The affine
// dynamics are constant, so every entry of A'PA, B'PB, B'PA, A'p, A'lam, B'p, B'lam is a fixed <= 9-term
// linear combination of entries of P / p / lam; the host enumerates the terms once per handle
// (mpc_solve.hip: build_plan) from the non-zeros of A (19) and B (10).  128 items = 2 per lane.
constexpr int PLAN_ITEMS = 128;
constexpr int PLAN_TERMS = 9;
constexpr int PLAN_TERMS_LIGHT = 3;   // the host sorts the items by term count: the second item of every lane has <= 3 terms
struct PlanItemMeta {            // one per item, ints
    int idx[PLAN_TERMS];         // LDS offsets (doubles) of the source cells (unused terms: the zero cell)
    int out;                     // LDS offset of the result (+ k * out_kstride)
    int out_kstride;
    int aux;                     // LDS offset of a per-stage addend (+ k * aux_kstride); the zero cell if none
    int aux_kstride;
};
// coefficients: [PLAN_ITEMS][PLAN_TERMS + 1] doubles, the last one multiplies the regularisation shift delta

// What a lane does in rounds B/C of a backward stage (host-built, build_plan): value = base - G(:,ci)' Hm^-1 g,
// with g = G(:,cj) for an entry of P and g = qu for an entry of p.  Everything is an index so that the round is
// branch-free: lanes without a role read the zero cell and write the dummy cell.
constexpr int LANE_META_INTS = 16;
struct LaneRole {
    int gi, gi_stride;           // column "i": sm[gi + a * gi_stride], a < 4
    int gj, gj_stride;           // column "j"
    int b_idx[3], b_ks[3];       // base = bconst + bdelta * delta + sum_n sm[b_idx[n] + k * b_ks[n]]
    int out1, out2;              // where the value goes (lower-triangle P[i][j] / p[i]); out2: unused since only the lower triangle of P is kept current
    int gain_col;                // >= 0: this lane also stores column gain_col of the gains (-Hm^-1 g)
    int lam_src, lam_dst;        // unused (the adjoint recursion is no longer carried by this sweep: adjoint_sweep's scans)
    int pad;
};

// Per-launch scheduling of the solve inside the control step (mpc_solve.hip: solve_kernel_body; step.hip: amk_step_batch).
// budget > 0: interior-point iterations a scene may make in THIS launch before it pauses (state to rec, done_rw[s] = 2) and is
// picked up by a later launch; max_passes: the step's mpc_max_iter (a scene that has made them all gets done_rw[s] = 1).
// All zero / null: the plain solve (amk_mpc_solve, amk_step_batch_frames).
struct SolveSched {
    int budget, max_passes;
    double *rec;     // [S][8 N + 8]: zl, zu, {mu, delta_last, it, n_reg, ls_fail}
    int *done_rw;
};

struct SceneIO {
    const double *ref;   // [N][10] reference states            (P[10 : 10+10N])
    const double *obs;   // [N][K][3] obstacle points           (P[10+10N : ...])
};

}  // namespace amk

// The device algorithm itself is in mpc_device_impl.h, compiled twice: in fp64 (namespace amk, the product
// default and the arithmetic of the reference's CasADi/IPOPT path) and in fp32 (namespace amk32, BASELINE
// config C5's "fp32 tolerance check": kNN distances stay fp64, only the NLP/interior-point arithmetic narrows).
#define AMK_RNS amk
#define AMK_REAL double
#define AMK_REAL_F32 0
#include "mpc_device_impl.h"
#undef AMK_RNS
#undef AMK_REAL
#undef AMK_REAL_F32
#define AMK_RNS amk32
#define AMK_REAL float
#define AMK_REAL_F32 1
#include "mpc_device_impl.h"
#undef AMK_RNS
#undef AMK_REAL
#undef AMK_REAL_F32
