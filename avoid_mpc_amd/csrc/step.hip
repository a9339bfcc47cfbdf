// One control step for a batch of scenes on gfx950: the TASK branch of AvoidanceStateMachine::Step
// (AM/src/AvoidanceStateMachine.cpp:322-355) with the single-frame FrameKDMap queries it makes
// (AM/src/FrameKDMap.cpp:254-275,322-427), as a fixed sequence of kernels on one stream -- no host
// round trip between the dual-KD-tree queries, the packing of P and the solves.
//
// Per outer iteration (<= mpc_max_iter):
//   step_knn_grid_kernel   one launch for both trees: the N K-NN queries at the reference points in the obstacle index
//                          (:204-215) and the 1-NN of reference point 0 in the edge index (:270), through the bucketed
//                          indices (step_scan_kernel<5> / <1>: the same through the streaming-scan cross-check path)
//   step_plan_pack_kernel  PlanWapionts: nearest-obstacle test, snap to the edge point, re-query (:259-281), then
//                          ProcessWaypoints padding/needReplan, early exit, GetRefStates      (:216-257,333-335)
//   mpc_solve_kernel       Solve + refill of the reference path                                (:337-342)
// The multi-frame map (keyframes, PtIsInFrame fast path, per-frame merge) is step_frames.hip.
#include <type_traits>

#include "kd_exact.h"
#include "mpc_handle.h"

using namespace amk;

extern "C" int amk__kd_ensure_soa(amk_kd *kd, void *stream);  // kd_index.hip

namespace {

__global__ void step_begin_kernel(int S, int *__restrict__ done, int *__restrict__ flags, double *__restrict__ u) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    done[s] = 0;
    flags[4 * s + 0] = 1;   // isSafety = true (:326)
    flags[4 * s + 1] = 0;
    flags[4 * s + 2] = -1;
    flags[4 * s + 3] = 0;
    u[4 * s + 0] = u[4 * s + 1] = u[4 * s + 2] = u[4 * s + 3] = 0.0;
}

// Raw k nearest neighbours (nanoflann's own answer: min(k, size) entries, no adaptor count rule) of
// n_queries reference points per scene; queries are read in place from the reference path
// (stride 10 doubles).  Empty slots: distance DBL_MAX.
template <int QPW>
__global__ __launch_bounds__(512) void step_scan_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                        const float *__restrict__ Z, int cap,
                                                        const int *__restrict__ sizes,
                                                        const float *__restrict__ pmaxs, int n_scenes,
                                                        const double *__restrict__ ref_path, int N, int n_queries,
                                                        int k, float *__restrict__ out_pts,
                                                        double *__restrict__ out_d2, const int *__restrict__ done) {
    const int groups = (n_queries + QPW - 1) / QPW;
    const int wpb = blockDim.x >> 6;
    const int bps = (groups + wpb - 1) / wpb;
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = (j / bps) * 8 + xcd;  // all blocks of a scene share one XCD's L2; its waves one CU's L1
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */;
    const int g = (j % bps) * wpb + w;
    if (s >= n_scenes || g >= groups || done[s]) return;
    const int lane = threadIdx.x & 63;
    const int size = sizes[s];
    const float *xs = X + (size_t)s * cap, *ys = Y + (size_t)s * cap, *zs = Z + (size_t)s * cap;
    extern __shared__ __attribute__((aligned(16))) unsigned char scan_smem[];
    ScanLds<QPW> *ws = reinterpret_cast<ScanLds<QPW> *>(scan_smem) + w;
    double *qt = reinterpret_cast<double *>(reinterpret_cast<ScanLds<QPW> *>(scan_smem) + wpb) + w * QPW * 3;
    const int q0 = g * QPW;
    const int nvalid = n_queries - q0 < QPW ? n_queries - q0 : QPW;
    if (lane < QPW * 3) {  // queries = positions of the reference points (stride 10 in mRefPath)
        const int qq = lane / 3 < nvalid ? lane / 3 : nvalid - 1;
        qt[lane] = ref_path[((size_t)s * N + q0 + qq) * SD + lane % 3];
    }
    scan_cloud<QPW>(xs, ys, zs, size, pmaxs[s], qt, 3, k, ws);
    for (int qq = 0; qq < nvalid; ++qq) {
        const size_t row = (size_t)s * n_queries + q0 + qq;
        if (lane < k) {
            const int li = ws->li[qq][lane];
            const bool ok = li != kNoIndex;
            out_d2[row * k + lane] = ok ? ws->ld[qq][lane] : DBL_MAX;
            float *o = out_pts + (row * k + lane) * 3;
            o[0] = ok ? xs[li] : 0.f;
            o[1] = ok ? ys[li] : 0.f;
            o[2] = ok ? zs[li] : 0.f;
        }
    }
}

// Same outputs through the bucketed indices (kd_grid.h), both trees in one launch: wavefront q < N answers
// the K-NN of reference point q in the obstacle index, wavefront q == N the 1-NN of reference point 0 in
// the edge index (the Edge-KD-tree query of PlanWapionts, :270).
// <= 48 VGPRs: a CU that holds its 8 solve waves (2 x 232 registers per SIMD, 142.6 of 160 KB of LDS since round 5) still
// has 48 registers per SIMD and 17 KB of LDS free -- exactly one block of this kernel (one wave per SIMD), which then runs
// in the issue slots the latency-bound solves leave empty instead of waiting for a CU to drain.
__global__ __launch_bounds__(256) void step_knn_grid_kernel(GridPtrs gobs, GridPtrs gedge, int n_scenes,
                                                            const double *__restrict__ ref_path, int N, int K,
                                                            float *__restrict__ knn_pts, double *__restrict__ knn_d2,
                                                            float *__restrict__ edge_pt, double *__restrict__ edge_d2,
                                                            const int *__restrict__ done) {
    __shared__ GridWaveLds wl[4];
    const int nq = N + 1;
    const int bps = (nq + 3) / 4;
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = (j / bps) * 8 + xcd;
    // wave-uniform by construction; telling the compiler so moves the scene pointers, the grid geometry and the query into
    // SGPRs (scalar loads) -- 14 VGPRs less, which is what lets the kernel fit in 64
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = (j % bps) * 4 + w;
    if (s >= n_scenes || q >= nq || done[s]) return;
    const bool is_edge = q == N;
    const double *qp = ref_path + ((size_t)s * N + (is_edge ? 0 : q)) * SD;  // read in place from mRefPath
    const int k = is_edge ? 1 : K;
    double ld;
    int li, lpos;
    const GridScene gs = is_edge ? gedge.scene(s) : gobs.scene(s);
    grid_knn(gs, qp[0], qp[1], qp[2], k, ld, li, lpos, &wl[w]);
    if (lane < k) {
        const bool ok = li != kNoIndex;
        const float4 rec = gs.pt[lpos];  // the neighbour's coordinates (lpos = 0 for an empty slot: a valid address)
        if (is_edge) {
            edge_d2[s] = ok ? ld : DBL_MAX;
            edge_pt[3 * s + 0] = ok ? rec.x : 0.f;
            edge_pt[3 * s + 1] = ok ? rec.y : 0.f;
            edge_pt[3 * s + 2] = ok ? rec.z : 0.f;
        } else {
            const size_t row = (size_t)s * N + q;
            knn_d2[row * K + lane] = ok ? ld : DBL_MAX;
            float *o = knn_pts + (row * K + lane) * 3;
            o[0] = ok ? rec.x : 0.f;
            o[1] = ok ? rec.y : 0.f;
            o[2] = ok ? rec.z : 0.f;
        }
    }
}

// Handles in AMK_TIES_NANOFLANN mode: the same raw results by nanoflann's own traversal of its own tree (kd_exact.h), one
// WAVEFRONT per (scene, query), overwriting what step_knn_grid_kernel wrote wherever the tree is available.  With exact ties
// (quantised edge clouds) this is what keeps the snapped edge point and the neighbour SET equal to the reference's.
__global__ __launch_bounds__(256) void step_knn_exact_kernel(ExactPtrs eobs, ExactPtrs eedge, int use_obs, int use_edge,
                                                             int n_scenes, const double *__restrict__ ref_path, int N, int K,
                                                             float *__restrict__ knn_pts, double *__restrict__ knn_d2,
                                                             float *__restrict__ edge_pt, double *__restrict__ edge_d2,
                                                             const int *__restrict__ done) {
    __shared__ ExactWaveStack stacks[4];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + w;
    const int nq = N + 1;
    if (t >= n_scenes * nq) return;
    const int s = t / nq, q = t - s * nq;
    if (done[s]) return;
    const bool is_edge = q == N;
    if (is_edge ? !use_edge : !use_obs) return;
    const double *qp = ref_path + ((size_t)s * N + (is_edge ? 0 : q)) * SD;
    const ExactTree T = is_edge ? eedge.scene(s) : eobs.scene(s);
    const int k = is_edge ? 1 : K;
    double rd;
    int ri;
    const int got = exact_knn_wave(T, qp[0], qp[1], qp[2], k, rd, ri, &stacks[w]);
    if (got < 0) return;
    if (lane < k) {
        const int j = lane;
        const bool ok = j < got;
        const float px = ok ? T.x[ri] : 0.f, py = ok ? T.y[ri] : 0.f, pz = ok ? T.z[ri] : 0.f;
        if (is_edge) {
            edge_d2[s] = ok ? rd : DBL_MAX;
            edge_pt[3 * s + 0] = px; edge_pt[3 * s + 1] = py; edge_pt[3 * s + 2] = pz;
        } else {
            const size_t row = (size_t)s * N + q;
            knn_d2[row * K + j] = ok ? rd : DBL_MAX;
            float *o = knn_pts + (row * K + j) * 3;
            o[0] = px; o[1] = py; o[2] = pz;
        }
    }
}

// PlanWapionts (:259-281) for reference point 0; called by the one wavefront that owns scene s.
template <bool EXACT, bool GRID>
__device__ __forceinline__ void plan_scene(int s, GridPtrs gpt, ExactPtrs eobs,
                                                          const float *__restrict__ X, const float *__restrict__ Y,
                                                          const float *__restrict__ Z, int cap,
                                                          const int *__restrict__ sizes_obs,
                                                          const float *__restrict__ pmax_obs,
                                                          const int *__restrict__ sizes_edge, int N, int K,
                                                          double safety_distance, double *__restrict__ ref_path,
                                                          float *__restrict__ knn_pts, double *__restrict__ knn_d2,
                                                          const float *__restrict__ edge_pt,
                                                          const double *__restrict__ edge_d2,
                                                          int *__restrict__ flags) {
    const int lane = threadIdx.x;
    const int size_o = sizes_obs[s];
    // GetNearestDistance (FrameKDMap.cpp:400-427): SearchForNearest(p, 1) -> no result unless the cloud
    // holds more than one point (kd_tree_two.h:119-124); DBL_MAX then.
    const double d2n = (size_o > 1) ? knn_d2[(size_t)s * N * K] : DBL_MAX;
    const double nearest = sqrt(d2n);
    int is_safety = 1;
    if (!(nearest > safety_distance)) {
        // QueryNearest(p1, 1, edgePts, distances, true) (:270): one result iff the edge cloud has > 1 point
        const bool has_edge = sizes_edge[s] > 1 && edge_d2[s] < DBL_MAX;
        if (!has_edge) {
            is_safety = 0;
        } else {
            double *p1 = ref_path + (size_t)s * N * SD;
            const double ex = (double)edge_pt[3 * s + 0], ey = (double)edge_pt[3 * s + 1], ez = (double)edge_pt[3 * s + 2];
            // the snapped point is what ProcessWaypoints queries next (:210-215): redo query 0
            const float *xs = X + (size_t)s * cap, *ys = Y + (size_t)s * cap, *zs = Z + (size_t)s * cap;
            // GRID is a template parameter: the streaming-scan cross-check path (amk__kd_set_mode) brought its registers and
            // LDS into the default kernel (93 VGPRs; 58 without it -- few enough to run beside the solves' waves)
            __shared__ typename std::conditional<GRID, int, ScanLds<1>>::type ws1_store;
            __shared__ typename std::conditional<GRID, GridWaveLds, int>::type wl1_store;
            __shared__ double q1[3];
            double gld = DBL_MAX;
            int gli = kNoIndex, glpos = 0;
            const GridScene gs = gpt.scene(s);
            int sli = kNoIndex;
            double sld = DBL_MAX;
            if constexpr (GRID) {
                grid_knn(gs, ex, ey, ez, K, gld, gli, glpos, &wl1_store);
            } else {
                ScanLds<1> &ws1 = ws1_store;
                q1[0] = ex; q1[1] = ey; q1[2] = ez;  // every lane stores the same values
                scan_cloud<1>(xs, ys, zs, size_o, pmax_obs[s], q1, 3, K, &ws1);
                if (lane < K) { sli = ws1.li[0][lane]; sld = ws1.ld[0][lane]; }
            }
            if (lane < K) {
                const int li = GRID ? gli : sli;
                const bool ok = li != kNoIndex;
                knn_d2[(size_t)s * N * K + lane] = ok ? (GRID ? gld : sld) : DBL_MAX;
                float *o = knn_pts + ((size_t)s * N * K + lane) * 3;
                float nx = 0.f, ny = 0.f, nz = 0.f;
                if (GRID) {
                    const float4 rec = gs.pt[glpos];
                    nx = rec.x; ny = rec.y; nz = rec.z;
                } else if (ok) {
                    nx = xs[li]; ny = ys[li]; nz = zs[li];
                }
                o[0] = ok ? nx : 0.f;
                o[1] = ok ? ny : 0.f;
                o[2] = ok ? nz : 0.f;
            }
            if (EXACT) {  // AMK_TIES_NANOFLANN: the re-query by the reference's own traversal (lane 0), as above
                __shared__ double xr[AMK_MAX_K];
                __shared__ int xi[AMK_MAX_K], xgot;
                __shared__ ExactStackStorage xstack;  // LDS, not scratch: one lane walks the tree
                const ExactTree T = eobs.scene(s);
                if (lane == 0) xgot = exact_knn_thread(T, ex, ey, ez, K, xr, xi, xstack.view());
                __syncthreads();
                if (xgot >= 0 && lane < K) {
                    const bool ok = lane < xgot;
                    knn_d2[(size_t)s * N * K + lane] = ok ? xr[lane] : DBL_MAX;
                    float *o = knn_pts + ((size_t)s * N * K + lane) * 3;
                    o[0] = ok ? T.x[xi[lane]] : 0.f;
                    o[1] = ok ? T.y[xi[lane]] : 0.f;
                    o[2] = ok ? T.z[xi[lane]] : 0.f;
                }
            }
            if (lane == 0) {
                p1[0] = ex;
                p1[1] = ey;
                p1[2] = ez;
            }
        }
    }
    if (lane == 0) flags[4 * s + 0] = is_safety;
}

// ProcessWaypoints' padding and needReplan (:216-231), the early exit (:333-335) and GetRefStates
// (:236-257); called by the one wavefront that owns scene s.
__device__ __forceinline__ void pack_scene(int s, const int *__restrict__ sizes_obs, int N, int K, int nref,
                                                          int iter, int max_iter, double speed, double T,
                                                          double safety_distance,
                                                          const double *__restrict__ state_quad,
                                                          const double *__restrict__ pos_x,
                                                          const double *__restrict__ ref_path,
                                                          const float *__restrict__ knn_pts,
                                                          const double *__restrict__ knn_d2,
                                                          double *__restrict__ ref_states, int *__restrict__ done,
                                                          const int *__restrict__ flags) {
    const int lane = threadIdx.x;
    // iter < 0: the scene's own pass counter (the solves it has finished this step, flags[1]) -- with an iteration budget on the
    // solve (amk_mpc_set_solve_budget) the scenes of a launch are no longer in the same pass
    if (iter < 0) iter = flags[4 * s + 1];
    // QueryNearest through either path returns K points iff the cloud holds more than K, else none
    // (FrameKDMap.cpp:298,339-345 + kd_tree_two.h:119-124)
    const int cnt = sizes_obs[s] > K ? K : 0;
    bool need = false;
    if (lane < N) need = (cnt == 0) || (sqrt(knn_d2[((size_t)s * N + lane) * K]) <= safety_distance);
    const bool need_replan = __ballot(need) != 0ull;
    if (!need_replan && iter > 0 && flags[4 * s + 0]) {  // :333-335
        if (lane == 0) done[s] = 1;
        return;
    }
    double *P = ref_states + (size_t)s * nref;
    const double *sq = state_quad + ((size_t)s * max_iter + iter) * SD;
    const double *rp = ref_path + (size_t)s * N * SD;
    if (lane < SD) P[lane] = sq[lane];
    for (int e = lane; e < SD * N; e += 64) P[SD + e] = rp[e];
    for (int e = lane; e < 3 * K * N; e += 64) {
        const int j = (e / 3) % K;
        P[SD + SD * N + e] = (j < cnt) ? (double)knn_pts[(size_t)s * N * K * 3 + e] : 10000.0;  // :223-226
    }
    if (lane < SD) {
        const double *last = rp + (N - 1) * SD;
        double v = last[lane];
        if (lane == 0) {
            double dX = speed * T - fmax(0., last[0] - pos_x[s]);
            dX = fmax(0., dX);
            v += dX;
        }
        if (lane == 1) v = 0.;
        P[SD + SD * N + 3 * K * N + lane] = v;
    }
}

// PlanWapionts, then ProcessWaypoints' bookkeeping + GetRefStates, for scene s = blockIdx.x (one wavefront): the
// second half reads what the first one wrote for this scene only (snapped point, its neighbours, isSafety).
// EXACT (obstacle handle in AMK_TIES_NANOFLANN mode) is a template parameter, not a flag: the traversal's stack lives in
// scratch memory, and a kernel that MAY use scratch makes every hardware queue reserve it (with 32 queues in flight the
// default path ran out of resources when the two shared one kernel).
template <bool EXACT, bool GRID>
__global__ __launch_bounds__(kWave) void step_plan_pack_kernel(
    GridPtrs gpt, ExactPtrs eobs, const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z,
    int cap, const int *__restrict__ sizes_obs, const float *__restrict__ pmax_obs, const int *__restrict__ sizes_edge,
    int N, int K, int nref, int iter, int max_iter, double speed, double T, double safety_distance,
    const double *__restrict__ state_quad, const double *__restrict__ pos_x, double *__restrict__ ref_path,
    float *__restrict__ knn_pts, double *__restrict__ knn_d2, const float *__restrict__ edge_pt,
    const double *__restrict__ edge_d2, double *__restrict__ ref_states, int *__restrict__ done,
    int *__restrict__ flags) {
    const int s = blockIdx.x;
    if (done[s]) return;
    plan_scene<EXACT, GRID>(s, gpt, eobs, X, Y, Z, cap, sizes_obs, pmax_obs, sizes_edge, N, K, safety_distance, ref_path, knn_pts,
               knn_d2, edge_pt, edge_d2, flags);
    __threadfence_block();
    __syncthreads();
    pack_scene(s, sizes_obs, N, K, nref, iter, max_iter, speed, T, safety_distance, state_quad, pos_x, ref_path, knn_pts,
               knn_d2, ref_states, done, flags);
}

}  // namespace

// Internal (tools/experiments, bench.py AMK_BENCH_SKIP): leave kernel classes out of amk_step_batch to see what each costs
// with the others in flight -- bit 0 queries, bit 1 plan/pack, bit 2 solves.  Results are garbage while set.
static int g_diag_skip = 0;
extern "C" void amk__diag_skip(int mask) { g_diag_skip = mask; }

extern "C" int amk_step_batch(amk_kd *obstacle, amk_kd *edge, amk_mpc *mpc, const amk_step_params *prm,
                              const double *d_state_quad, const double *d_pos_x, double *d_ref_path, double *d_u,
                              double *d_x0array, int *d_flags, void *stream_) {
    if (!obstacle || !edge || !mpc || !prm || !d_state_quad || !d_pos_x || !d_ref_path || !d_u || !d_flags)
        return AMK_ERR_INVALID_ARG;
    if (obstacle->n_scenes != mpc->S || edge->n_scenes != mpc->S) return AMK_ERR_INVALID_ARG;
    if (prm->mpc_max_iter < 1 || prm->mpc_max_iter > AMK_MAX_OUTER_ITER || mpc->K < 1) return AMK_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int N = mpc->N, K = mpc->K;
    if (!mpc->done.p) {
        const size_t Sa = mpc->S;
        AMK_HIP(mpc->knn_pts.alloc(Sa * N * K * 3));
        AMK_HIP(mpc->knn_d2.alloc(Sa * N * K));
        AMK_HIP(mpc->edge_pt.alloc(Sa * 3));
        AMK_HIP(mpc->edge_d2.alloc(Sa));
        AMK_HIP(mpc->ref_states.alloc(Sa * mpc->nref));
        AMK_HIP(mpc->done.alloc(Sa));
    }
    const int S = mpc->launch_scenes();   // (amk_pipeline: a gang that is not full runs its leading scenes only)
    { TimedLaunch tl(KC_BEGIN, stream);
    hipLaunchKernelGGL(step_begin_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, S, mpc->done.p, d_flags, d_u); }
    int qpw, groups, wpb;
    scan_geometry(N, qpw, groups, wpb);
    const int bps = (groups + wpb - 1) / wpb;
    const int S8 = (S + 7) / 8 * 8;
    const int use_grid = (obstacle->mode == 0 && edge->mode == 0) ? 1 : 0;
    if (!use_grid) {  // the streaming-scan cross-check path reads the index-ordered planes
        int st = amk__kd_ensure_soa(obstacle, stream_);
        if (st == AMK_OK) st = amk__kd_ensure_soa(edge, stream_);
        if (st != AMK_OK) return st;
    }
    const GridPtrs gobs{obstacle->gpt.p, obstacle->cell_start.p, obstacle->gparams.p, obstacle->cap, obstacle->ntiles};
    const GridPtrs gedge{edge->gpt.p, edge->cell_start.p, edge->gparams.p, edge->cap, edge->ntiles};
    // handles in AMK_TIES_NANOFLANN mode (their reference-shaped trees were built by amk_kd_build)
    const int ex_obs = use_grid && obstacle->tie_order && obstacle->ex_valid, ex_edge = use_grid && edge->tie_order && edge->ex_valid;
    const ExactPtrs eobs = ex_obs ? amk_exact_ptrs(obstacle) : ExactPtrs{}, eedge = ex_edge ? amk_exact_ptrs(edge) : ExactPtrs{};
    // Rounds.  Plain schedule (no budget): round r IS pass r of every scene still in the loop, mpc_max_iter rounds.  With an
    // iteration budget B (amk_mpc_set_solve_budget) a solve launch of the first `budget_rounds` rounds ends after B iterations per
    // scene; a scene that is not finished by then pauses (done[s] = 2), sits out the next round's queries and packing, and is
    // resumed INSIDE that round's solve launch beside the fresh solves of the scenes that moved on.  Every scene still makes its
    // passes in order with the reference's data flow (queries at ITS refilled path, its own pass counter for GetCurStateQuad and
    // the early exit), so the results are the plain schedule's bit for bit; what changes is that a launch no longer lasts as
    // long as its slowest scene.  The rounds behind the budgeted ones run without a budget: a scene that enters them in pass p
    // needs mpc_max_iter - p of them, hence mpc_max_iter catch-up rounds (a round nobody needs is three empty launches).
    const int mi = prm->mpc_max_iter;
    const int budget = (g_diag_skip & 4) ? 0 : mpc->solve_budget;
    const int brounds = budget > 0 ? (mpc->budget_rounds > 0 ? mpc->budget_rounds : mi - 1) : 0;
    const int rounds = budget > 0 && brounds > 0 ? brounds + mi : mi;
    const bool per_scene = rounds != mi;
    for (int iter = 0; iter < rounds; ++iter) {
        if (g_diag_skip & 1) {
        } else if (use_grid) {
            TimedLaunch tl(KC_SCAN_OBS, stream);
            hipLaunchKernelGGL(step_knn_grid_kernel, dim3(S8 * ((N + 4) / 4)), dim3(256), 0, stream, gobs, gedge, S,
                               d_ref_path, N, K, mpc->knn_pts.p, mpc->knn_d2.p, mpc->edge_pt.p, mpc->edge_d2.p,
                               mpc->done.p);
            if (ex_obs || ex_edge)
                hipLaunchKernelGGL(step_knn_exact_kernel, dim3((S * (N + 1) + 3) / 4), dim3(256), 0, stream, eobs, eedge, ex_obs,
                                   ex_edge, S, d_ref_path, N, K, mpc->knn_pts.p, mpc->knn_d2.p, mpc->edge_pt.p,
                                   mpc->edge_d2.p, mpc->done.p);
        } else {
        { TimedLaunch tl(KC_SCAN_OBS, stream);
        hipLaunchKernelGGL(step_scan_kernel<5>, dim3(S8 * bps), dim3(wpb * kWave), scan_lds_bytes<5>(wpb), stream,
                           obstacle->x.p, obstacle->y.p, obstacle->z.p, obstacle->cap, obstacle->size.p,
                           obstacle->pmax.p, S, d_ref_path, N, N, K, mpc->knn_pts.p, mpc->knn_d2.p, mpc->done.p); }
        { TimedLaunch tl(KC_SCAN_EDGE, stream);
        hipLaunchKernelGGL(step_scan_kernel<1>, dim3(S8), dim3(kWave), scan_lds_bytes<1>(1), stream, edge->x.p,
                           edge->y.p, edge->z.p, edge->cap, edge->size.p, edge->pmax.p, S, d_ref_path, N, 1, 1,
                           mpc->edge_pt.p, mpc->edge_d2.p, mpc->done.p); }
        }
        if (!(g_diag_skip & 2)) { TimedLaunch tl(KC_PLAN, stream);
        auto plan_kernel = step_plan_pack_kernel<false, true>;
        if (!use_grid) plan_kernel = step_plan_pack_kernel<false, false>;
        else if (ex_obs) plan_kernel = step_plan_pack_kernel<true, true>;
        hipLaunchKernelGGL(plan_kernel, dim3(S), dim3(kWave), 0, stream, gobs, eobs, obstacle->x.p,
                           obstacle->y.p, obstacle->z.p, obstacle->cap, obstacle->size.p, obstacle->pmax.p, edge->size.p, N,
                           K, mpc->nref, per_scene ? -1 : iter, prm->mpc_max_iter, prm->speed, mpc->T, prm->safety_distance, d_state_quad,
                           d_pos_x, d_ref_path, mpc->knn_pts.p, mpc->knn_d2.p, mpc->edge_pt.p, mpc->edge_d2.p,
                           mpc->ref_states.p, mpc->done.p, d_flags); }
        AMK_HIP(hipGetLastError());
        if (g_diag_skip & 4) continue;
        int st = launch_solve(mpc, mpc->ref_states.p, d_u, d_x0array, nullptr, mpc->done.p, d_ref_path, d_flags, stream,
                              per_scene && iter < brounds ? budget : 0, per_scene ? mi : 0);
        if (st != AMK_OK) return st;
    }
    return AMK_OK;
}

extern "C" int amk_step_batch_host(amk_kd *obstacle, amk_kd *edge, amk_mpc *mpc, const amk_step_params *prm,
                                   const double *h_state_quad, const double *h_pos_x, double *h_ref_path, double *h_u,
                                   double *h_x0array, int *h_flags) {
    if (!mpc || !prm || !h_state_quad || !h_pos_x || !h_ref_path || !h_u || !h_flags) return AMK_ERR_INVALID_ARG;
    if (prm->mpc_max_iter < 1 || prm->mpc_max_iter > AMK_MAX_OUTER_ITER) return AMK_ERR_INVALID_ARG;
    const size_t S = mpc->S, N = mpc->N, mi = prm->mpc_max_iter;
    if (mpc->sh_sq.n < S * mi * SD) AMK_HIP(mpc->sh_sq.alloc(S * AMK_MAX_OUTER_ITER * SD));
    if (!mpc->sh_ref.p) {
        AMK_HIP(mpc->sh_posx.alloc(S));
        AMK_HIP(mpc->sh_ref.alloc(S * N * SD));
        AMK_HIP(mpc->sh_u.alloc(S * 4));
        AMK_HIP(mpc->sh_x0.alloc(S * N * 14));
        AMK_HIP(mpc->sh_flags.alloc(S * 4));
    }
    AMK_HIP(hipMemcpy(mpc->sh_sq.p, h_state_quad, sizeof(double) * S * mi * SD, hipMemcpyHostToDevice));
    AMK_HIP(hipMemcpy(mpc->sh_posx.p, h_pos_x, sizeof(double) * S, hipMemcpyHostToDevice));
    AMK_HIP(hipMemcpy(mpc->sh_ref.p, h_ref_path, sizeof(double) * S * N * SD, hipMemcpyHostToDevice));
    int st = amk_step_batch(obstacle, edge, mpc, prm, mpc->sh_sq.p, mpc->sh_posx.p, mpc->sh_ref.p, mpc->sh_u.p,
                            mpc->sh_x0.p, mpc->sh_flags.p, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipDeviceSynchronize());
    AMK_HIP(hipMemcpy(h_ref_path, mpc->sh_ref.p, sizeof(double) * S * N * SD, hipMemcpyDeviceToHost));
    AMK_HIP(hipMemcpy(h_u, mpc->sh_u.p, sizeof(double) * S * 4, hipMemcpyDeviceToHost));
    if (h_x0array) AMK_HIP(hipMemcpy(h_x0array, mpc->sh_x0.p, sizeof(double) * S * N * 14, hipMemcpyDeviceToHost));
    AMK_HIP(hipMemcpy(h_flags, mpc->sh_flags.p, sizeof(int) * S * 4, hipMemcpyDeviceToHost));
    return AMK_OK;
}

#ifdef AMK_KNN_COUNT
extern "C" int amk__knn_counters(unsigned long long *h8, int reset) {   // diagnostics build only: step.hip's copy of grid_knn's counters
    if (hipDeviceSynchronize() != hipSuccess) return AMK_ERR_HIP;
    if (hipMemcpyFromSymbol(h8, HIP_SYMBOL(amk::g_knn_cnt), sizeof(unsigned long long) * 8) != hipSuccess) return AMK_ERR_HIP;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(amk::g_knn_cnt), z, sizeof z) != hipSuccess) return AMK_ERR_HIP; }
    return AMK_OK;
}
#endif
