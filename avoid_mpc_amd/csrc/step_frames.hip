// One control step over a MULTI-FRAME map (current frame + keyframes) for a batch of scenes on gfx950.
//
// amk_step_batch (step.hip) covers the map of BASELINE's synthetic configs, mVecQueryVector = [cur].  With keyframes
// FrameKDMap answers a query differently (AM/src/FrameKDMap.cpp):
//   QueryNearest (:322-376)     fast path -- current frame only -- when that frame's cloud holds >= k points AND the query
//                               projects into the current image (PtIsInFrame, :215-231); otherwise every frame f of
//                               mVecQueryVector is searched with k' = min(k, size_f) (:298) and the union is sorted by
//                               squared distance, first k kept (:371-375).  With KDTreeTwo's count rule (kd_tree_two.h:
//                               119-124: size == n -> no result) a frame contributes k points iff it holds MORE than k.
//   GetNearestDistance (:400-427)  min over the frames (non-empty obstacle cloud) of the 1-NN squared distance, sqrt.
// This file is the TASK branch of Step (AM/src/AvoidanceStateMachine.cpp:322-355) on such a map: obstacles remembered only
// through keyframes (already out of the field of view) stay in needReplan, PlanWapionts and P.
//
// Per outer iteration:
//   step_knn_frames_kernel        raw K-NN of the N reference points + edge 1-NN of point 0 in EVERY frame (grid.y = frame)
//   step_merge_plan_pack_kernel   one workgroup (4 wavefronts) per scene: PtIsInFrame per reference point, GetNearestDistance, PlanWapionts
//                                 (snap to the nearest edge point over the frames, re-query), fast path / merge per
//                                 reference point, padding, needReplan, early exit, GetRefStates
//   mpc_solve_kernel              Solve + refill of the reference path (mpc_solve.hip)
// Frames whose handles are in AMK_TIES_NANOFLANN mode are queried by nanoflann's own traversal of its own tree
// (kd_exact.h): step_knn_frames_exact_kernel overwrites the raw results, the snap re-query follows suit.
#include "kd_exact.h"
#include "mpc_handle.h"

using namespace amk;

namespace {

struct FrameSet {  // kernel argument: where every frame's indices live
    GridPtrs obs[AMK_MAX_FRAMES], edge[AMK_MAX_FRAMES];
    const int *size_obs[AMK_MAX_FRAMES], *size_edge[AMK_MAX_FRAMES];
    int n;
    // Map mode (the keyframe map, kfmap.hip): every frame of every scene lives in ONE pool handle (obs[0] / edge[0] /
    // size_*[0]); frame f of scene s is pool scene fmap[f * S + s], or absent (< 0: this scene's map is shorter) -- an absent
    // frame behaves like an empty cloud, which contributes nothing to any query (FrameKDMap.cpp:298,385-387).  n may then
    // exceed AMK_MAX_FRAMES (the per-frame arrays above are not used beyond [0]).
    const int *fmap;
    int S;
    __device__ __forceinline__ int scene_of(int f, int s) const { return fmap ? fmap[(size_t)f * S + s] : s; }
    __device__ __forceinline__ GridScene obs_scene(int f, int m) const { return (fmap ? obs[0] : obs[f]).scene(m); }
    __device__ __forceinline__ GridScene edge_scene(int f, int m) const { return (fmap ? edge[0] : edge[f]).scene(m); }
    __device__ __forceinline__ int n_obs(int f, int s) const {
        const int m = scene_of(f, s);
        return m < 0 ? 0 : (fmap ? size_obs[0] : size_obs[f])[m];
    }
    __device__ __forceinline__ int n_edge(int f, int s) const {
        const int m = scene_of(f, s);
        return m < 0 ? 0 : (fmap ? size_edge[0] : size_edge[f])[m];
    }
};

struct FrameBufs {  // per-frame raw query results, frame-major
    float *knn_pts;   // [F][S][N][K][3]
    double *knn_d2;   // [F][S][N][K]
    float *edge_pt;   // [F][S][3]
    double *edge_d2;  // [F][S]
};

struct FrameExact {  // lives in device memory (too large for the kernel argument segment next to FrameSet)
    ExactPtrs obs[AMK_MAX_FRAMES], edge[AMK_MAX_FRAMES];
    int use_obs[AMK_MAX_FRAMES], use_edge[AMK_MAX_FRAMES];
};

// PtIsInFrame (FrameKDMap.cpp:215-231): Twc rigid, its inverse is [R' | -R' t]
__device__ __forceinline__ bool pt_in_frame(const double *__restrict__ T, const amk_frame_camera &cam, double px, double py,
                                            double pz) {
#pragma clang fp contract(off)   // two kernels take this decision for the same point (the searches skip what the merge will not read): same bits in both
    const double dx = px - T[3], dy = py - T[7], dz = pz - T[11];
    const double x = T[0] * dx + T[4] * dy + T[8] * dz;
    const double y = T[1] * dx + T[5] * dy + T[9] * dz;
    const double z = T[2] * dx + T[6] * dy + T[10] * dz;
    if (z > cam.depth_max || z < 0) return false;
    const double u = cam.fx * x / z + cam.cx;
    const double v = cam.fy * y / z + cam.cy;
    if (u < 0 || u >= cam.width || v < 0 || v >= cam.height) return false;
    return true;
}

// blockIdx.y = the frame (a list of handles: every frame exists) or a chunk of consecutive frames (a keyframe map with room for
// 101 frames holds ~6 on a flight, and a launch of 101 x S x (N + 4) / 4 blocks of which 94 % return at once is mostly dispatch)
template <bool MAP>
__global__ __launch_bounds__(256) void step_knn_frames_kernel(FrameSet fs, int n_scenes, const double *__restrict__ ref_path,
                                                              int N, int K, FrameBufs fb, const int *__restrict__ done, int fc,
                                                              const double *__restrict__ Twc, amk_frame_camera cam) {
    __shared__ GridWaveLds wl[4];
    // map mode: reference point 0 is searched in EVERY frame (GetNearestDistance reads it, the snap may move it) -- one wavefront
    // walking the ~6 frames of a flight's map in turn was the tail of every launch.  Frames 1 .. fc - 1 of the first chunk get a
    // wavefront each: "queries" N + 1 .. N + fc - 1 are reference point 0 in frame q - N.
    const int nq = N + 1, nv = MAP ? fc - 1 : 0;
    const int bps = (nq + nv + 3) / 4;
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = (j / bps) * 8 + xcd;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */, lane = threadIdx.x & 63;
    int q = (j % bps) * 4 + w;
    if (s >= n_scenes || q >= nq + nv || done[s]) return;
    int f_only = -1;
    if (q >= nq) {
        if (blockIdx.y != 0) return;
        f_only = q - N;
        q = 0;
    }
    const bool is_edge = q == N;
    const double *qp = ref_path + ((size_t)s * N + (is_edge ? 0 : q)) * SD;
    const double qx = qp[0], qy = qp[1], qz = qp[2];
    const int k = is_edge ? 1 : K;
    // Frames behind the current one are searched only for the queries that will read them: QueryNearest answers from the current
    // frame alone when that frame holds >= k points and the query projects into the current image (FrameKDMap.cpp:339-345) -- on a
    // flight nearly every reference point does.  The merge kernel takes the same decision from the same numbers (reference point 0
    // may be moved by the snap in between and GetNearestDistance reads its rows in every frame: it is always searched everywhere).
    bool cur_only = false;
    if (q != 0) {
        const int n0 = is_edge ? fs.n_edge(0, s) : fs.n_obs(0, s);
        cur_only = n0 >= k && (!Twc || pt_in_frame(Twc + (size_t)s * 16, cam, qx, qy, qz));
    }
    const GridPtrs pool = is_edge ? fs.edge[0] : fs.obs[0];   // (map mode: one set of pool pointers stays live over the loop)
    // map mode: chunk c = frames [fc (2^c - 1), fc (2^(c+1) - 1)) -- 8, 16, 32, ... of them: a map on a flight holds ~6 frames of its
    // 101, and every chunk beyond the first is a grid of blocks that find nothing to do
    int f_begin = MAP ? fc * ((1 << blockIdx.y) - 1) : blockIdx.y;
    int f_end = MAP ? min(fs.n, fc * ((2 << blockIdx.y) - 1)) : (int)blockIdx.y + 1;
    if (MAP && blockIdx.y == 0) {
        if (f_only >= 0) { f_begin = f_only; f_end = min(f_end, f_only + 1); }
        else if (q == 0) f_end = min(f_end, 1);   // (its other frames of this chunk: the wavefronts above)
    }
    for (int f = f_begin; f < f_end; ++f) {
        if (cur_only && f > 0) break;
        const int m = MAP ? fs.fmap[(size_t)f * fs.S + s] : s;
        if (m < 0) continue;   // (map mode: this scene's map has no frame f; nobody reads its rows -- n_obs / n_edge are 0)
        double ld;
        int li, lpos;
        const GridScene gs = MAP ? pool.scene(m) : (is_edge ? fs.edge[f] : fs.obs[f]).scene(m);
        grid_knn(gs, qx, qy, qz, k, ld, li, lpos, &wl[w]);
        if (lane < k) {
            const bool ok = li != kNoIndex;
            const float4 rec = gs.pt[lpos];
            if (is_edge) {
                const size_t o = (size_t)f * n_scenes + s;
                fb.edge_d2[o] = ok ? ld : DBL_MAX;
                fb.edge_pt[3 * o + 0] = ok ? rec.x : 0.f;
                fb.edge_pt[3 * o + 1] = ok ? rec.y : 0.f;
                fb.edge_pt[3 * o + 2] = ok ? rec.z : 0.f;
            } else {
                const size_t row = ((size_t)f * n_scenes + s) * N + q;
                fb.knn_d2[row * K + lane] = ok ? ld : DBL_MAX;
                float *o = fb.knn_pts + (row * K + lane) * 3;
                o[0] = ok ? rec.x : 0.f;
                o[1] = ok ? rec.y : 0.f;
                o[2] = ok ? rec.z : 0.f;
            }
        }
    }
}

// one THREAD per (frame, scene, query): the same raw results by the reference's traversal, where the frame has its tree
__global__ __launch_bounds__(256) void step_knn_frames_exact_kernel(const FrameExact *__restrict__ fe, int n_scenes,
                                                                    const double *__restrict__ ref_path, int N, int K,
                                                                    FrameBufs fb, const int *__restrict__ done) {
    __shared__ ExactWaveStack stacks[4];
    const int f = blockIdx.y;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + w;
    const int nq = N + 1;
    if (t >= n_scenes * nq) return;
    const int s = t / nq, q = t - s * nq;
    if (done[s]) return;
    const bool is_edge = q == N;
    if (is_edge ? !fe->use_edge[f] : !fe->use_obs[f]) return;
    const double *qp = ref_path + ((size_t)s * N + (is_edge ? 0 : q)) * SD;
    const ExactTree T = is_edge ? fe->edge[f].scene(s) : fe->obs[f].scene(s);
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
            const size_t o = (size_t)f * n_scenes + s;
            fb.edge_d2[o] = ok ? rd : DBL_MAX;
            fb.edge_pt[3 * o + 0] = px; fb.edge_pt[3 * o + 1] = py; fb.edge_pt[3 * o + 2] = pz;
        } else {
            const size_t row = ((size_t)f * n_scenes + s) * N + q;
            fb.knn_d2[row * K + j] = ok ? rd : DBL_MAX;
            float *o = fb.knn_pts + (row * K + j) * 3;
            o[0] = px; o[1] = py; o[2] = pz;
        }
    }
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

constexpr int kMaxCandPerLane = (AMK_MAX_FRAMES * AMK_MAX_K + 63) / 64;
constexpr int kMaxCandPerLaneMap = 0;   // the keyframe map beyond 64 kMaxCandPerLane candidates (the reference's max_frame_count = 100:
                                        // 101 frames x K): the candidates are re-read every round instead of living in registers

// EXACT (some frame in AMK_TIES_NANOFLANN mode): a template parameter so that the default kernel needs no scratch memory.
// CPL: merge candidates (frame, neighbour) a lane may hold -- F K <= 64 CPL.
template <bool EXACT, int CPL = kMaxCandPerLane>
__global__ __launch_bounds__(4 * kWave) void step_merge_plan_pack_kernel(
    FrameSet fs, const FrameExact *__restrict__ fe, FrameBufs fb, int S, const double *__restrict__ Twc, amk_frame_camera cam, int N, int K, int nref, int iter,
    int max_iter, double speed, double T, double safety_distance, const double *__restrict__ state_quad,
    const double *__restrict__ pos_x, double *__restrict__ ref_path, float *__restrict__ knn_pts,
    double *__restrict__ knn_d2, double *__restrict__ ref_states, int *__restrict__ done, int *__restrict__ flags) {
    // One workgroup per scene: nw = blockDim.x / 64 wavefronts (4; 1 when a frame is in AMK_TIES_NANOFLANN mode).  Wavefront 0
    // decides PlanWapionts; the snapped point's re-queries (one search per frame) and the per-reference-point merges are dealt
    // round-robin to the wavefronts, the rows that take QueryNearest's fast path are copied by all threads at once.
    const int s = blockIdx.x, lane = threadIdx.x & 63, tid = threadIdx.x, nthr = blockDim.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    if (done[s]) return;
    // frames this scene's map holds: every loop below runs over them only (an absent frame contributes nothing to any query, and the
    // candidate ids f K + j of the others do not move).  A map with room for 101 frames holds ~6 on a flight; each pass over
    // absent frames is a chain of dependent loads (fmap, then the size) per frame.
    int F = fs.n;
    if (fs.fmap) {
        int hi = 0;
        for (int f0 = 0; f0 < fs.n; f0 += 64) {
            const int f = f0 + lane;
            const unsigned long long b = __ballot(f < fs.n && fs.fmap[(size_t)f * fs.S + s] >= 0);
            if (b) hi = f0 + 64 - __clzll((long long)b);
        }
        F = hi;
    }
    __shared__ GridWaveLds wl[4];
    __shared__ int cntq[AMK_MAX_HORIZON];
    __shared__ int sh_safety, sh_snap;
    __shared__ double sh_e[3];
    double *rp = ref_path + (size_t)s * N * SD;
    const double *Ts = Twc ? Twc + (size_t)s * 16 : nullptr;
    auto in_frame = [&](double x, double y, double z) { return Ts ? pt_in_frame(Ts, cam, x, y, z) : true; };
    const int n_obs0 = fs.n_obs(0, s);
    // ---- PlanWapionts (:259-281) for reference point 0
    if (w == 0) {
        const double p0x = rp[0], p0y = rp[1], p0z = rp[2];
        // GetNearestDistance: 1-NN per frame exists iff the frame holds more than one point (lane = frame)
        unsigned long long d2n_key = ~0ull;
        for (int f0 = 0; f0 < F; f0 += 64) {
            const int f = f0 + lane;
            if (f < F && fs.n_obs(f, s) > 1) {
                const double d = fb.knn_d2[(((size_t)f * S + s) * N) * K];
                // fmin semantics: a NaN distance is ignored; d >= 0, so the bit pattern orders like the value
                if (d == d) { const unsigned long long k64 = (unsigned long long)__double_as_longlong(d); d2n_key = k64 < d2n_key ? k64 : d2n_key; }
            }
        }
        d2n_key = wave_min_u64(d2n_key);
        const double d2n = d2n_key == ~0ull ? DBL_MAX : __longlong_as_double((long long)d2n_key);
        int is_safety = 1, snap = 0;
        if (!(sqrt(d2n) > safety_distance)) {
            // QueryNearest(p1, 1, ..., queryEdge = true): fast path iff the current edge cloud holds >= 1 point and p1 is in frame
            int bf = -1;
            if (fs.n_edge(0, s) >= 1 && in_frame(p0x, p0y, p0z)) {
                if (fs.n_edge(0, s) > 1 && fb.edge_d2[s] < DBL_MAX) bf = 0;
            } else {
                // k' = min(1, size_f): a result iff size_f > 1; ties keep the earlier frame (lane = frame; strict < in frame order)
                unsigned long long bk = ~0ull;
                int mf = 0x7fffffff;
                for (int f0 = 0; f0 < F; f0 += 64) {
                    const int f = f0 + lane;
                    if (f < F && fs.n_edge(f, s) > 1) {
                        const double d = fb.edge_d2[(size_t)f * S + s];
                        if (d < DBL_MAX) {
                            const unsigned long long k64 = (unsigned long long)__double_as_longlong(d);
                            if (k64 < bk) { bk = k64; mf = f; }
                        }
                    }
                }
                const unsigned long long wb = wave_min_u64(bk);
                if (wb != ~0ull) {
                    int win = bk == wb ? mf : 0x7fffffff;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) win = min(win, __shfl_xor(win, off));
                    bf = win;
                }
            }
            if (bf < 0) {
                is_safety = 0;
            } else {
                snap = 1;
                const float *ep = fb.edge_pt + 3 * ((size_t)bf * S + s);
                if (lane == 0) { sh_e[0] = (double)ep[0]; sh_e[1] = (double)ep[1]; sh_e[2] = (double)ep[2]; }
            }
        }
        if (lane == 0) { sh_safety = is_safety; sh_snap = snap; flags[4 * s + 0] = is_safety; }
    }
    __syncthreads();
    const int is_safety = sh_safety;
    if (sh_snap) {
        const double ex = sh_e[0], ey = sh_e[1], ez = sh_e[2];
        for (int f = w; f < F; f += nw) {  // the snapped point is what ProcessWaypoints queries next (:210-215)
            double gld;
            int gli, glpos;
            const int mf = fs.scene_of(f, s);
            if (mf >= 0) {   // (wave-uniform)
                const GridScene gs = fs.obs_scene(f, mf);
                grid_knn(gs, ex, ey, ez, K, gld, gli, glpos, &wl[w]);
                if (lane < K) {
                    const bool ok = gli != kNoIndex;
                    const float4 rec = gs.pt[glpos];
                    const size_t row = ((size_t)f * S + s) * N;
                    fb.knn_d2[row * K + lane] = ok ? gld : DBL_MAX;
                    float *o = fb.knn_pts + (row * K + lane) * 3;
                    o[0] = ok ? rec.x : 0.f; o[1] = ok ? rec.y : 0.f; o[2] = ok ? rec.z : 0.f;
                }
            }
            if constexpr (EXACT) {   // (nw == 1: the barriers below are this wavefront's own)
                __syncthreads();
                if (mf >= 0 && fe->use_obs[f]) {  // AMK_TIES_NANOFLANN frame: the re-query by the reference's traversal (lane 0)
                    __shared__ double xr[AMK_MAX_K];
                    __shared__ int xi[AMK_MAX_K], xgot;
                    __shared__ ExactStackStorage xstack;  // LDS, not scratch: one lane walks the tree
                    const ExactTree T = fe->obs[f].scene(s);
                    if (lane == 0) xgot = exact_knn_thread(T, ex, ey, ez, K, xr, xi, xstack.view());
                    __syncthreads();
                    if (xgot >= 0 && lane < K) {
                        const bool ok = lane < xgot;
                        const size_t row = ((size_t)f * S + s) * N;
                        fb.knn_d2[row * K + lane] = ok ? xr[lane] : DBL_MAX;
                        float *o = fb.knn_pts + (row * K + lane) * 3;
                        o[0] = ok ? T.x[xi[lane]] : 0.f; o[1] = ok ? T.y[xi[lane]] : 0.f; o[2] = ok ? T.z[xi[lane]] : 0.f;
                    }
                    __syncthreads();
                }
            }
        }
        if (tid == 0) { rp[0] = ex; rp[1] = ey; rp[2] = ez; }
    }
    __threadfence_block();
    __syncthreads();
    // ---- ProcessWaypoints' queries (:204-215): fast path or merge over the frames, per reference point
    // QueryNearestWithCurFrame (:254-275, 339-345) for the reference points the current image sees (lane = reference point)
    bool inf = false;
    if (lane < N) inf = n_obs0 >= K && in_frame(rp[lane * SD], rp[lane * SD + 1], rp[lane * SD + 2]);
    const unsigned long long fast = __ballot(inf);
    {
        const int cnt_fast = n_obs0 > K ? K : 0;      // kd_tree_two.h:119-124
        const size_t base = (size_t)s * N * K;        // frame 0's rows of this scene = the output rows' layout
        for (int e = tid; e < N * K; e += nthr) {
            const int i = e / K;
            if ((fast >> i) & 1ull) {
                knn_d2[base + e] = fb.knn_d2[base + e];
                for (int c = 0; c < 3; ++c) knn_pts[(base + e) * 3 + c] = fb.knn_pts[(base + e) * 3 + c];
            }
        }
        if (w == 0 && inf) cntq[lane] = cnt_fast;
    }
    for (int i = w; i < N; i += nw) {
        if ((fast >> i) & 1ull) continue;
        const size_t orow = ((size_t)s * N + i) * K;
        // QueryNearestThreadWorker over mVecQueryVector (:276-321) + sort (:371): candidate c = f * K + j
        const int ncand = F * K;
        int cnt = 0;
        if constexpr (CPL > 0) {
        unsigned long long key[CPL];
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const int c = lane + 64 * r;
            key[r] = ~0ull;
            if (c < ncand) {
                const int f = c / K, jj = c - f * K;
                if (fs.n_obs(f, s) > K) {  // k' = min(K, size_f) results exist iff size_f > k'
                    const double d = fb.knn_d2[(((size_t)f * S + s) * N + i) * K + jj];
                    if (d < DBL_MAX) key[r] = (unsigned long long)__double_as_longlong(d);  // d >= 0: order-preserving
                }
            }
        }
        for (int m = 0; m < K; ++m) {  // K rounds of "smallest remaining (distance, candidate id)"
            unsigned long long loc = ~0ull;
#pragma unroll
            for (int r = 0; r < CPL; ++r) loc = key[r] < loc ? key[r] : loc;
            const unsigned long long best = wave_min_u64(loc);
            if (best == ~0ull) break;
            int myc = 0x7fffffff;  // lowest candidate id holding `best`
#pragma unroll
            for (int r = CPL - 1; r >= 0; --r)
                if (key[r] == best) myc = lane + 64 * r;
            int win = myc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) win = min(win, __shfl_xor(win, off));
            if (myc == win) {
                const int f = win / K, jj = win - f * K;
                const size_t irow = (((size_t)f * S + s) * N + i) * K + jj;
                knn_d2[orow + m] = __longlong_as_double((long long)best);
                for (int c = 0; c < 3; ++c) knn_pts[(orow + m) * 3 + c] = fb.knn_pts[irow * 3 + c];
#pragma unroll
                for (int r = 0; r < CPL; ++r)
                    if (lane + 64 * r == win) key[r] = ~0ull;
            }
            ++cnt;
        }
        } else {
        // wide map: the same K rounds, the candidates re-read from the raw rows every round (L2-resident: F K doubles per
        // reference point), a lane's taken candidates remembered as bits (candidate lane + 64 r = bit r; F K <= 64 x 128)
        unsigned long long taken0 = 0ull, taken1 = 0ull;
        for (int m = 0; m < K; ++m) {
            unsigned long long loc = ~0ull;
            int myc = 0x7fffffff;
            for (int r = 0; lane + 64 * r < ncand; ++r) {
                if ((r < 64 ? taken0 >> r : taken1 >> (r - 64)) & 1ull) continue;
                const int c = lane + 64 * r;
                const int f = c / K, jj = c - f * K;
                if (fs.n_obs(f, s) > K) {
                    const double d = fb.knn_d2[(((size_t)f * S + s) * N + i) * K + jj];
                    if (d < DBL_MAX) {
                        const unsigned long long k64 = (unsigned long long)__double_as_longlong(d);
                        if (k64 < loc) { loc = k64; myc = c; }   // (ascending r: the lowest candidate id among equal keys of this lane)
                    }
                }
            }
            const unsigned long long best = wave_min_u64(loc);
            if (best == ~0ull) break;
            int win = loc == best ? myc : 0x7fffffff;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) win = min(win, __shfl_xor(win, off));
            if (loc == best && myc == win) {
                const int f = win / K, jj = win - f * K;
                const size_t irow = (((size_t)f * S + s) * N + i) * K + jj;
                knn_d2[orow + m] = __longlong_as_double((long long)best);
                for (int c = 0; c < 3; ++c) knn_pts[(orow + m) * 3 + c] = fb.knn_pts[irow * 3 + c];
                const int r = win >> 6;
                if (r < 64) taken0 |= 1ull << r; else taken1 |= 1ull << (r - 64);
            }
            ++cnt;
        }
        }
        if (lane == 0) cntq[i] = cnt;
    }
    __threadfence_block();
    __syncthreads();
    // ---- padding, needReplan (:216-231), early exit (:333-335), GetRefStates (:236-257)
    bool need = false;
    if (lane < N) need = (cntq[lane] == 0) || (sqrt(knn_d2[((size_t)s * N + lane) * K]) <= safety_distance);
    const bool need_replan = __ballot(need) != 0ull;   // (every wavefront evaluates the same rows: one decision per workgroup)
    if (!need_replan && iter > 0 && is_safety) {
        if (tid == 0) done[s] = 1;
        return;
    }
    double *P = ref_states + (size_t)s * nref;
    const double *sq = state_quad + ((size_t)s * max_iter + iter) * SD;
    if (tid < SD) P[tid] = sq[tid];
    for (int e = tid; e < SD * N; e += nthr) P[SD + e] = rp[e];
    for (int e = tid; e < 3 * K * N; e += nthr) {
        const int i = e / (3 * K), jj = (e / 3) % K;
        P[SD + SD * N + e] = (jj < cntq[i]) ? (double)knn_pts[(size_t)s * N * K * 3 + e] : 10000.0;
    }
    if (tid < SD) {
        const double *last = rp + (N - 1) * SD;
        double v = last[tid];
        if (tid == 0) {
            double dX = speed * T - fmax(0., last[0] - pos_x[s]);
            dX = fmax(0., dX);
            v += dX;
        }
        if (tid == 1) v = 0.;
        P[SD + SD * N + 3 * K * N + tid] = v;
    }
}

__global__ void step_frames_begin_kernel(int S, int *__restrict__ done, int *__restrict__ flags, double *__restrict__ u) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    done[s] = 0;
    flags[4 * s + 0] = 1;
    flags[4 * s + 1] = 0;
    flags[4 * s + 2] = -1;
    flags[4 * s + 3] = 0;
    u[4 * s + 0] = u[4 * s + 1] = u[4 * s + 2] = u[4 * s + 3] = 0.0;
}

}  // namespace

// tests only (tests/test_kfmap_gpu.py): take the wide-map merge (candidates re-read every round) whatever the map's size
static int g_force_wide = 0;
extern "C" void amk__frames_force_wide(int on) { g_force_wide = on; }

// the rounds of the step over a frame set (plain handles or the keyframe map's pool), shared by the two entry points below
static int run_frames(const FrameSet &fs, amk_kd *const *obstacle, amk_kd *const *edge, const double *d_Twc,
                      const amk_frame_camera *cam, amk_mpc *mpc, const amk_step_params *prm, const double *d_state_quad,
                      const double *d_pos_x, double *d_ref_path, double *d_u, double *d_x0array, int *d_flags, hipStream_t stream) {
    const int N = mpc->N, K = mpc->K, F = fs.n;
    {
        const size_t S = mpc->S;
        if (!mpc->done.p) {
            AMK_HIP(mpc->knn_pts.alloc(S * N * K * 3));
            AMK_HIP(mpc->knn_d2.alloc(S * N * K));
            AMK_HIP(mpc->edge_pt.alloc(S * 3));
            AMK_HIP(mpc->edge_d2.alloc(S));
            AMK_HIP(mpc->ref_states.alloc(S * mpc->nref));
            AMK_HIP(mpc->done.alloc(S));
        }
    }
    const int Sall = mpc->S;
    if (mpc->mf_frames < F) {   // sized ONCE for the largest map of its kind (AMK_MAX_FRAMES handles; a keyframe map: its own
        // 1 + max_frame_count): a map that gains a keyframe between two calls must not free a buffer an earlier call's kernels
        // may still be reading
        const size_t FM = F > AMK_MAX_FRAMES ? F : AMK_MAX_FRAMES;
        if (mpc->mf_frames > 0) AMK_HIP(hipDeviceSynchronize());   // (growing: an explicit change of the map's capacity)
        AMK_HIP(mpc->mf_knn_pts.alloc(FM * Sall * N * K * 3));
        AMK_HIP(mpc->mf_knn_d2.alloc(FM * Sall * N * K));
        AMK_HIP(mpc->mf_edge_pt.alloc(FM * Sall * 3));
        AMK_HIP(mpc->mf_edge_d2.alloc(FM * Sall));
        mpc->mf_frames = (int)FM;
    }
    const int S = mpc->launch_scenes();   // (amk_pipeline: a gang that is not full runs its leading scenes only; the per-frame
                                          // rows of this call are laid out with this S, the frame map keeps the handle's: fs.S)
    const FrameBufs fb{mpc->mf_knn_pts.p, mpc->mf_knn_d2.p, mpc->mf_edge_pt.p, mpc->mf_edge_d2.p};
    // frames in AMK_TIES_NANOFLANN mode (their reference-shaped trees were built by amk_kd_build / amk_kd_push_keyframe)
    bool any_exact = false;
    FrameExact *fe_dev = nullptr;
    if (!fs.fmap) {
        std::vector<char> cur(sizeof(FrameExact), 0);
        FrameExact *h = reinterpret_cast<FrameExact *>(cur.data());
        for (int f = 0; f < F; ++f) {
            h->use_obs[f] = obstacle[f]->tie_order && obstacle[f]->ex_valid;
            h->use_edge[f] = edge[f]->tie_order && edge[f]->ex_valid;
            // bytewise copies of zero-filled structs: the table is compared bytewise below, padding included
            if (h->use_obs[f]) { const ExactPtrs t = amk_exact_ptrs(obstacle[f]); std::memcpy(&h->obs[f], &t, sizeof t); }
            if (h->use_edge[f]) { const ExactPtrs t = amk_exact_ptrs(edge[f]); std::memcpy(&h->edge[f], &t, sizeof t); }
            any_exact |= h->use_obs[f] || h->use_edge[f];
        }
        if (any_exact) {
            // The table lives in device memory (too large for the kernel-argument segment).  It is uploaded only when the
            // frame list or a frame's mode changed -- a host event (AddVertex / keyframe push) -- and then synchronously after
            // a device-wide wait: earlier calls on other streams may still be reading the old table.  The steady state issues
            // no copy at all, so a step over an unchanged exact-mode map is graph-capturable like the other step paths; the
            // call that follows a change of the map is not.
            if (!mpc->mf_exact.p) AMK_HIP(mpc->mf_exact.alloc(sizeof(FrameExact)));
            if (mpc->mf_exact_host != cur) {
                AMK_HIP(hipDeviceSynchronize());
                AMK_HIP(hipMemcpy(mpc->mf_exact.p, h, sizeof(FrameExact), hipMemcpyHostToDevice));
                mpc->mf_exact_host = cur;
            }
            fe_dev = reinterpret_cast<FrameExact *>(mpc->mf_exact.p);
        }
    }
    amk_frame_camera c{};
    if (cam) c = *cam;
    // the merge keeps ceil(F K / 64) candidates per lane in registers: instantiated for 1 / 2 / 4 / 16 (a 4-frame map with K = 8 needs
    // ONE; the 16-wide instantiation carries 32 key registers and 16 x unrolled selection loops: 204 VGPRs against 70-odd)
    const int need_cpl = (F * K + 63) / 64;
    // a keyframe map beyond 4 candidates per lane takes the wide merge: 61 VGPRs (it runs beside the solves' waves; the 16-wide
    // instantiation cannot), and with the loops bounded by the frames a scene actually holds its re-reads are few
    const bool wide = F * K > 64 * kMaxCandPerLane || (fs.fmap && need_cpl > 4) || (g_force_wide && !any_exact);
    auto merge_kernel = step_merge_plan_pack_kernel<false, kMaxCandPerLane>;
    if (wide) merge_kernel = step_merge_plan_pack_kernel<false, kMaxCandPerLaneMap>;
    else if (any_exact) merge_kernel = step_merge_plan_pack_kernel<true>;
    else if (need_cpl <= 1) merge_kernel = step_merge_plan_pack_kernel<false, 1>;
    else if (need_cpl <= 2) merge_kernel = step_merge_plan_pack_kernel<false, 2>;
    else if (need_cpl <= 4) merge_kernel = step_merge_plan_pack_kernel<false, 4>;
    hipLaunchKernelGGL(step_frames_begin_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, S, mpc->done.p, d_flags, d_u);
    const int S8 = (S + 7) / 8 * 8;
    const int fc = 8;   // map mode: frames of the first chunk of search blocks (the chunks double: 8, 16, 32, ...)
    int n_chunks = 1;
    while (fc * ((1 << n_chunks) - 1) < F) ++n_chunks;
    for (int iter = 0; iter < prm->mpc_max_iter; ++iter) {
        if (fs.fmap)
            hipLaunchKernelGGL(step_knn_frames_kernel<true>, dim3(S8 * ((N + 1 + (fc - 1) + 3) / 4), n_chunks), dim3(256), 0, stream, fs, S,
                               d_ref_path, N, K, fb, mpc->done.p, fc, d_Twc, c);
        else
            hipLaunchKernelGGL(step_knn_frames_kernel<false>, dim3(S8 * ((N + 4) / 4), F), dim3(256), 0, stream, fs, S, d_ref_path,
                               N, K, fb, mpc->done.p, 1, d_Twc, c);
        if (any_exact)
            hipLaunchKernelGGL(step_knn_frames_exact_kernel, dim3((S * (N + 1) + 3) / 4, F), dim3(256), 0, stream, fe_dev, S,
                               d_ref_path, N, K, fb, mpc->done.p);
        hipLaunchKernelGGL(merge_kernel, dim3(S), dim3(any_exact ? kWave : 4 * kWave), 0, stream, fs, fe_dev, fb, S, d_Twc, c, N, K, mpc->nref,
                           iter, prm->mpc_max_iter, prm->speed, mpc->T, prm->safety_distance, d_state_quad, d_pos_x,
                           d_ref_path, mpc->knn_pts.p, mpc->knn_d2.p, mpc->ref_states.p, mpc->done.p, d_flags);
        AMK_HIP(hipGetLastError());
        int st = launch_solve(mpc, mpc->ref_states.p, d_u, d_x0array, nullptr, mpc->done.p, d_ref_path, d_flags, stream);
        if (st != AMK_OK) return st;
    }
    return AMK_OK;
}

extern "C" int amk_step_batch_frames(amk_kd *const *obstacle, amk_kd *const *edge, int n_frames, const double *d_Twc,
                                     const amk_frame_camera *cam, amk_mpc *mpc, const amk_step_params *prm,
                                     const double *d_state_quad, const double *d_pos_x, double *d_ref_path, double *d_u,
                                     double *d_x0array, int *d_flags, void *stream_) {
    if (!obstacle || !edge || !mpc || !prm || !d_state_quad || !d_pos_x || !d_ref_path || !d_u || !d_flags || n_frames < 1)
        return AMK_ERR_INVALID_ARG;
    if (d_Twc && !cam) return AMK_ERR_INVALID_ARG;
    if (n_frames > AMK_MAX_FRAMES) return AMK_ERR_UNSUPPORTED;
    if (prm->mpc_max_iter < 1 || prm->mpc_max_iter > AMK_MAX_OUTER_ITER || mpc->K < 1) return AMK_ERR_INVALID_ARG;
    const int S = mpc->S, F = n_frames;
    FrameSet fs{};
    fs.n = F;
    fs.fmap = nullptr;
    fs.S = S;
    for (int f = 0; f < F; ++f) {
        if (!obstacle[f] || !edge[f] || obstacle[f]->n_scenes != S || edge[f]->n_scenes != S) return AMK_ERR_INVALID_ARG;
        if (obstacle[f]->mode != 0 || edge[f]->mode != 0) return AMK_ERR_UNSUPPORTED;  // bucketed indices only
        fs.obs[f] = GridPtrs{obstacle[f]->gpt.p, obstacle[f]->cell_start.p, obstacle[f]->gparams.p, obstacle[f]->cap, obstacle[f]->ntiles};
        fs.edge[f] = GridPtrs{edge[f]->gpt.p, edge[f]->cell_start.p, edge[f]->gparams.p, edge[f]->cap, edge[f]->ntiles};
        fs.size_obs[f] = obstacle[f]->size.p;
        fs.size_edge[f] = edge[f]->size.p;
    }
    return run_frames(fs, obstacle, edge, d_Twc, cam, mpc, prm, d_state_quad, d_pos_x, d_ref_path, d_u, d_x0array, d_flags,
                      (hipStream_t)stream_);
}

namespace amk {
// The same step over a keyframe map (kfmap.hip): every frame of every scene in the two pool handles, frame f of scene s = pool
// scene d_fmap[f * S + s] (< 0: absent); n_frames = 1 + max_frame_count of the map, <= AMK_MAX_MAP_FRAMES.
int step_batch_map(amk_kd *obs_pool, amk_kd *edge_pool, int n_frames, const int *d_fmap, const double *d_Twc,
                   const amk_frame_camera *cam, amk_mpc *mpc, const amk_step_params *prm, const double *d_state_quad,
                   const double *d_pos_x, double *d_ref_path, double *d_u, double *d_x0array, int *d_flags, hipStream_t stream) {
    if (!obs_pool || !edge_pool || !d_fmap || !mpc || !prm || !d_state_quad || !d_pos_x || !d_ref_path || !d_u || !d_flags || n_frames < 1)
        return AMK_ERR_INVALID_ARG;
    if (d_Twc && !cam) return AMK_ERR_INVALID_ARG;
    if (n_frames > AMK_MAX_MAP_FRAMES || n_frames * mpc->K > 64 * 128) return AMK_ERR_UNSUPPORTED;
    if (prm->mpc_max_iter < 1 || prm->mpc_max_iter > AMK_MAX_OUTER_ITER || mpc->K < 1) return AMK_ERR_INVALID_ARG;
    FrameSet fs{};
    fs.n = n_frames;
    fs.fmap = d_fmap;
    fs.S = mpc->S;
    fs.obs[0] = GridPtrs{obs_pool->gpt.p, obs_pool->cell_start.p, obs_pool->gparams.p, obs_pool->cap, obs_pool->ntiles};
    fs.edge[0] = GridPtrs{edge_pool->gpt.p, edge_pool->cell_start.p, edge_pool->gparams.p, edge_pool->cap, edge_pool->ntiles};
    fs.size_obs[0] = obs_pool->size.p;
    fs.size_edge[0] = edge_pool->size.p;
    return run_frames(fs, nullptr, nullptr, d_Twc, cam, mpc, prm, d_state_quad, d_pos_x, d_ref_path, d_u, d_x0array, d_flags, stream);
}
}  // namespace amk
