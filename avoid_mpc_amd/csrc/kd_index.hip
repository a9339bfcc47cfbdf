// KD index of the Avoid-MPC hot path for gfx950 (MI355X): batch of KDTreeTwo<double> objects.
//
// Replaces (AM = roswrapper/ros/src/avoid_mpc in the reference tree):
//   amk_kd_build   <- KDTreeTwo::InitializeNew      AM/include/kd_tree_two.h:76-78,88-106
//                     (+ nanoflann buildIndex       AM/include/nanoflann_two.hpp:1518-1541)
//   amk_kd_search  <- KDTreeTwo::SearchForNearest   AM/include/kd_tree_two.h:108-133
//                     (+ nanoflann findNeighbors    AM/include/nanoflann_two.hpp:1563-1586)
//
// Design: only the RESULTS of a search must equal the reference's, the
// tree shape is free.  build = one launch, one 512-thread block per scene, reading the caller's AoS cloud directly:
// sampled bounding box, histogram pass (which also applies the NaN-x filter by ballot/popcount), scan, scatter of one
// 16-byte record (x, y, z, cloud index) per point into a bucketed grid (kd_grid.h).  search = one wavefront per
// (scene, query) walking Chebyshev rings of grid cells with nanoflann's branch-and-bound stop rule.  The streaming scan
// of index-ordered SoA planes (kd_device.h; the first correct path) is kept as a cross-check (amk__kd_set_mode) and
// makes its planes on demand.
#include "kd_exact.h"

#include <cstring>

namespace amk {
thread_local int g_last_hip_error = 0;

struct Timing {
    int mode = 0;  // 0 off, 1 KC_SOLVE and KC_GRID (the dominant kernel and the HBM-heavy one), 2 every kernel class
    struct Rec { int kclass; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
};
Timing &timing() { static Timing t; return t; }

TimedLaunch::TimedLaunch(int kclass, hipStream_t s) : slot(-1), stream(s) {
    Timing &t = timing();
    if (t.mode == 0 || (t.mode == 1 && kclass != KC_SOLVE && kclass != KC_GRID)) return;
    Timing::Rec r{kclass, t.get(), t.get()};
    (void)hipEventRecord(r.a, stream);
    slot = (int)t.recs.size();
    t.recs.push_back(r);
}
TimedLaunch::~TimedLaunch() {
    if (slot >= 0) (void)hipEventRecord(timing().recs[slot].b, stream);
}
}  // namespace amk

extern "C" void amk__timing_enable(int mode) { amk::timing().mode = mode; }
// Waits for the recorded events; adds elapsed milliseconds / launch counts per kernel class; resets.
extern "C" int amk__timing_collect(double *ms, int *counts) {
    amk::Timing &t = amk::timing();
    for (int i = 0; i < amk::KC_COUNT; ++i) { ms[i] = 0.0; counts[i] = 0; }
    for (auto &r : t.recs) {
        float f = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&f, r.a, r.b) != hipSuccess) return AMK_ERR_HIP;
        ms[r.kclass] += f;
        counts[r.kclass] += 1;
        t.pool.push_back(r.a);
        t.pool.push_back(r.b);
    }
    t.recs.clear();
    return AMK_OK;
}

// Like amk__timing_collect, but keeps every launch: class, start and end in milliseconds after the first recorded launch
// (tools/experiments/burst_timeline.py: who runs when, without a tracer slowing the host down).  Returns the number of launches.
extern "C" int amk__timing_timeline(int max_recs, int *kclass, double *start_ms, double *end_ms) {
    amk::Timing &t = amk::timing();
    int n = 0;
    for (auto &r : t.recs) {
        float fa = 0.f, fb = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&fa, t.recs[0].a, r.a) != hipSuccess ||
            hipEventElapsedTime(&fb, t.recs[0].a, r.b) != hipSuccess)
            return AMK_ERR_HIP;
        if (n < max_recs) { kclass[n] = r.kclass; start_ms[n] = fa; end_ms[n] = fb; ++n; }
    }
    for (auto &r : t.recs) { t.pool.push_back(r.a); t.pool.push_back(r.b); }
    t.recs.clear();
    return n;
}

using amk::kWave;

// ------------------------------------------------------------------------------------------------
// build: order-preserving compaction of the points whose x is not NaN (kd_tree_two.h:96-101)
// ------------------------------------------------------------------------------------------------
// 512, not 1024: a 1024-thread workgroup needs four wave slots on every SIMD of one CU at the same moment, and the
// dispatcher holds everything behind it until a CU qualifies -- measured: 1024-thread builds do not overlap with
// the solves (or the searches) of other streams AT ALL (time = sum), 512/256-thread builds overlap almost fully
// (solves + builds on separate streams: 32 ms vs 27 ms for the solves alone, 48 ms as a sum).  Alone on the chip
// the 1024-thread version is ~25 % faster (16 instead of 8 waves per CU hide the load latency better).
#ifndef AMK_BUILD_THREADS
#define AMK_BUILD_THREADS 512
#endif
constexpr int kCompactThreads = AMK_BUILD_THREADS;

// Pre-pass of the index build: bounding box of a SAMPLE of the caller's cloud (of a large cloud: runs of 64 consecutive
// points, one run in 16; finite points whose x is not NaN).  The grid geometry only needs a box that holds most points: a
// point outside is clamped into a boundary cell (kd_grid.h), so 1/16 of the cloud is read here instead of all of it.
__device__ __forceinline__ void sample_bbox_scene(int s, const float *__restrict__ src, int point_stride, int n,
                                                  float *__restrict__ bbox_out) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NW = kCompactThreads / kWave;
    __shared__ float wave_bb[6][NW];
#ifndef AMK_BBOX_STEP
#define AMK_BBOX_STEP 16
#endif
    const int step = n >= 16384 ? AMK_BBOX_STEP : 1;
    float bmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    // the sample: runs of 64 consecutive points, one run in `step` (lines of the cloud are fetched whole, so a point in 16
    // cost 2/3 of a full pass in traffic; a run in 16 costs 1/16)
    for (int t = tid;; t += kCompactThreads) {
        const long long i = (long long)(t >> 6) * (64 * step) + (t & 63);
        if (i >= n) break;
        const float *p = src + (size_t)i * point_stride;
        const float px = p[0], py = p[1], pz = p[2];
        if (amk::finite3(px, py, pz)) {
            bmn[0] = fminf(bmn[0], px); bmx[0] = fmaxf(bmx[0], px);
            bmn[1] = fminf(bmn[1], py); bmx[1] = fmaxf(bmx[1], py);
            bmn[2] = fminf(bmn[2], pz); bmx[2] = fmaxf(bmx[2], pz);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bmn[a] = fminf(bmn[a], __shfl_xor(bmn[a], off));
            bmx[a] = fmaxf(bmx[a], __shfl_xor(bmx[a], off));
        }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { wave_bb[a][w] = bmn[a]; wave_bb[3 + a][w] = bmx[a]; }
    __syncthreads();
    if (tid < 6) {
        float v = wave_bb[tid][0];
        for (int j = 1; j < NW; ++j) v = tid < 3 ? fminf(v, wave_bb[tid][j]) : fmaxf(v, wave_bb[tid][j]);
        bbox_out[6 * s + tid] = v;
    }
    __threadfence_block();
    __syncthreads();  // read back by the index build below
}

// InitializeNew for scene s = blockIdx.x in one launch: the sampled bounding box, then the bucketed index straight
// from the caller's cloud in ONE pass over it (grid_build_tiles_scene: the cloud in tiles of 4096 points, each loaded once,
// its records written into the tile's own window).  The index-ordered SoA planes are NOT written: whoever needs them
// (scan-mode searches, amk_kd_points_host, the keyframe sweep) gets them from the records on demand (ensure_soa).
static_assert(kCompactThreads == amk::kGridBuildThreads, "one block shape for both halves of the build");
struct BuildArgs {   // one tree's InitializeNew
    const float *xyz;
    int point_stride;
    long long scene_stride;
    const int *counts;
    int max_points, cap;
    int *size_out;
    float *pmax_out, *bbox_out;
    float4 *GP;
    int *cell_start;
    int ntiles;
    double *gparams;
    const int *keep_if_zero;   // [S] or null: scene s keeps its previous index when keep_if_zero[s] == 0 (FrameKDMap::AddVertex
                               // returns before BOTH InitializeNew calls when the frame's obstacle cloud is empty, FrameKDMap.cpp:39-41)
    float *soa_x, *soa_y, *soa_z;   // or null: the index-ordered planes of the handle's first scene of this entry, written by the build
    const int *out_scene;      // [S] or null: input scene s is built into scene out_scene[s] of the handle (< 0: not built) -- the
                               // keyframe map's pool, where every scene's current frame lives in a physical slot of its own (kfmap.hip)
};
// grid = (scenes, trees): blockIdx.y selects the tree.  FrameKDMap::AddVertex builds TWO trees per depth frame (obstacle +
// edge cloud, FrameKDMap.cpp:44-47): amk_kd_build_pair issues them as one launch, so the small edge build (24 us alone,
// mostly latency) runs in the shadow of the obstacle build instead of behind it.
constexpr int kBuildMaxEntries = 2 * AMK_PIPELINE_MAX_GANG;   // (tree, frame) pairs of one launch
struct BuildArgs2 { BuildArgs t[kBuildMaxEntries]; };
__global__ __launch_bounds__(kCompactThreads, 4) void kd_build_kernel(const BuildArgs2 args) {   // 4 waves per SIMD = two blocks per CU: <= 128 VGPRs
    const BuildArgs &a = args.t[blockIdx.y];   // a scalar load from the kernel-argument segment
    const int s_in = blockIdx.x;
    if (a.keep_if_zero && a.keep_if_zero[s_in] == 0) return;   // (block-uniform)
    const int s = a.out_scene ? a.out_scene[s_in] : s_in;      // where the index goes
    if (s < 0) return;
    const float *src = a.xyz + (long long)s_in * a.scene_stride;
    int n = a.counts ? a.counts[s_in] : a.max_points;
    n = n < 0 ? 0 : (n > a.max_points ? a.max_points : n);
    sample_bbox_scene(s, src, a.point_stride, n, a.bbox_out);
    const size_t so = (size_t)s * a.cap;
    amk::grid_build_tiles_scene(s, src, a.point_stride, a.cap, n, a.bbox_out, a.GP, a.cell_start, a.ntiles, a.gparams,
                                a.size_out + s, a.pmax_out + s, a.soa_x ? a.soa_x + so : nullptr, a.soa_y ? a.soa_y + so : nullptr,
                                a.soa_z ? a.soa_z + so : nullptr);
}
// so: first scene of the handle this entry writes (a gang launch builds frame f into scenes [f * S, (f + 1) * S) of the handle)
static BuildArgs build_args(amk_kd *kd, const float *d_xyz, int point_stride, long long scene_stride, const int *d_counts,
                            size_t so = 0) {
    return BuildArgs{d_xyz, point_stride, scene_stride, d_counts, kd->max_points, kd->cap,
                     kd->size.p + so, kd->pmax.p + so, kd->bbox.p + so * 6, kd->gpt.p + so * kd->cap,
                     kd->cell_start.p + so * kd->ntiles * (amk::kGridMaxCells + 2), kd->ntiles,
                     kd->gparams.p + so * amk::kGridParamDoubles, nullptr, nullptr, nullptr, nullptr, nullptr};
}
// A handle in nanoflann tie order builds its tree from the index-ordered planes right after the index: the build kernel
// writes them itself (coalesced) instead of kd_records_to_soa_kernel scattering them from the records afterwards.  Returns
// whether `a` now asks for them (false: allocation failed or not wanted -- ensure_soa makes them on demand as before).
static bool build_writes_soa(amk_kd *kd, BuildArgs &a, size_t so = 0) {
    if (!kd->tie_order || kd->cap <= 0) return false;
    if (!kd->x.p || !kd->y.p || !kd->z.p) {   // (a failed allocation leaves what it got: ensure_soa retries the rest)
        const size_t tot = (size_t)kd->n_scenes * kd->cap;
        if ((!kd->x.p && kd->x.alloc(tot) != hipSuccess) || (!kd->y.p && kd->y.alloc(tot) != hipSuccess) ||
            (!kd->z.p && kd->z.alloc(tot) != hipSuccess))
            return false;
    }
    a.soa_x = kd->x.p + so * kd->cap; a.soa_y = kd->y.p + so * kd->cap; a.soa_z = kd->z.p + so * kd->cap;
    return true;
}

// index-ordered planes from the bucket records (position -> cloud index), NaN padding behind them
__global__ __launch_bounds__(256) void kd_records_to_soa_kernel(const float4 *__restrict__ GP, const int *__restrict__ sizes,
                                                                float *__restrict__ X, float *__restrict__ Y,
                                                                float *__restrict__ Z, int cap) {
    const int s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const size_t base = (size_t)s * cap;
    if (i < sizes[s]) {
        const float4 r = GP[base + i];
        const int idx = __float_as_int(r.w);
        X[base + idx] = r.x; Y[base + idx] = r.y; Z[base + idx] = r.z;
    } else {
        const float qnan = __builtin_nanf("");
        X[base + i] = qnan; Y[base + i] = qnan; Z[base + i] = qnan;
    }
}

// makes the SoA planes of `kd` valid on `stream` (no-op when they already are)
static int ensure_soa(amk_kd *kd, hipStream_t stream) {
    if (kd->soa_valid) return AMK_OK;
    if (!kd->x.p || !kd->y.p || !kd->z.p) {
        const size_t tot = (size_t)kd->n_scenes * kd->cap;
        if (!kd->x.p) AMK_HIP(kd->x.alloc(tot));
        if (!kd->y.p) AMK_HIP(kd->y.alloc(tot));
        if (!kd->z.p) AMK_HIP(kd->z.alloc(tot));
    }
    if (kd->cap > 0)
        hipLaunchKernelGGL(kd_records_to_soa_kernel, dim3((kd->cap + 255) / 256, kd->n_scenes), dim3(256), 0, stream,
                           kd->gpt.p, kd->size.p, kd->x.p, kd->y.p, kd->z.p, kd->cap);
    AMK_HIP(hipGetLastError());
    kd->soa_valid = 1;
    return AMK_OK;
}
extern "C" int amk__kd_ensure_soa(amk_kd *kd, void *stream) { return kd ? ensure_soa(kd, (hipStream_t)stream) : AMK_ERR_INVALID_ARG; }

// ------------------------------------------------------------------------------------------------
// search: one wavefront per (scene, group of QPW queries); device scan in kd_device.h
// ------------------------------------------------------------------------------------------------
template <int QPW>
__global__ __launch_bounds__(512) void kd_scan_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z, int cap,
    const int *__restrict__ sizes, const float *__restrict__ pmaxs, int n_scenes,
    const double *__restrict__ queries, int n_queries, int k, int *__restrict__ out_idx,
    double *__restrict__ out_d2, float *__restrict__ out_pts, int *__restrict__ out_cnt) {
    // One wavefront per (scene, group of QPW queries); the waves of a block are consecutive groups of
    // ONE scene, so they stream the same tiles at the same pace and share them in the CU's L1.
    // XCD-aware placement: the dispatcher puts block b on XCD b % 8, so all blocks of one scene are
    // mapped to the same XCD and share that XCD's L2 copy of the scene's cloud.
    const int groups = (n_queries + QPW - 1) / QPW;
    const int wpb = blockDim.x >> 6;
    const int bps = (groups + wpb - 1) / wpb;  // blocks per scene
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = (j / bps) * 8 + xcd;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */;
    const int g = (j % bps) * wpb + w;
    if (s >= n_scenes || g >= groups) return;
    const int lane = threadIdx.x & 63;
    const int size = sizes[s];
    const float *xs = X + (size_t)s * cap, *ys = Y + (size_t)s * cap, *zs = Z + (size_t)s * cap;

    extern __shared__ __attribute__((aligned(16))) unsigned char scan_smem[];
    amk::ScanLds<QPW> *ws = reinterpret_cast<amk::ScanLds<QPW> *>(scan_smem) + w;
    double *qt = reinterpret_cast<double *>(reinterpret_cast<amk::ScanLds<QPW> *>(scan_smem) + wpb) + w * QPW * 3;
    const int q0 = g * QPW;
    const int nvalid = n_queries - q0 < QPW ? n_queries - q0 : QPW;
    // stage this group's queries (a ragged tail group pads with copies of its last query; not stored)
    if (lane < QPW * 3) {
        const int qq = lane / 3 < nvalid ? lane / 3 : nvalid - 1;
        qt[lane] = queries[((size_t)s * n_queries + q0 + qq) * 3 + lane % 3];
    }
    amk::scan_cloud<QPW>(xs, ys, zs, size, pmaxs[s], qt, 3, k, ws);

    // KDTreeTwo::SearchForNearest count rule, kd_tree_two.h:119-124
    const int cnt = size < k ? size : (size > k ? k : 0);
    for (int qq = 0; qq < nvalid; ++qq) {
        const size_t row = (size_t)s * n_queries + q0 + qq;
        if (lane == 0 && out_cnt) out_cnt[row] = cnt;
        if (lane < k) {
            const int li = ws->li[qq][lane];
            const bool ok = lane < cnt && li != amk::kNoIndex;
            const int idx = ok ? li : -1;
            if (out_idx) out_idx[row * k + lane] = idx;
            if (out_d2) out_d2[row * k + lane] = ok ? ws->ld[qq][lane] : DBL_MAX;
            if (out_pts) {
                float *o = out_pts + (row * k + lane) * 3;
                o[0] = ok ? xs[idx] : 0.f;
                o[1] = ok ? ys[idx] : 0.f;
                o[2] = ok ? zs[idx] : 0.f;
            }
        }
    }
}

// search over the bucketed index: one wavefront per (scene, query), four queries of a scene per block
__global__ __launch_bounds__(256) void kd_grid_search_kernel(amk::GridPtrs gpt, const int *__restrict__ sizes, int n_scenes,
                                                             const double *__restrict__ queries, int n_queries,
                                                             int k, int *__restrict__ out_idx,
                                                             double *__restrict__ out_d2,
                                                             float *__restrict__ out_pts, int *__restrict__ out_cnt) {
    __shared__ amk::GridWaveLds wl[4];
    const int bps = (n_queries + 3) / 4;  // blocks per scene
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = (j / bps) * 8 + xcd;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */, lane = threadIdx.x & 63;
    const int q = (j % bps) * 4 + w;
    if (s >= n_scenes || q >= n_queries) return;
    const size_t row = (size_t)s * n_queries + q;
    const double *qp = queries + row * 3;
    double ld;
    int li, lpos;
    const amk::GridScene gs = gpt.scene(s);
    amk::grid_knn(gs, qp[0], qp[1], qp[2], k, ld, li, lpos, &wl[w]);
    const int size = sizes[s];
    const int cnt = size < k ? size : (size > k ? k : 0);  // kd_tree_two.h:119-124
    if (lane == 0 && out_cnt) out_cnt[row] = cnt;
    if (lane < k) {
        const bool ok = lane < cnt && li != amk::kNoIndex;
        const int idx = ok ? li : -1;
        if (out_idx) out_idx[row * k + lane] = idx;
        if (out_d2) out_d2[row * k + lane] = ok ? ld : DBL_MAX;
        if (out_pts) {
            const float4 rec = gs.pt[lpos];  // lpos = 0 for empty slots: a valid address
            float *o = out_pts + (row * k + lane) * 3;
            o[0] = ok ? rec.x : 0.f;
            o[1] = ok ? rec.y : 0.f;
            o[2] = ok ? rec.z : 0.f;
        }
    }
}

// Tie visibility (amk_kd_tie_flags): the k + 1 nearest of every query; flag = 1 when two of them are at the same squared
// distance -- either two kept neighbours, or the k-th kept one and the best rejected one.  Only then can the index list
// (and, at the k-th slot, the neighbour SET) differ from nanoflann's, which keeps the first VISITED of equal distances
// (KNNResultSet::addPoint, nanoflann_two.hpp:219-246) where this library keeps the lowest index.
__global__ __launch_bounds__(256) void kd_tie_flags_kernel(amk::GridPtrs gpt, const int *__restrict__ sizes, int n_scenes,
                                                           const double *__restrict__ queries, int query_stride,
                                                           int n_queries, int k, int *__restrict__ out_flags) {
    __shared__ amk::GridWaveLds wl[4];
    const int bps = (n_queries + 3) / 4;
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int s = (j / bps) * 8 + xcd;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */, lane = threadIdx.x & 63;
    const int q = (j % bps) * 4 + w;
    if (s >= n_scenes || q >= n_queries) return;
    const size_t row = (size_t)s * n_queries + q;
    const double *qp = queries + row * query_stride;
    double ld;
    int li, lpos;
    amk::grid_knn(gpt.scene(s), qp[0], qp[1], qp[2], k + 1, ld, li, lpos, &wl[w]);
    const int size = sizes[s];
    const int cnt = size < k ? size : (size > k ? k : 0);  // what SearchForNearest(k) returns, kd_tree_two.h:119-124
    const double ld_next = __shfl_down(ld, 1);
    const int li_next = __shfl_down(li, 1);
    const bool tie = lane < cnt && li != amk::kNoIndex && li_next != amk::kNoIndex && ld == ld_next;
    const unsigned long long any = __ballot(tie);
    if (lane == 0) out_flags[row] = any != 0ull;
}

// ------------------------------------------------------------------------------------------------
// opt-in nanoflann tie order (kd_exact.h): the reference's own tree beside the bucketed index
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(amk::kExactTopThreads) void kd_exact_build_top_kernel(amk::ExactPtrs ep, const int *__restrict__ sizes) {
    const int s = blockIdx.x;
    amk::exact_build_top(ep.scene(s), sizes[s]);
}
__global__ __launch_bounds__(amk::kExactThreads) void kd_exact_build_kernel(amk::ExactPtrs ep, const int *__restrict__ sizes, int qcap) {
    const int s = blockIdx.x;
    amk::exact_build_rest(ep.scene(s), sizes[s], qcap);
}
// tests only (tests/test_kd_gpu.py): a smaller ring of open nodes, so that the give-up path of exact_build_rest is reachable
// with a cloud that fits a test (0 = the compiled capacity)
static int g_exact_queue_cap = amk::kExactQueue;
extern "C" int amk__exact_set_queue_cap(int cap) {
    if (cap == 0) cap = amk::kExactQueue;
    if (cap < 2 || cap > amk::kExactQueue || (cap & (cap - 1))) return AMK_ERR_INVALID_ARG;
    g_exact_queue_cap = cap;
    return AMK_OK;
}

// one WAVEFRONT per (scene, query): nanoflann's own traversal (kd_exact.h: exact_knn_wave).  Overwrites the outputs of the
// bucketed search (launched before it on the same stream) wherever the tree is available; a scene whose tree is not (node
// capacity or traversal depth exceeded on pathological data) keeps the bucketed index's answer.
__global__ __launch_bounds__(256) void kd_exact_search_kernel(amk::ExactPtrs ep, const int *__restrict__ sizes, int n_scenes,
                                                              const double *__restrict__ queries, int n_queries, int k,
                                                              int *__restrict__ out_idx, double *__restrict__ out_d2,
                                                              float *__restrict__ out_pts, int *__restrict__ out_cnt) {
    __shared__ amk::ExactWaveStack stacks[4];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)  /* wave-uniform: keeps what derives from it in SGPRs */, lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + w;
    if (row >= (size_t)n_scenes * n_queries) return;
    const int s = (int)(row / n_queries);
    const amk::ExactTree T = ep.scene(s);
    const double *qp = queries + row * 3;
    const int size = sizes[s];
    double rd;
    int ri;
    const int got = amk::exact_knn_wave(T, qp[0], qp[1], qp[2], k, rd, ri, &stacks[w]);
    if (got < 0) return;
    const int cnt = size < k ? size : (size > k ? k : 0);  // kd_tree_two.h:119-124
    if (out_cnt && lane == 0) out_cnt[row] = cnt;
    if (lane < k) {
        const int j = lane;
        const bool ok = j < cnt && j < got;
        const int idx = ok ? ri : -1;
        if (out_idx) out_idx[row * k + j] = idx;
        if (out_d2) out_d2[row * k + j] = ok ? rd : DBL_MAX;
        if (out_pts) {
            float *o = out_pts + (row * k + j) * 3;
            o[0] = ok ? T.x[idx] : 0.f;
            o[1] = ok ? T.y[idx] : 0.f;
            o[2] = ok ? T.z[idx] : 0.f;
        }
    }
}

static amk::ExactPtrs exact_ptrs(amk_kd *kd) { return amk_exact_ptrs(kd); }

// builds the reference's tree of every scene from the index-ordered planes (made on demand from the bucket records)
static int exact_build(amk_kd *kd, hipStream_t stream) {
    const size_t S = kd->n_scenes;
    if (!kd->ex_vind.p) {
        kd->ex_max_nodes = kd->cap / 2 + 64;  // ~0.29 nodes per point with 10-point leaves; more = pathological data
        const size_t pc = S * (size_t)kd->cap, nc = S * (size_t)kd->ex_max_nodes;
        AMK_HIP(kd->ex_vind.alloc(pc)); AMK_HIP(kd->ex_sa.alloc(pc)); AMK_HIP(kd->ex_sb.alloc(pc));
        AMK_HIP(kd->ex_pc.alloc(3 * pc));
        AMK_HIP(kd->ex_left.alloc(nc)); AMK_HIP(kd->ex_right.alloc(nc)); AMK_HIP(kd->ex_feat.alloc(nc));
        AMK_HIP(kd->ex_child.alloc(nc)); AMK_HIP(kd->ex_low.alloc(nc)); AMK_HIP(kd->ex_high.alloc(nc));
        AMK_HIP(kd->ex_nbbox.alloc(nc * 6)); AMK_HIP(kd->ex_root.alloc(S * 6)); AMK_HIP(kd->ex_nn.alloc(S));
    }
    const int st = ensure_soa(kd, stream);
    if (st != AMK_OK) return st;
    hipLaunchKernelGGL(kd_exact_build_top_kernel, dim3(kd->n_scenes), dim3(amk::kExactTopThreads), 0, stream, exact_ptrs(kd), kd->size.p);
    hipLaunchKernelGGL(kd_exact_build_kernel, dim3(kd->n_scenes), dim3(amk::kExactThreads), 0, stream, exact_ptrs(kd),
                       kd->size.p, g_exact_queue_cap);
    AMK_HIP(hipGetLastError());
    kd->ex_valid = 1;
    return AMK_OK;
}

// ------------------------------------------------------------------------------------------------
// keyframe sweep (FrameKDMap::KeyframeThreadWorker, AM/src/FrameKDMap.cpp:462-485)
// ------------------------------------------------------------------------------------------------
// one thread per keyframe point: outlier iff its nearest neighbour in the current frame is farther than th.  The points are
// taken in the keyframe's RECORD order (bucket-contiguous: the lanes of a wavefront hold neighbours in space, so their bucket-table
// reads and point reads of the current frame fall into a few cache lines; in cloud order every lane reads its own -- 6.3-7.5 ms
// against 5.2-6.2 ms per 512-scene sweep of the 50 k-point flight frames); the flag goes to the point's cloud index (record.w).
__global__ __launch_bounds__(256) void kd_sweep_mark_kernel(amk::GridPtrs cur, const int *__restrict__ cur_sizes,
                                                            const float4 *__restrict__ KGP, int kcap,
                                                            const int *__restrict__ ksizes, double th_dist,
                                                            unsigned char *__restrict__ flags,
                                                            const int *__restrict__ kf_list = nullptr,
                                                            const int *__restrict__ cur_list = nullptr) {
    // kf_list / cur_list (the keyframe map's pool, kfmap.hip): row blockIdx.y sweeps scene kf_list[row] of the keyframe arrays
    // against scene cur_list[row] of the current-frame arrays; kf_list[row] < 0: nothing to sweep.  Null: scene = row in both.
    const int s = kf_list ? kf_list[blockIdx.y] : blockIdx.y;
    if (s < 0) return;
    const int sc = cur_list ? cur_list[blockIdx.y] : s;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = ksizes[s];
    if (i >= n) return;
    const float4 rec = KGP[(size_t)s * kcap + i];
    unsigned char f = 0;
    if (cur_sizes[sc] > 1) {  // SearchForNearest(pt, 1) yields a result only then (kd_tree_two.h:119-124)
        // (a wave-cooperative second phase for the queries that find no neighbour quickly -- lane j scanning candidate run j of one
        // undecided query at a time -- was built and measured SLOWER: 8.3 ms per 512-scene sweep against 5.2-6.2; the runs of a
        // 13-tile index hold ~5 points each, the phase was all bookkeeping)
        f = (unsigned char)amk::grid_outlier_thread(cur.scene(sc), (double)rec.x, (double)rec.y, (double)rec.z, th_dist);
    }
    flags[(size_t)s * kcap + __float_as_int(rec.w)] = f;
}

// (A persistent-lane version of this kernel -- a wavefront owns 256-1024 records, its lanes draw queries as they finish -- was built when
// the sensor-like flights showed that nearly every wavefront holds an outlier and runs at the outlier's pace: 9.3 -> 6.1 ms per
// 512-scene sweep of 50 k-point frames, but slower on 3072-point frames, and beside the point once the pool swept against a fine grid:
// tools/experiments/patches/r05_sweep_persistent_lanes.patch, profiles/r05_sweep_target.txt.)
static int g_sweep_target = [] { const char *e = getenv("AMK_SWEEP_TARGET"); return e ? atoi(e) : 1; }();   // 1: the pool sweeps against a fine hashed grid of the current frame (below); 0: against the frame's own index (A/B, tests)
extern "C" void amk__sweep_set_target(int v) { g_sweep_target = v; }
static int g_sweep_order = [] { const char *e = getenv("AMK_SWEEP_ORDER"); return e ? atoi(e) : 1; }();   // 1: keyframe points in the order of last sweep's grid where it is theirs; 0: always in record order (A/B)
extern "C" void amk__sweep_set_order(int v) { g_sweep_order = v; }

// one block per scene: count the outliers; with >= th_count of them compact the keyframe's planes in place
// (order preserved: the write cursor never passes the read cursor) and refresh size / bbox / max|coordinate|
__global__ __launch_bounds__(kCompactThreads) void kd_sweep_compact_kernel(
    float *__restrict__ X, float *__restrict__ Y, float *__restrict__ Z, int cap, int *__restrict__ sizes,
    float *__restrict__ pmax_out, float *__restrict__ bbox_out, const unsigned char *__restrict__ flags, int th_count,
    int *__restrict__ sweep_cnt, int *__restrict__ out_outliers, int *__restrict__ out_rebuilt,
    const int *__restrict__ kf_list = nullptr) {
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s = kf_list ? kf_list[row] : row;   // (out_outliers / out_rebuilt / sweep_cnt are per ROW)
    if (s < 0) {
        if (tid == 0) {
            if (sweep_cnt) { sweep_cnt[2 * row] = 0; sweep_cnt[2 * row + 1] = 0; }
            if (out_outliers) out_outliers[row] = 0;
            if (out_rebuilt) out_rebuilt[row] = 0;
        }
        return;
    }
    float *xs = X + (size_t)s * cap, *ys = Y + (size_t)s * cap, *zs = Z + (size_t)s * cap;
    const unsigned char *fl = flags + (size_t)s * cap;
    const int n = sizes[s];
    __shared__ int wave_tot[kCompactThreads / kWave];
    __shared__ float wave_max[kCompactThreads / kWave];
    __shared__ float wave_bb[6][kCompactThreads / kWave];
    __shared__ int total_sh;
    // Four consecutive elements per thread and trip (round 6: one element per trip was 98 trips of dependent loads and two barriers each at
    // 50 k points: 183 us per 512-row launch).  The rows start at multiples of 256 elements and are NaN- / zero-padded to cap >= n + 1024,
    // so the 4-byte flag words and 16-byte coordinate vectors are aligned and may overhang n.
    int cnt = 0;
    for (int i = 4 * tid; i < n; i += 4 * kCompactThreads) {
        const uchar4 f4 = *reinterpret_cast<const uchar4 *>(fl + i);
        cnt += (f4.x != 0) + ((i + 1 < n) & (f4.y != 0)) + ((i + 2 < n) & (f4.z != 0)) + ((i + 3 < n) & (f4.w != 0));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) wave_tot[w] = cnt;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int j = 0; j < kCompactThreads / kWave; ++j) t += wave_tot[j];
        total_sh = t;
    }
    __syncthreads();
    const int total = total_sh;
    const int rebuilt = total >= th_count ? 1 : 0;  // :477-479
    if (tid == 0) {
        if (sweep_cnt) { sweep_cnt[2 * row] = total; sweep_cnt[2 * row + 1] = rebuilt; }
        if (out_outliers) out_outliers[row] = total;
        if (out_rebuilt) out_rebuilt[row] = rebuilt;
    }
    if (!rebuilt) return;
    float amax = 0.f, bmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 4 * kCompactThreads) {
        const int i = c0 + 4 * tid;
        float px[4] = {0.f, 0.f, 0.f, 0.f}, py[4] = {0.f, 0.f, 0.f, 0.f}, pz[4] = {0.f, 0.f, 0.f, 0.f};
        bool valid[4] = {false, false, false, false};
        if (i < n) {
            const float4 x4 = *reinterpret_cast<const float4 *>(xs + i), y4 = *reinterpret_cast<const float4 *>(ys + i),
                         z4 = *reinterpret_cast<const float4 *>(zs + i);
            const uchar4 f4 = *reinterpret_cast<const uchar4 *>(fl + i);
            px[0] = x4.x; px[1] = x4.y; px[2] = x4.z; px[3] = x4.w;
            py[0] = y4.x; py[1] = y4.y; py[2] = y4.z; py[3] = y4.w;
            pz[0] = z4.x; pz[1] = z4.y; pz[2] = z4.z; pz[3] = z4.w;
            valid[0] = f4.x != 0; valid[1] = i + 1 < n && f4.y != 0; valid[2] = i + 2 < n && f4.z != 0; valid[3] = i + 3 < n && f4.w != 0;
        }
        const int mine = (int)valid[0] + (int)valid[1] + (int)valid[2] + (int)valid[3];
        const int incl = amk::wave_incl_scan_i32(mine);
        __syncthreads();  // every read of this chunk is done before anybody writes (in-place; the write cursor never passes the chunk's start)
        if (lane == 63) wave_tot[w] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < kCompactThreads / kWave; ++j) {
            const int t = wave_tot[j];
            woff += (j < w) ? t : 0;
            tot += t;
        }
        int o = base + woff + incl - mine;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (valid[e]) {
                xs[o] = px[e]; ys[o] = py[e]; zs[o] = pz[e];
                ++o;
                amax = fmaxf(amax, fmaxf(fabsf(px[e]), fmaxf(fabsf(py[e]), fabsf(pz[e]))));
                if (amk::finite3(px[e], py[e], pz[e])) {
                    bmn[0] = fminf(bmn[0], px[e]); bmx[0] = fmaxf(bmx[0], px[e]);
                    bmn[1] = fminf(bmn[1], py[e]); bmx[1] = fmaxf(bmx[1], py[e]);
                    bmn[2] = fminf(bmn[2], pz[e]); bmx[2] = fmaxf(bmx[2], pz[e]);
                }
            }
        base += tot;
        __syncthreads();
    }
    const float qnan = __builtin_nanf("");
    for (int i = base + tid; i < n + 1024 && i < cap; i += kCompactThreads) { xs[i] = qnan; ys[i] = qnan; zs[i] = qnan; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, off));
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bmn[a] = fminf(bmn[a], __shfl_xor(bmn[a], off));
            bmx[a] = fmaxf(bmx[a], __shfl_xor(bmx[a], off));
        }
    }
    if (lane == 0) {
        wave_max[w] = amax;
#pragma unroll
        for (int a = 0; a < 3; ++a) { wave_bb[a][w] = bmn[a]; wave_bb[3 + a][w] = bmx[a]; }
    }
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int j = 0; j < kCompactThreads / kWave; ++j) m = fmaxf(m, wave_max[j]);
        sizes[s] = base;
        pmax_out[s] = m;
    }
    if (tid < 6) {
        float v = wave_bb[tid][0];
        for (int j = 1; j < kCompactThreads / kWave; ++j) v = tid < 3 ? fminf(v, wave_bb[tid][j]) : fmaxf(v, wave_bb[tid][j]);
        bbox_out[6 * s + tid] = v;
    }
}

extern "C" int amk_kd_keyframe_sweep(amk_kd *keyframe, amk_kd *current, double th_dist, int th_count, int *d_outliers,
                                     int *d_rebuilt, void *stream_) {
    if (!keyframe || !current || keyframe == current || keyframe->n_scenes != current->n_scenes) return AMK_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int S = keyframe->n_scenes;
    if (!keyframe->flags.p) {
        AMK_HIP(keyframe->flags.alloc((size_t)S * keyframe->cap));
        AMK_HIP(keyframe->sweep_cnt.alloc((size_t)S * 2));
    }
    const amk::GridPtrs cur{current->gpt.p, current->cell_start.p, current->gparams.p, current->cap, current->ntiles};
    {   // the key frame's points are the queries and are compacted in place: they must exist in index order
        const int st = ensure_soa(keyframe, stream);
        if (st != AMK_OK) return st;
    }
    if (keyframe->max_points > 0) {
        hipLaunchKernelGGL(kd_sweep_mark_kernel, dim3((keyframe->max_points + 255) / 256, S), dim3(256), 0, stream, cur,
                           current->size.p, keyframe->gpt.p, keyframe->cap, keyframe->size.p, th_dist, keyframe->flags.p);
    }
    hipLaunchKernelGGL(kd_sweep_compact_kernel, dim3(S), dim3(kCompactThreads), 0, stream, keyframe->x.p, keyframe->y.p,
                       keyframe->z.p, keyframe->cap, keyframe->size.p, keyframe->pmax.p, keyframe->bbox.p, keyframe->flags.p,
                       th_count, keyframe->sweep_cnt.p, d_outliers, d_rebuilt);
    // the bucketed index of every scene is rebuilt (a no-op in effect for the untouched ones)
    hipLaunchKernelGGL(amk::kd_grid_build_kernel, dim3(S), dim3(amk::kGridBuildThreads), 0, stream, keyframe->x.p,
                       keyframe->y.p, keyframe->z.p, keyframe->cap, keyframe->size.p, keyframe->bbox.p, keyframe->gpt.p,
                       keyframe->cell_start.p, keyframe->ntiles, keyframe->gparams.p);
    keyframe->async_pending = 1;
    AMK_HIP(hipGetLastError());
    keyframe->ex_valid = 0;   // a rebuilt keyframe's old tree describes another cloud
    if (keyframe->tie_order) return exact_build(keyframe, stream);  // (the planes are valid: the sweep compacted them)
    return AMK_OK;
}

// ------------------------------------------------------------------------------------------------
// the keyframe map's pool (kfmap.hip): one handle holds P physical frames x S scenes, scene index = slot * S + s
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(amk::kGridBuildThreads) void kd_grid_build_list_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z, int cap,
    const int *__restrict__ sizes, const float *__restrict__ bbox, float4 *__restrict__ GP, int *__restrict__ cell_start,
    int ntiles, double *__restrict__ gparams, const int *__restrict__ list, const int *__restrict__ rebuilt) {
    const int s = list[blockIdx.x];
    if (s < 0 || !rebuilt[blockIdx.x]) return;   // (block-uniform)
    amk::grid_build_scene(s, X + (size_t)s * cap, Y + (size_t)s * cap, Z + (size_t)s * cap, cap, sizes[s], bbox, GP, cell_start,
                          ntiles, gparams);
}

// ------------------------------------------------------------------------------------------------
// The pool's sweep target as a FINE HASHED GRID.  The frames' own indices have cells of ~1 m at 50 k points (<= 1024 cells: what the
// K-NN searches want), the sweep asks "any point within th = 0.1 m?": a query read 70 (inlier) to 1200 (outlier) candidates of the
// few cells its cube touches -- 8.4 ms per 512-scene sweep even with every cell's points in one run, 62 % of a flight's kernel time
// once the frames are a forward-looking sensor's and every robot sweeps every period (profiles/r05_sweep_target.txt).  Here the
// current frame of every sweep row is sorted once more into cubic cells of edge 2.5 th on a world-fixed lattice, hashed into
// kSweepBuckets buckets (a bucket may hold several cells: more candidates, the same answer -- the distance test decides): the cube of
// a query touches <= 2 cells per axis, ~5 points each on a surface.  Points with a non-finite coordinate stay out (they are
// within th of nothing); their absence is what `finite == 0` reports.
// ------------------------------------------------------------------------------------------------
constexpr int kSweepBuckets = 16384;   // at most: 64 KB of LDS histogram per build block; a pool of small frames takes fewer (sweep_buckets)
constexpr int kSweepBuildThreads = 1024;

// Cell of a coordinate: 32-bit, from fp32 arithmetic -- mul, floor, clamp, convert (the fp64 / 64-bit version of round 5 cost ~12 VALU
// instructions per coordinate and ~40 per hash, in kernels that are bound by their VALU instructions).  What the sweep needs of it is
// MONOTONICITY, which every step keeps (round-to-nearest, floor, clamp): a point p with q - rr <= p <= q + rr then has
// cell(q - rr) <= cell(p) <= cell(q + rr), whatever the rounding did to the cell boundaries.
__device__ __forceinline__ int sweep_cell(float p, float inv_hf) {
    return (int)fminf(fmaxf(floorf(p * inv_hf), -5.0e8f), 5.0e8f);
}
__device__ __forceinline__ int sweep_cell(double p, float inv_hf) { return sweep_cell((float)p, inv_hf); }
// Bucket of a cell.  Cells are hashed BLOCK-wise: a block of 4 x 4 x 4 cells (10 th = 1 m at th 0.1) owns 64 CONSECUTIVE buckets, the hash
// only picks which run of 64 (nb / 64 runs).  Neighbouring cells therefore share a run unless they straddle a block face: queries taken in
// grid order (kd_sweep_mapped's two generations) find their <= 8 cells' table entries and records near each other (mark kernel 1.56 ->
// 1.47 ms per 512 x 50 k sweep against the cell-wise hash).  Cells per bucket are what a cell-wise hash gives (cells / nb on average).
// (A workgroup per run with the run's records staged in LDS was built on top of this and is SLOWER, 1.94 ms: the kernel is bound by its
// VALU instructions and their divergence, not by its gathers -- tools/experiments/patches/r06_sweep_block_lds.patch, profiles/r06_sweep.txt.)
constexpr int kSweepBlockCells = 64;   // 4 x 4 x 4
__device__ __forceinline__ int sweep_block(int ix, int iy, int iz, int nb) {
    unsigned h = (unsigned)(ix >> 2) * 73856093u ^ (unsigned)(iy >> 2) * 19349663u ^ (unsigned)(iz >> 2) * 83492791u;
    h ^= h >> 15;
    return (int)(h & (unsigned)(nb / kSweepBlockCells - 1));
}
__device__ __forceinline__ int sweep_local(int ix, int iy, int iz) { return (ix & 3) | ((iy & 3) << 2) | ((iz & 3) << 4); }
__device__ __forceinline__ int sweep_bucket(int ix, int iy, int iz, int nb) {
    return sweep_block(ix, iy, iz, nb) * kSweepBlockCells + sweep_local(ix, iy, iz);
}
// buckets of a pool's sweep grids: a power of two, about two per point, between 1024 and kSweepBuckets
static int sweep_buckets(int max_points) {
    static const int forced = [] { const char *e = getenv("AMK_SWEEP_NB"); return e ? atoi(e) : 0; }();   // (experiments: a power of two in [1024, 16384])
    if (forced >= 1024 && forced <= kSweepBuckets && (forced & (forced - 1)) == 0) return forced;
    int nb = 1024;
    while (nb < kSweepBuckets && nb < 2 * max_points) nb *= 2;
    return nb;
}

// one block per sweep row: counting sort of the current frame's points by bucket (LDS histogram, block scan, LDS cursors).  The points are
// read from the frame's OWN bucketed records (kd_build's output: coarse-cell order, ~1 m cells), not from the index-ordered planes: the 64
// points of a wave-instruction then lie in one or two blocks of fine cells and their 16-byte records are scattered into a few KB instead of
// all over the row's 800 KB (round 6: the scatter, not the arithmetic, is what this kernel's time is).
__global__ __launch_bounds__(kSweepBuildThreads) void kd_sweep_hash_build_kernel(
    const float4 *__restrict__ GP, int cap, const int *__restrict__ sizes,
    double inv_h, int nb, float4 *__restrict__ recs, int *__restrict__ table, const int *__restrict__ kf_list,
    const int *__restrict__ cur_list, int *__restrict__ src, const int *__restrict__ src_prev, unsigned char *__restrict__ flags) {
    extern __shared__ int hist[];   // [nb] + [kSweepBuildThreads / 64] wave sums
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float inv_hf = (float)inv_h;
    const int kf = kf_list[row];
    if (kf < 0) {   // (block-uniform) no sweep in this row: whatever grid this generation held for it is not the next sweep's keyframe
        if (tid == 0) src[row] = -1;
        return;
    }
    const int sc = cur_list[row];
    if (tid == 0) src[row] = sc;
    // the keyframe is the frame this row swept against last time: the mark kernel takes its points from that grid, which leaves out the
    // points with a non-finite coordinate -- their flag is 0 (no outlier: within th of nothing, but SearchForNearest gives them no result)
    if (src_prev[row] == kf)
        for (int i = tid; i < sizes[kf]; i += kSweepBuildThreads) flags[(size_t)kf * cap + i] = 0;
    const int n = sizes[sc];
    const float4 *in = GP + (size_t)sc * cap;   // the frame's own records: coarse-cell order (see above)
    int *wsum = hist + nb;
    int *tab = table + (size_t)row * (nb + 1);
    float4 *out = recs + (size_t)row * cap;
    for (int i = tid; i < nb; i += kSweepBuildThreads) hist[i] = 0;
    __syncthreads();
    constexpr int U = 4;   // points per thread and trip: their loads fly together (a block's passes are two chains of dependent trips)
    for (int i0 = tid; i0 < n; i0 += U * kSweepBuildThreads) {
        float x[U], y[U], z[U];
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 r = in[min(i0 + u * kSweepBuildThreads, n - 1)];
            x[u] = r.x; y[u] = r.y; z[u] = r.z; id[u] = __float_as_int(r.w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * kSweepBuildThreads < n && amk::finite3(x[u], y[u], z[u]))
                atomicAdd(&hist[sweep_bucket(sweep_cell(x[u], inv_hf), sweep_cell(y[u], inv_hf), sweep_cell(z[u], inv_hf), nb)], 1);
    }
    __syncthreads();
    // exclusive scan of the nb counts: nb / kSweepBuildThreads consecutive buckets per thread (nb >= 1024 = the block)
    const int per = nb / kSweepBuildThreads;
    int loc = 0;
    for (int j = 0; j < per; ++j) loc += hist[tid * per + j];
    const int incl = amk::wave_incl_scan_i32(loc);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = 0;
    for (int j = 0; j < w; ++j) base += wsum[j];
    int run = base + incl - loc;
    for (int j = 0; j < per; ++j) {
        const int c = hist[tid * per + j];
        hist[tid * per + j] = run;   // the bucket's cursor
        tab[tid * per + j] = run;
        run += c;
    }
    if (tid == kSweepBuildThreads - 1) tab[nb] = run;   // = the points with finite coordinates
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += U * kSweepBuildThreads) {
        float x[U], y[U], z[U];
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 r = in[min(i0 + u * kSweepBuildThreads, n - 1)];
            x[u] = r.x; y[u] = r.y; z[u] = r.z; id[u] = __float_as_int(r.w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSweepBuildThreads;
            if (i < n && amk::finite3(x[u], y[u], z[u])) {
                const int pos = atomicAdd(&hist[sweep_bucket(sweep_cell(x[u], inv_hf), sweep_cell(y[u], inv_hf), sweep_cell(z[u], inv_hf), nb)], 1);
#ifndef AMK_DIAG_NOSTORE   // (diagnostics: without the scatter the kernel takes 169 of its 458 us)
                out[pos] = make_float4(x[u], y[u], z[u], __int_as_float(id[u]));
#else
                if (pos == -12345) out[0] = make_float4(x[u], y[u], z[u], 0.f);
#endif   // (order inside a bucket: whatever the atomics gave -- the sweep asks "any", not "which")
            }
        }
    }
}

// kSweepStepH records against one query.  The counters of the first version said what bounds this kernel: 3 000 VALU instructions per
// wavefront -- the VALU pipes ~90 % busy, most of it the candidates' fp64 distances (11 instructions each: three conversions, the
// differences, the squares, the sum) -- not the loads (TA 66 % busy, 81 % of L2 requests hit).  So the candidates are screened in fp32 first: both
// points ARE floats, the fp32 squared distance is within 3e-7 relative of the real one, and a candidate above t2 (1 + 1e-5) cannot pass
// the exact test; only the few below it get the fp64 distance and the exact test (same bits as before: same flags).
__device__ __forceinline__ bool sweep_step_hits(const float4 &q, const float4 (&pr)[4], float t2f, double qx, double qy, double qz, double th,
                                                double t2lo, double t2hi) {
    bool hit = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float dx = q.x - pr[e].x, dy = q.y - pr[e].y, dz = q.z - pr[e].z;
        const float d32 = dx * dx + dy * dy + dz * dz;
        if (d32 <= t2f) {
            const double d = amk::sq_dist(qx, qy, qz, pr[e].x, pr[e].y, pr[e].z);
            hit = hit || d <= t2lo || (d <= t2hi && sqrt(d) <= th);
        }
    }
    return hit;
}

// one thread per keyframe record (record order): the <= 8 cells of its cube, its own cell first; their table entries fetched together,
// then kSweepStepH records of a run per step.  A query whose cube would span more than two cells along an axis (coordinates so large
// that the rounding allowance exceeds the cell) reads every point of the grid instead.
constexpr int kSweepStepH = 4;   // (= the array bound of sweep_step_hits)
#ifndef AMK_SWEEP_MARK_THREADS
#define AMK_SWEEP_MARK_THREADS 256
#endif
constexpr int kSweepMarkThreads = AMK_SWEEP_MARK_THREADS;
__global__ __launch_bounds__(kSweepMarkThreads) void kd_sweep_mark_hash_kernel(const int *__restrict__ cur_sizes,
                                                                 const float4 *__restrict__ trecs, const int *__restrict__ table,
                                                                 double inv_h, int nb, const float4 *__restrict__ KGP, int kcap,
                                                                 const int *__restrict__ ksizes, double th,
                                                                 unsigned char *__restrict__ flags, const int *__restrict__ kf_list,
                                                                 const int *__restrict__ cur_list, const float4 *__restrict__ prev_recs,
                                                                 const int *__restrict__ prev_table, const int *__restrict__ src_prev) {
    const int row = blockIdx.y;
    const int s = kf_list[row];
    if (s < 0) return;
    const int sc = cur_list[row];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the keyframe's points: in the order of the grid it was sorted into when it was the current frame (the sweep before this one, same
    // row) -- the lanes of a wavefront then ask for the same few buckets and runs -- else in its own records' order
    const bool ordered = src_prev[row] == s;   // (row-uniform)
    const bool live = i < (ordered ? prev_table[(size_t)row * (nb + 1) + nb] : ksizes[s]);
    // Two phases per workgroup.  A: every thread scans its query's OWN cell (half the inliers end there).  B: the queries that are still open
    // are compacted into LDS and taken by the first threads of the block, one each, for the other <= 7 cells of their cubes.  One query per
    // thread throughout made nearly every wavefront run at its slowest lane's pace (an outlier scans all 8 cells: a real pair of frames with
    // 14 % outliers cost 90 % of an all-outlier pair); after the compaction the wavefronts of phase B are full of open queries and the others
    // have retired.  Same cells, same screen, same exact test: same flags.
    __shared__ float4 open_q[kSweepMarkThreads];
    __shared__ int4 open_c[kSweepMarkThreads];   // the open query's own cell and, per axis, which other cell its cube reaches (bits 0-2: has one, 3-5: it is the next one up)
    __shared__ int n_open;
    if (threadIdx.x == 0) n_open = 0;
    __syncthreads();
    const int *tab = table + (size_t)row * (nb + 1);
    const float4 *pts = trecs + (size_t)row * kcap;
    const double h = 1.0 / inv_h;
    const float inv_hf = (float)inv_h;
    const double t2 = th * th, t2lo = t2 * (1.0 - 1e-15), t2hi = t2 * (1.0 + 1e-15);
    const float t2f = (float)(t2 * (1.0 + 1e-5)) * (1.0f + 1e-6f);   // fp32 screen: above it no candidate can pass the exact test
    const bool usable = cur_sizes[sc] > 1 && tab[nb] > 0;   // SearchForNearest(pt, 1) yields a result only for a tree of more than one point
                                                            // (kd_tree_two.h:119-124), and an outlier needs a nearest point at all
    if (live) {
        const float4 rec = ordered ? prev_recs[(size_t)row * kcap + i] : KGP[(size_t)s * kcap + i];
        unsigned char f = 0;
        bool open = false;
        int4 oc = make_int4(0, 0, 0, 0);
        const double qx = (double)rec.x, qy = (double)rec.y, qz = (double)rec.z;
        if (usable && qx == qx && qy == qy && qz == qz) {
            const double rr = th + 1e-9 * h + 1e-12 * (fabs(qx) + fabs(qy) + fabs(qz) + th);   // (rounding allowance, as grid_outlier_thread's)
            const int lx = sweep_cell(qx - rr, inv_hf), hx = sweep_cell(qx + rr, inv_hf);
            const int ly = sweep_cell(qy - rr, inv_hf), hy = sweep_cell(qy + rr, inv_hf);
            const int lz = sweep_cell(qz - rr, inv_hf), hz = sweep_cell(qz + rr, inv_hf);
            f = 1;
            if (hx - lx > 1 || hy - ly > 1 || hz - lz > 1) {
                // coordinates so large that the rounding allowance exceeds a cell: every point of the grid is a candidate (a plain loop)
                for (int j = 0; f && j < tab[nb]; ++j) {
                    const float4 p = pts[j];
                    const double d = amk::sq_dist(qx, qy, qz, p.x, p.y, p.z);
                    if (d <= t2lo || (d <= t2hi && sqrt(d) <= th)) f = 0;
                }
            } else {
                const int ox = sweep_cell(rec.x, inv_hf), oy = sweep_cell(rec.y, inv_hf), oz = sweep_cell(rec.z, inv_hf);   // own cell: within [l, h]
                const int b = sweep_bucket(ox, oy, oz, nb);
                const int s0 = tab[b], s1 = tab[b + 1];
                for (int pos = s0; f && pos < s1; pos += kSweepStepH) {
                    const int last = s1 - 1;
                    float4 pr[kSweepStepH];
#pragma unroll
                    for (int e = 0; e < kSweepStepH; ++e) pr[e] = pts[min(pos + e, last)];
                    if (sweep_step_hits(rec, pr, t2f, qx, qy, qz, th, t2lo, t2hi)) f = 0;
                }
                open = f && (hx != lx || hy != ly || hz != lz);
                oc = make_int4(ox, oy, oz, (hx != lx ? 1 : 0) | (hy != ly ? 2 : 0) | (hz != lz ? 4 : 0) | (ox == lx ? 8 : 0) | (oy == ly ? 16 : 0) | (oz == lz ? 32 : 0));
            }
        }
        if (open) { const int slot = atomicAdd(&n_open, 1); open_q[slot] = rec; open_c[slot] = oc; }
        else flags[(size_t)s * kcap + __float_as_int(rec.w)] = f;
    }
    __syncthreads();
#ifdef AMK_SWEEP_SKIPB
    if (false) {
#else
    if ((int)threadIdx.x < n_open) {
#endif
        const float4 rec = open_q[threadIdx.x];
        const int4 oc = open_c[threadIdx.x];
        const double qx = (double)rec.x, qy = (double)rec.y, qz = (double)rec.z;
        // the other cell of an axis: the next one up or down (phase A's cube [l, h] with h - l <= 1 around the own cell)
        const int ax = oc.x + ((oc.w & 8) ? 1 : -1), ay = oc.y + ((oc.w & 16) ? 1 : -1), az = oc.z + ((oc.w & 32) ? 1 : -1);
        const int mask = oc.w & 7;
        int s0[8], s1[8];
        // cells that share a FACE with the own cell first (one bit), then edges, then the corner: an inlier's neighbour is most often there
        constexpr int kOrder[8] = {0, 1, 2, 4, 3, 5, 6, 7};
#pragma unroll
        for (int j = 1; j < 8; ++j) {   // bit a of kk set = the OTHER cell of axis a; the table entries fetched together
            const int kk = kOrder[j];
            s0[j] = s1[j] = 0;
            if ((kk & ~mask) == 0) {
                const int b = sweep_bucket((kk & 1) ? ax : oc.x, (kk & 2) ? ay : oc.y, (kk & 4) ? az : oc.z, nb);
                s0[j] = tab[b];
                s1[j] = tab[b + 1];
            }
        }
        unsigned char f = 1;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            for (int pos = s0[k]; f && pos < s1[k]; pos += kSweepStepH) {
                const int last = s1[k] - 1;
                float4 pr[kSweepStepH];
#pragma unroll
                for (int e = 0; e < kSweepStepH; ++e) pr[e] = pts[min(pos + e, last)];
                if (sweep_step_hits(rec, pr, t2f, qx, qy, qz, th, t2lo, t2hi)) f = 0;
            }
        }
        flags[(size_t)s * kcap + __float_as_int(rec.w)] = f;
    }
}

namespace amk {
static int pool_planes(amk_kd *kd) {   // the index-ordered planes of a pool handle: written by every build, compacted by the sweeps
    const size_t tot = (size_t)kd->n_scenes * kd->cap;
    if (!kd->x.p) AMK_HIP(kd->x.alloc(tot));
    if (!kd->y.p) AMK_HIP(kd->y.alloc(tot));
    if (!kd->z.p) AMK_HIP(kd->z.alloc(tot));
    return AMK_OK;
}

// Everything a map's obstacle pool will ever allocate, at once (amk_kfmap_create): the index-ordered planes, the sweep's outlier flags and
// its two generations of hashed grids -- so that a pool that does not fit fails when the map is CREATED (amk_kfmap_pool_bytes says what it
// needs), not at some later submit when the first sweep runs.
int kd_pool_reserve(amk_kd *pool, int n_rows) {
    if (!pool || n_rows < 1) return AMK_ERR_INVALID_ARG;
    const int st = pool_planes(pool);
    if (st != AMK_OK) return st;
    if (!pool->flags.p) AMK_HIP(pool->flags.alloc((size_t)pool->n_scenes * pool->cap));
    if (pool->max_points > 0 && pool->sw_rows < n_rows) {
        const int nb = sweep_buckets(pool->max_points);
        AMK_HIP(pool->sw_gpt.alloc((size_t)2 * n_rows * pool->cap));
        AMK_HIP(pool->sw_cs.alloc((size_t)2 * n_rows * (nb + 1)));
        AMK_HIP(pool->sw_src.alloc((size_t)3 * n_rows));
        AMK_HIP(hipMemset(pool->sw_src.p, 0xff, sizeof(int) * 3 * (size_t)n_rows));
        pool->sw_rows = n_rows;
    }
    return AMK_OK;
}

// FrameKDMap::AddVertex's two InitializeNew calls (FrameKDMap.cpp:44-47) for n_in scenes, each into the pool scene
// d_out_scene[s] (< 0: that scene gets no new frame); the obstacle pool's planes are written along.
int kd_build_mapped(amk_kd *obs_pool, amk_kd *edge_pool, int n_in, const float *d_xyz, const int *d_counts,
                    const float *d_edge_xyz, const int *d_edge_counts, int point_stride, const int *d_out_scene,
                    hipStream_t stream) {
    if (!obs_pool || !edge_pool || n_in < 1 || !d_xyz || !d_edge_xyz || !d_out_scene || point_stride < 3) return AMK_ERR_INVALID_ARG;
    const int st = pool_planes(obs_pool);
    if (st != AMK_OK) return st;
    BuildArgs a = build_args(obs_pool, d_xyz, point_stride, (long long)obs_pool->max_points * point_stride, d_counts);
    BuildArgs b = build_args(edge_pool, d_edge_xyz, point_stride, (long long)edge_pool->max_points * point_stride, d_edge_counts);
    a.out_scene = b.out_scene = d_out_scene;
    a.soa_x = obs_pool->x.p; a.soa_y = obs_pool->y.p; a.soa_z = obs_pool->z.p;
    BuildArgs2 args{};
    args.t[0] = a; args.t[1] = b;
    {
        amk::TimedLaunch tg(amk::KC_GRID, stream);
        hipLaunchKernelGGL(kd_build_kernel, dim3(n_in, 2), dim3(kCompactThreads), 0, stream, args);
    }
    AMK_HIP(hipGetLastError());
    for (amk_kd *kd : {obs_pool, edge_pool}) { kd->soa_valid = 0; kd->ex_valid = 0; kd->async_pending = 1; }
    return AMK_OK;
}

// The same for the G frames of a pipeline gang in ONE launch (grid.y = 2 G trees): frame f's scenes are built into the pool scenes
// d_out_scene[f * frame_scenes + s].  (One add_vertex per gang position was 2 G launches per period: the map's periods at the
// reference's 3072-point frames are bound by their launch count.)
int kd_build_mapped_gang(amk_kd *obs_pool, amk_kd *edge_pool, int n_frames, int frame_scenes, const float *const *d_xyz,
                         const int *const *d_counts, const float *const *d_edge_xyz, const int *const *d_edge_counts, int point_stride,
                         const int *d_out_scene, hipStream_t stream) {
    if (!obs_pool || !edge_pool || n_frames < 1 || n_frames > AMK_PIPELINE_MAX_GANG || frame_scenes < 1 || !d_out_scene || point_stride < 3)
        return AMK_ERR_INVALID_ARG;
    const int st = pool_planes(obs_pool);
    if (st != AMK_OK) return st;
    BuildArgs2 args{};
    for (int f = 0; f < n_frames; ++f) {
        if (!d_xyz[f] || !d_edge_xyz[f]) return AMK_ERR_INVALID_ARG;
        BuildArgs a = build_args(obs_pool, d_xyz[f], point_stride, (long long)obs_pool->max_points * point_stride, d_counts[f]);
        BuildArgs b = build_args(edge_pool, d_edge_xyz[f], point_stride, (long long)edge_pool->max_points * point_stride, d_edge_counts[f]);
        a.out_scene = b.out_scene = d_out_scene + (size_t)f * frame_scenes;
        a.soa_x = obs_pool->x.p; a.soa_y = obs_pool->y.p; a.soa_z = obs_pool->z.p;
        args.t[2 * f] = a; args.t[2 * f + 1] = b;
    }
    {
        amk::TimedLaunch tg(amk::KC_GRID, stream);
        hipLaunchKernelGGL(kd_build_kernel, dim3(frame_scenes, 2 * n_frames), dim3(kCompactThreads), 0, stream, args);
    }
    AMK_HIP(hipGetLastError());
    for (amk_kd *kd : {obs_pool, edge_pool}) { kd->soa_valid = 0; kd->ex_valid = 0; kd->async_pending = 1; }
    return AMK_OK;
}

// KeyframeThreadWorker's sweep (FrameKDMap.cpp:463-485) for n_rows scenes of a map: row r sweeps the points of pool scene
// d_kf_list[r] (the newest keyframe) against pool scene d_cur_list[r] (the current frame); with >= th_count outliers the
// keyframe's planes are compacted to them in place and its index is rebuilt.  d_outliers / d_rebuilt: [n_rows].
int kd_sweep_mapped(amk_kd *pool, int n_rows, const int *d_kf_list, const int *d_cur_list, double th_dist, int th_count,
                    int *d_outliers, int *d_rebuilt, hipStream_t stream) {
    if (!pool || n_rows < 1 || !d_kf_list || !d_cur_list || !d_rebuilt) return AMK_ERR_INVALID_ARG;
    int st = pool_planes(pool);
    if (st != AMK_OK) return st;
    if (!pool->flags.p) AMK_HIP(pool->flags.alloc((size_t)pool->n_scenes * pool->cap));
    const amk::GridPtrs cur{pool->gpt.p, pool->cell_start.p, pool->gparams.p, pool->cap, pool->ntiles};
    if (pool->max_points > 0 && g_sweep_target) {   // the current frames once more, as fine hashed grids (one per sweep row)
        const int nb = sweep_buckets(pool->max_points);
        if (pool->sw_rows < n_rows) {
            if (pool->sw_rows > 0) AMK_HIP(hipDeviceSynchronize());   // (growing: earlier sweeps may still read the old arrays)
            AMK_HIP(pool->sw_gpt.alloc((size_t)2 * n_rows * pool->cap));
            AMK_HIP(pool->sw_cs.alloc((size_t)2 * n_rows * (nb + 1)));
            AMK_HIP(pool->sw_src.alloc((size_t)3 * n_rows));   // (a third row of -1s: "no grid", what the A/B switch hands the mark kernel)
            AMK_HIP(hipMemset(pool->sw_src.p, 0xff, sizeof(int) * 3 * (size_t)n_rows));   // -1: no grid yet
            pool->sw_rows = n_rows;
        }
        // > 64 KB of dynamic LDS needs the attribute; it is per device and the call is cheap, so it is made before every launch
        // (a process-wide flag would leave a second device, or a second thread's first launch, without it)
        AMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kd_sweep_hash_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(int) * (kSweepBuckets + kSweepBuildThreads / 64)) + 65536));
        static const double factor = [] { const char *e = getenv("AMK_SWEEP_CELL"); const double v = e ? atof(e) : 2.5; return v >= 2.1 ? v : 2.5; }();
        const double cell = fmax(factor * th_dist, 1e-3);   // edge of a cell: the cube [q - th, q + th] touches <= 2 cells per axis
        const double inv_h = 1.0 / cell;
        // generations alternate per call; a call with another row count than the one the arrays were sized for is refused (rows are the
        // map's scenes)
        if (n_rows != pool->sw_rows) return AMK_ERR_INVALID_ARG;
        if (pool->sw_inv_h != inv_h) {   // another lattice than the last call's: the previous generation's grids are not this one's cells
            if (pool->sw_inv_h != 0.0) AMK_HIP(hipMemsetAsync(pool->sw_src.p, 0xff, sizeof(int) * 2 * (size_t)n_rows, stream));
            pool->sw_inv_h = inv_h;
        }
        const int g = pool->sw_flip ^= 1;
        float4 *recs_g = pool->sw_gpt.p + (size_t)g * n_rows * pool->cap, *recs_p = pool->sw_gpt.p + (size_t)(g ^ 1) * n_rows * pool->cap;
        int *tab_g = pool->sw_cs.p + (size_t)g * n_rows * (nb + 1), *tab_p = pool->sw_cs.p + (size_t)(g ^ 1) * n_rows * (nb + 1);
        int *src_g = pool->sw_src.p + (size_t)g * n_rows, *src_p = pool->sw_src.p + (size_t)(g ^ 1) * n_rows;
        static const int extra_lds = [] { const char *e = getenv("AMK_SWEEP_BUILD_EXTRA_LDS"); return e ? atoi(e) : 0; }();   // (experiments: blocks per CU)
        hipLaunchKernelGGL(kd_sweep_hash_build_kernel, dim3(n_rows), dim3(kSweepBuildThreads), sizeof(int) * (nb + kSweepBuildThreads / 64) + extra_lds,
                           stream, pool->gpt.p, pool->cap, pool->size.p, inv_h, nb, recs_g, tab_g, d_kf_list, d_cur_list,
                           src_g, src_p, pool->flags.p);
        hipLaunchKernelGGL(kd_sweep_mark_hash_kernel, dim3((pool->max_points + kSweepMarkThreads - 1) / kSweepMarkThreads, n_rows), dim3(kSweepMarkThreads), 0, stream, pool->size.p,
                           recs_g, tab_g, inv_h, nb, pool->gpt.p, pool->cap, pool->size.p, th_dist, pool->flags.p, d_kf_list, d_cur_list,
                           recs_p, tab_p, g_sweep_order ? src_p : pool->sw_src.p + (size_t)2 * n_rows);
    }
    else if (pool->max_points > 0)
        hipLaunchKernelGGL(kd_sweep_mark_kernel, dim3((pool->max_points + 255) / 256, n_rows), dim3(256), 0, stream, cur, pool->size.p,
                           pool->gpt.p, pool->cap, pool->size.p, th_dist, pool->flags.p, d_kf_list, d_cur_list);
    hipLaunchKernelGGL(kd_sweep_compact_kernel, dim3(n_rows), dim3(kCompactThreads), 0, stream, pool->x.p, pool->y.p, pool->z.p,
                       pool->cap, pool->size.p, pool->pmax.p, pool->bbox.p, pool->flags.p, th_count, (int *)nullptr, d_outliers,
                       d_rebuilt, d_kf_list);
    hipLaunchKernelGGL(kd_grid_build_list_kernel, dim3(n_rows), dim3(amk::kGridBuildThreads), 0, stream, pool->x.p, pool->y.p,
                       pool->z.p, pool->cap, pool->size.p, pool->bbox.p, pool->gpt.p, pool->cell_start.p, pool->ntiles,
                       pool->gparams.p, d_kf_list, d_rebuilt);
    AMK_HIP(hipGetLastError());
    pool->async_pending = 1;
    return AMK_OK;
}
}  // namespace amk

extern "C" int amk_kd_keyframe_sweep_host(amk_kd *keyframe, amk_kd *current, double th_dist, int th_count,
                                          int *h_outliers, int *h_rebuilt) {
    int st = amk_kd_keyframe_sweep(keyframe, current, th_dist, th_count, nullptr, nullptr, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipDeviceSynchronize());
    std::vector<int> tmp((size_t)keyframe->n_scenes * 2);
    AMK_HIP(hipMemcpy(tmp.data(), keyframe->sweep_cnt.p, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost));
    for (int s = 0; s < keyframe->n_scenes; ++s) {
        if (h_outliers) h_outliers[s] = tmp[2 * s];
        if (h_rebuilt) h_rebuilt[s] = tmp[2 * s + 1];
    }
    return AMK_OK;
}

extern "C" int amk_kd_points_host(amk_kd *kd, float *h_xyz, int *h_sizes) {
    if (!kd || !h_xyz || !h_sizes) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipDeviceSynchronize());
    {
        const int st = ensure_soa(kd, nullptr);
        if (st != AMK_OK) return st;
        AMK_HIP(hipDeviceSynchronize());
    }
    AMK_HIP(hipMemcpy(h_sizes, kd->size.p, sizeof(int) * kd->n_scenes, hipMemcpyDeviceToHost));
    std::vector<float> plane((size_t)kd->cap);
    for (int s = 0; s < kd->n_scenes; ++s) {
        const int n = h_sizes[s];
        const float *src[3] = {kd->x.p, kd->y.p, kd->z.p};
        for (int c = 0; c < 3; ++c) {
            if (n > 0) AMK_HIP(hipMemcpy(plane.data(), src[c] + (size_t)s * kd->cap, sizeof(float) * n, hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i) h_xyz[((size_t)s * kd->max_points + i) * 3 + c] = plane[i];
        }
    }
    return AMK_OK;
}

// internal (tests / benchmarks): 0 = bucketed index (default), 1 = streaming scan
extern "C" int amk__kd_set_mode(amk_kd *kd, int mode) {
    if (!kd) return AMK_ERR_INVALID_ARG;
    kd->mode = mode;
    return AMK_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int amk_version(void) { return 100; }

const char *amk_status_string(int status) {
    switch (status) {
        case AMK_OK: return "ok";
        case AMK_ERR_INVALID_ARG: return "invalid argument";
        case AMK_ERR_HIP: return "HIP runtime error";
        case AMK_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case AMK_ERR_UNSUPPORTED: return "unsupported size";
        case AMK_ERR_TIMEOUT: return "timed out waiting for a collective (amk_shard_wait)";
        default: return "unknown status";
    }
}

int amk_last_hip_error(void) { return amk::g_last_hip_error; }

int amk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int amk_kd_create(int n_scenes, int max_points, amk_kd **out) {
    if (!out || n_scenes <= 0 || max_points < 0) return AMK_ERR_INVALID_ARG;
    *out = nullptr;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    amk_kd *kd = new amk_kd();
    kd->n_scenes = n_scenes;
    kd->max_points = max_points;
    kd->cap = amk::round_up(max_points, 256) + 1024;  // NaN padding: full vector loads + 3-tile look-ahead
    const size_t tot = (size_t)n_scenes * kd->cap;
    hipError_t e;
    // the index-ordered planes x/y/z are allocated by ensure_soa, the first time something asks for them
    if ((e = kd->size.alloc(n_scenes)) != hipSuccess ||
        (e = kd->pmax.alloc(n_scenes)) != hipSuccess || (e = hipMemset(kd->pmax.p, 0, sizeof(float) * n_scenes)) != hipSuccess) {
        delete kd;
        return amk::hip_fail(e);
    }
    if ((e = hipMemset(kd->size.p, 0, sizeof(int) * n_scenes)) != hipSuccess ||
        (e = kd->gpt.alloc(tot)) != hipSuccess || (e = kd->bbox.alloc((size_t)n_scenes * 6)) != hipSuccess ||
        (e = kd->cell_start.alloc((size_t)n_scenes * (kd->ntiles = amk::grid_tiles(max_points)) * (amk::kGridMaxCells + 2))) != hipSuccess ||
        (e = kd->gparams.alloc((size_t)n_scenes * amk::kGridParamDoubles)) != hipSuccess ||
        (e = hipMemset(kd->cell_start.p, 0, sizeof(int) * (size_t)n_scenes * kd->ntiles * (amk::kGridMaxCells + 2))) != hipSuccess ||
        (e = hipMemset(kd->gparams.p, 0, sizeof(double) * (size_t)n_scenes * amk::kGridParamDoubles)) != hipSuccess) {
        delete kd;
        return amk::hip_fail(e);
    }
    *out = kd;
    return AMK_OK;
}

int amk_kd_destroy(amk_kd *kd) {
    if (kd && kd->hpin) (void)hipHostFree(kd->hpin);
    if (kd && kd->hstream) (void)hipStreamDestroy(kd->hstream);
    if (!kd) return AMK_ERR_INVALID_ARG;
    delete kd;
    return AMK_OK;
}

int amk_kd_build(amk_kd *kd, const float *d_xyz, int point_stride, long long scene_stride,
                 const int *d_counts, void *stream) {
    if (!kd || (!d_xyz && kd->max_points > 0) || point_stride < 3 || scene_stride < 0) return AMK_ERR_INVALID_ARG;
    {
        amk::TimedLaunch tg(amk::KC_GRID, (hipStream_t)stream);
        BuildArgs a = build_args(kd, d_xyz, point_stride, scene_stride, d_counts);
        const bool soa = build_writes_soa(kd, a);
        BuildArgs2 args{};
        args.t[0] = a;
        hipLaunchKernelGGL(kd_build_kernel, dim3(kd->n_scenes, 1), dim3(kCompactThreads), 0, (hipStream_t)stream, args);
        kd->soa_valid = soa ? 1 : 0;
        kd->ex_valid = 0;   // the exact tree (if any) describes the previous cloud until exact_build has run
        kd->async_pending = 1;
    }
    AMK_HIP(hipGetLastError());
    if (kd->tie_order) return exact_build(kd, (hipStream_t)stream);
    return AMK_OK;
}

int amk_kd_build_pair(amk_kd *obstacle, const float *d_xyz, const int *d_counts, amk_kd *edge, const float *d_edge_xyz,
                      const int *d_edge_counts, int point_stride, void *stream) {
    if (!obstacle || !edge || obstacle->n_scenes != edge->n_scenes || point_stride < 3 || (!d_xyz && obstacle->max_points > 0) ||
        (!d_edge_xyz && edge->max_points > 0))
        return AMK_ERR_INVALID_ARG;
    {
        amk::TimedLaunch tg(amk::KC_GRID, (hipStream_t)stream);
        BuildArgs a = build_args(obstacle, d_xyz, point_stride, (long long)obstacle->max_points * point_stride, d_counts);
        BuildArgs b = build_args(edge, d_edge_xyz, point_stride, (long long)edge->max_points * point_stride, d_edge_counts);
        const bool soa_a = build_writes_soa(obstacle, a), soa_b = build_writes_soa(edge, b);
        BuildArgs2 args{};
        args.t[0] = a; args.t[1] = b;
        hipLaunchKernelGGL(kd_build_kernel, dim3(obstacle->n_scenes, 2), dim3(kCompactThreads), 0, (hipStream_t)stream, args);
        obstacle->soa_valid = soa_a ? 1 : 0; edge->soa_valid = soa_b ? 1 : 0;
        for (amk_kd *kd : {obstacle, edge}) {
            kd->ex_valid = 0;
            kd->async_pending = 1;
        }
    }
    AMK_HIP(hipGetLastError());
    for (amk_kd *kd : {obstacle, edge})
        if (kd->tie_order) {
            const int st = exact_build(kd, (hipStream_t)stream);
            if (st != AMK_OK) return st;
        }
    return AMK_OK;
}

// Internal (csrc/pipeline.hip, gang > 1): the frames of a gang in ONE launch -- frame f's two clouds are built into scenes
// [f * frame_scenes, (f + 1) * frame_scenes) of the two handles (which hold n_frames * frame_scenes scenes); grid.y = (tree, frame).
}  // extern "C"
namespace amk {
int kd_build_gang(amk_kd *obstacle, amk_kd *edge, int n_frames, int frame_scenes, const float *const *d_xyz,
                  const int *const *d_counts, const float *const *d_edge_xyz, const int *const *d_edge_counts, int point_stride,
                  hipStream_t stream, const int *const *d_keep_if_zero) {
    if (!obstacle || !edge || n_frames < 1 || n_frames > AMK_PIPELINE_MAX_GANG || point_stride < 3 ||
        obstacle->n_scenes < n_frames * frame_scenes || edge->n_scenes != obstacle->n_scenes)
        return AMK_ERR_INVALID_ARG;
    {
        amk::TimedLaunch tg(amk::KC_GRID, stream);
        BuildArgs2 args{};
        for (int f = 0; f < n_frames; ++f) {
            const size_t so = (size_t)f * frame_scenes;
            args.t[2 * f] = build_args(obstacle, d_xyz[f], point_stride, (long long)obstacle->max_points * point_stride, d_counts[f], so);
            args.t[2 * f + 1] = build_args(edge, d_edge_xyz[f], point_stride, (long long)edge->max_points * point_stride, d_edge_counts[f], so);
            if (d_keep_if_zero) args.t[2 * f].keep_if_zero = args.t[2 * f + 1].keep_if_zero = d_keep_if_zero[f];
        }
        // (a scene that keeps its previous index keeps planes that may predate the tie-order mode: ensure_soa remakes them all)
        bool soa_o = !d_keep_if_zero, soa_e = !d_keep_if_zero;
        for (int f = 0; f < n_frames; ++f) {
            soa_o = soa_o && build_writes_soa(obstacle, args.t[2 * f], (size_t)f * frame_scenes);
            soa_e = soa_e && build_writes_soa(edge, args.t[2 * f + 1], (size_t)f * frame_scenes);
        }
        if (!soa_o) for (int f = 0; f < n_frames; ++f) args.t[2 * f].soa_x = args.t[2 * f].soa_y = args.t[2 * f].soa_z = nullptr;
        if (!soa_e) for (int f = 0; f < n_frames; ++f) args.t[2 * f + 1].soa_x = args.t[2 * f + 1].soa_y = args.t[2 * f + 1].soa_z = nullptr;
        hipLaunchKernelGGL(kd_build_kernel, dim3(frame_scenes, 2 * n_frames), dim3(kCompactThreads), 0, stream, args);
        obstacle->soa_valid = soa_o ? 1 : 0; edge->soa_valid = soa_e ? 1 : 0;
        for (amk_kd *kd : {obstacle, edge}) {
            kd->ex_valid = 0;
            kd->async_pending = 1;
        }
    }
    AMK_HIP(hipGetLastError());
    for (amk_kd *kd : {obstacle, edge})
        if (kd->tie_order) {
            const int st = exact_build(kd, stream);
            if (st != AMK_OK) return st;
        }
    return AMK_OK;
}
}  // namespace amk
extern "C" {

// internal (tests): number of nodes of every scene's reference-shaped tree (-1: not available), after synchronising
int amk__kd_exact_nodes(amk_kd *kd, int *h_nodes) {
    if (!kd || !h_nodes || !kd->ex_nn.p) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipDeviceSynchronize());
    AMK_HIP(hipMemcpy(h_nodes, kd->ex_nn.p, sizeof(int) * kd->n_scenes, hipMemcpyDeviceToHost));
    return AMK_OK;
}

#ifdef AMK_EXACT_TRACE
// diagnostics build only (tools/experiments/exact_phase_clocks.py): the first n words of the scene-0 list buffer, where the
// lower-level build kernel leaves its per-wavefront phase clocks
int amk__kd_exact_trace(amk_kd *kd, unsigned *h, int n) {
    if (!kd || !h || !kd->ex_sa.p) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipDeviceSynchronize());
    AMK_HIP(hipMemcpy(h, kd->ex_sa.p, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    return AMK_OK;
}
#endif

// depth of every scene's reference-shaped tree: children are created after their parent (ids grow downwards), so one pass in
// id order carries the depths; one lane per scene, the build's list scratch (sa) holds depth[node] -- a diagnostic call
static __global__ __launch_bounds__(64) void kd_exact_status_kernel(int S, const int *__restrict__ n_nodes, const int *__restrict__ feat,
                                                             const int *__restrict__ child, unsigned *__restrict__ scratch,
                                                             int max_nodes, int cap, int *__restrict__ status) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= S) return;
    const int nn = n_nodes[s];
    if (nn < 0) { status[s] = AMK_EXACT_GAVE_UP; return; }
    const int *f = feat + (size_t)s * max_nodes, *c = child + (size_t)s * max_nodes;
    unsigned *d = scratch + (size_t)s * cap;
    unsigned deepest = 0;
    if (nn > 0) d[0] = 0;
    for (int id = 0; id < nn; ++id) {
        const unsigned dep = d[id];
        if (f[id] >= 0) { d[c[id]] = dep + 1; d[c[id] + 1] = dep + 1; }   // an internal node: the traversal pushes one frame here
        else deepest = dep > deepest ? dep : deepest;                      // a leaf at depth dep is reached with dep frames on the stack
    }
    status[s] = deepest > (unsigned)amk::kExactMaxDepth ? AMK_EXACT_TOO_DEEP : AMK_EXACT_IN_USE;
}

// Which index answers a handle's searches, per scene (header).  Stream-ordered.
int amk_kd_exact_status(amk_kd *kd, int *d_status, void *stream) {
    if (!kd || !d_status) return AMK_ERR_INVALID_ARG;
    if (!kd->tie_order || !kd->ex_valid || !kd->ex_nn.p) {   // the bucketed index answers everything: not a fallback
        AMK_HIP(hipMemsetAsync(d_status, 0xff, sizeof(int) * kd->n_scenes, (hipStream_t)stream));   // AMK_EXACT_OFF == -1
        return AMK_OK;
    }
    hipLaunchKernelGGL(kd_exact_status_kernel, dim3((kd->n_scenes + 63) / 64), dim3(64), 0, (hipStream_t)stream, kd->n_scenes,
                       kd->ex_nn.p, kd->ex_feat.p, kd->ex_child.p, kd->ex_sa.p, kd->ex_max_nodes, kd->cap, d_status);
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_kd_exact_status_host(amk_kd *kd, int *h_status) {   // synchronises
    if (!kd || !h_status) return AMK_ERR_INVALID_ARG;
    amk::DevBuf<int> d;
    AMK_HIP(d.alloc(kd->n_scenes));
    const int st = amk_kd_exact_status(kd, d.p, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipDeviceSynchronize());
    AMK_HIP(hipMemcpy(h_status, d.p, sizeof(int) * kd->n_scenes, hipMemcpyDeviceToHost));
    return AMK_OK;
}

int amk_kd_set_tie_order(amk_kd *kd, int mode) {
    if (!kd) return AMK_ERR_INVALID_ARG;
    if (mode != AMK_TIES_LOWEST_INDEX && mode != AMK_TIES_NANOFLANN) return AMK_ERR_UNSUPPORTED;
    kd->tie_order = mode;
    return AMK_OK;
}

int amk_kd_sizes(amk_kd *kd, int *h_sizes, void *stream) {
    if (!kd || !h_sizes) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipMemcpyAsync(h_sizes, kd->size.p, sizeof(int) * kd->n_scenes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    AMK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return AMK_OK;
}

int amk_kd_search(amk_kd *kd, const double *d_queries, int n_queries, int k, int *d_indices, double *d_sqdist,
                  float *d_pts, int *d_counts, void *stream) {
    if (!kd || !d_queries || n_queries <= 0 || k <= 0) return AMK_ERR_INVALID_ARG;
    if (k > AMK_MAX_K || n_queries > AMK_MAX_QUERIES) return AMK_ERR_UNSUPPORTED;
    if (kd->mode == 0) {
        const amk::GridPtrs gpt{kd->gpt.p, kd->cell_start.p, kd->gparams.p, kd->cap, kd->ntiles};
        const int blocks = (kd->n_scenes + 7) / 8 * 8 * ((n_queries + 3) / 4);
        hipLaunchKernelGGL(kd_grid_search_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gpt, kd->size.p,
                           kd->n_scenes, d_queries, n_queries, k, d_indices, d_sqdist, d_pts, d_counts);
        AMK_HIP(hipGetLastError());
        if (kd->tie_order && kd->ex_valid) {  // nanoflann's own traversal where its tree is available (and current)
            const size_t rows = (size_t)kd->n_scenes * n_queries;
            hipLaunchKernelGGL(kd_exact_search_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                               exact_ptrs(kd), kd->size.p, kd->n_scenes, d_queries, n_queries, k, d_indices, d_sqdist, d_pts,
                               d_counts);
            AMK_HIP(hipGetLastError());
        }
        return AMK_OK;
    }
    {   // the streaming scan reads the index-ordered planes
        const int st = ensure_soa(kd, (hipStream_t)stream);
        if (st != AMK_OK) return st;
    }
    int qpw, groups, wpb;
    amk::scan_geometry(n_queries, qpw, groups, wpb);
    const int bps = (groups + wpb - 1) / wpb;
    const int blocks = (kd->n_scenes + 7) / 8 * 8 * bps;
#define AMK_LAUNCH_SCAN(Q)                                                                                      \
    hipLaunchKernelGGL(kd_scan_kernel<Q>, dim3(blocks), dim3(wpb * kWave), amk::scan_lds_bytes<Q>(wpb),          \
                       (hipStream_t)stream, kd->x.p, kd->y.p, kd->z.p, kd->cap, kd->size.p, kd->pmax.p,          \
                       kd->n_scenes, d_queries, n_queries, k, d_indices, d_sqdist, d_pts, d_counts)
    if (qpw == 1) AMK_LAUNCH_SCAN(1);
    else AMK_LAUNCH_SCAN(5);
#undef AMK_LAUNCH_SCAN
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_kd_tie_flags(amk_kd *kd, const double *d_queries, int query_stride, int n_queries, int k, int *d_tie_flags,
                     void *stream) {
    if (!kd || !d_queries || !d_tie_flags || n_queries <= 0 || k <= 0 || query_stride < 3) return AMK_ERR_INVALID_ARG;
    if (k + 1 > AMK_MAX_K || n_queries > AMK_MAX_QUERIES) return AMK_ERR_UNSUPPORTED;
    const amk::GridPtrs gpt{kd->gpt.p, kd->cell_start.p, kd->gparams.p, kd->cap, kd->ntiles};
    const int blocks = (kd->n_scenes + 7) / 8 * 8 * ((n_queries + 3) / 4);
    hipLaunchKernelGGL(kd_tie_flags_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gpt, kd->size.p, kd->n_scenes,
                       d_queries, query_stride, n_queries, k, d_tie_flags);
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_kd_build_host(amk_kd *kd, const float *h_xyz, int point_stride, long long scene_stride, const int *h_counts) {
    if (!kd || (!h_xyz && kd->max_points > 0) || point_stride < 3) return AMK_ERR_INVALID_ARG;
    const long long min_stride = (long long)kd->max_points * point_stride;
    if (scene_stride < min_stride && kd->n_scenes > 1) return AMK_ERR_INVALID_ARG;
    const size_t tot = (size_t)(kd->n_scenes - 1) * scene_stride + (size_t)min_stride;
    if (kd->stage_xyz.n < tot) AMK_HIP(kd->stage_xyz.alloc(tot > 0 ? tot : 1));
    if (h_counts) {  // only the points every scene holds (the caller's buffer need not extend to the capacity)
        for (int s = 0; s < kd->n_scenes; ++s) {
            if (h_counts[s] < 0 || h_counts[s] > kd->max_points) return AMK_ERR_INVALID_ARG;
            const size_t nf = (size_t)h_counts[s] * point_stride;
            if (nf) AMK_HIP(hipMemcpy(kd->stage_xyz.p + (size_t)s * scene_stride, h_xyz + (size_t)s * scene_stride,
                                      nf * sizeof(float), hipMemcpyHostToDevice));
        }
    } else if (tot) {
        AMK_HIP(hipMemcpy(kd->stage_xyz.p, h_xyz, tot * sizeof(float), hipMemcpyHostToDevice));
    }
    const int *d_counts = nullptr;
    if (h_counts) {
        if (kd->stage_counts.n < (size_t)kd->n_scenes) AMK_HIP(kd->stage_counts.alloc(kd->n_scenes));
        AMK_HIP(hipMemcpy(kd->stage_counts.p, h_counts, sizeof(int) * kd->n_scenes, hipMemcpyHostToDevice));
        d_counts = kd->stage_counts.p;
    }
    int st = amk_kd_build(kd, kd->stage_xyz.p, point_stride, scene_stride, d_counts, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipDeviceSynchronize());
    kd->async_pending = 0;
    return AMK_OK;
}

int amk_kd_search_host(amk_kd *kd, const double *h_queries, int n_queries, int k, int *h_indices, double *h_sqdist,
                       float *h_pts, int *h_counts) {
    if (!kd || !h_queries || n_queries <= 0 || k <= 0) return AMK_ERR_INVALID_ARG;
    if (k > AMK_MAX_K || n_queries > AMK_MAX_QUERIES) return AMK_ERR_UNSUPPORTED;
    const size_t rows = (size_t)kd->n_scenes * n_queries;
    // one device block and one pinned host block: [queries | sqdist | indices | counts | pts]
    const size_t o_q = 0, o_d2 = o_q + rows * 3 * sizeof(double), o_idx = o_d2 + rows * k * sizeof(double),
                 o_cnt = o_idx + rows * k * sizeof(int), o_pts = o_cnt + rows * sizeof(int),
                 total = o_pts + rows * k * 3 * sizeof(float);
    if (!kd->hstream) AMK_HIP(hipStreamCreateWithFlags(&kd->hstream, hipStreamNonBlocking));
    if (kd->stage_out.n < total) AMK_HIP(kd->stage_out.alloc(total));
    if (kd->hpin_bytes < total) {
        if (kd->hpin) (void)hipHostFree(kd->hpin);
        kd->hpin = nullptr;
        kd->hpin_bytes = 0;
        AMK_HIP(hipHostMalloc(&kd->hpin, total, hipHostMallocDefault));
        kd->hpin_bytes = total;
    }
    if (kd->async_pending) {  // a build / sweep enqueued on some other stream: order behind it once
        AMK_HIP(hipDeviceSynchronize());
        kd->async_pending = 0;
    }
    unsigned char *hp = static_cast<unsigned char *>(kd->hpin), *dp = kd->stage_out.p;
    memcpy(hp + o_q, h_queries, rows * 3 * sizeof(double));
    AMK_HIP(hipMemcpyAsync(dp + o_q, hp + o_q, rows * 3 * sizeof(double), hipMemcpyHostToDevice, kd->hstream));
    int st = amk_kd_search(kd, reinterpret_cast<const double *>(dp + o_q), n_queries, k, reinterpret_cast<int *>(dp + o_idx),
                           reinterpret_cast<double *>(dp + o_d2), reinterpret_cast<float *>(dp + o_pts),
                           reinterpret_cast<int *>(dp + o_cnt), kd->hstream);
    if (st != AMK_OK) return st;
    AMK_HIP(hipMemcpyAsync(hp + o_d2, dp + o_d2, total - o_d2, hipMemcpyDeviceToHost, kd->hstream));
    AMK_HIP(hipStreamSynchronize(kd->hstream));
    if (h_indices) memcpy(h_indices, hp + o_idx, rows * k * sizeof(int));
    if (h_sqdist) memcpy(h_sqdist, hp + o_d2, rows * k * sizeof(double));
    if (h_pts) memcpy(h_pts, hp + o_pts, rows * k * 3 * sizeof(float));
    if (h_counts) memcpy(h_counts, hp + o_cnt, rows * sizeof(int));
    return AMK_OK;
}

}  // extern "C"
