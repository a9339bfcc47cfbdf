// amk_kfmap: FrameKDMap's keyframe list for a BATCH of scenes, on the device (gfx950).
//
// The reference keeps, per robot, the current frame and a deque of keyframes (AM/src/FrameKDMap.cpp; its keyframe thread is on
// by default: max_frame_count = 100, only_trust_vel = false, :29-32), and every QueryNearest / GetNearestDistance of a control
// step runs over mVecQueryVector = [current, every keyframe but the newest] (:64-74, 322-427).  This file is that map for S
// scenes at once, with no host round trip between a frame's arrival and the step that uses the map:
//
//   amk_kfmap_add_vertex   FrameKDMap::AddVertex after ProcessDepth (:39-51): two fresh indices per scene whose frame is not
//                          empty, mCurFrame.Twc, mbNeedProcessPtCloud
//   amk_kfmap_update       one pass of KeyframeThreadWorker's body per scene that got a frame (:443-486): first keyframe;
//                          pop while the list is too long or the drone has passed the oldest one (DroneBehindPts, :233-252);
//                          n x 1-NN sweep of the newest keyframe against the current frame, rebuild from the outliers,
//                          InsertKeyFrame
//   amk_kfmap_step         the TASK branch of Step over the map (step_frames.hip: PtIsInFrame fast path, per-frame merge)
//
// Storage.  A keyframe shares the current frame's trees in the reference (InsertKeyFrame copies shared_ptrs, :428-431) and a
// later AddVertex gives mCurFrame NEW trees; here every scene owns P = max_frame_count + 2 physical slots in two POOL handles
// (obstacle, edge: scene index = slot * S + scene): the current frame is BUILT into a slot no keyframe holds (no copy on
// insertion), the deque is a list of slot numbers, a popped keyframe's slot becomes free.  The deque never holds more than
// max_frame_count + 1 keyframes (the pop loop runs before the insertion), so a free slot always exists.  Scenes are
// independent: their deques differ in length, the step sees a per-scene frame map (absent frames = empty clouds).
#include "kd_grid.h"
#include "mpc_handle.h"

#include <cstdio>
#include <vector>

namespace amk {
int step_batch_map(amk_kd *obs_pool, amk_kd *edge_pool, int n_frames, const int *d_fmap, const double *d_Twc,
                   const amk_frame_camera *cam, amk_mpc *mpc, const amk_step_params *prm, const double *d_state_quad,
                   const double *d_pos_x, double *d_ref_path, double *d_u, double *d_x0array, int *d_flags, hipStream_t stream);
}

struct amk_kfmap {
    int S = 0, P = 0, F = 0;   // scenes, physical slots per scene, frames of the query vector (1 + max_frame_count)
    amk_kfmap_params prm{};
    double Tbc_inv[16];
    amk_kd *obs = nullptr, *edge = nullptr;   // the pools: P * S scenes each
    amk::DevBuf<int> cur_slot;   // [S]    slot of mCurFrame's trees, -1 before the first frame
    amk::DevBuf<int> kf_n;       // [S]    mKeyFrameMap.size()
    amk::DevBuf<int> kf_slots;   // [S][P] mKeyFrameMap, oldest first
    amk::DevBuf<int> need;       // [S]    mbNeedProcessPtCloud
    amk::DevBuf<double> Twc;     // [S][16] mCurFrame.Twc (identity before the first frame)
    amk::DevBuf<double> tinv;    // [16]   Tbc^-1
    amk::DevBuf<int> out_scene;  // [S]    scratch of add_vertex: pool scene the new frame is built into, -1: no new frame
    amk::DevBuf<int> kf_list, cur_list, outliers, rebuilt;   // [S] scratch of update (rows of the sweep)
    amk::DevBuf<int> fmap;       // [F][S] query vector: pool scene of frame f, -1 absent
};

namespace {
using namespace amk;

__global__ __launch_bounds__(256) void kf_init_kernel(int S, int P, int *__restrict__ cur_slot, int *__restrict__ kf_n,
                                                      int *__restrict__ kf_slots, int *__restrict__ need, double *__restrict__ Twc,
                                                      int *__restrict__ fmap, int F, int first, int n) {
    const int s = first + blockIdx.x * 256 + threadIdx.x;
    if (s >= first + n) return;
    cur_slot[s] = -1; kf_n[s] = 0; need[s] = 0;
    for (int i = 0; i < P; ++i) kf_slots[(size_t)s * P + i] = -1;
    for (int e = 0; e < 16; ++e) Twc[(size_t)s * 16 + e] = (e % 5 == 0) ? 1.0 : 0.0;
    for (int f = 0; f < F; ++f) fmap[(size_t)f * S + s] = -1;
}

// AddVertex, bookkeeping half: which physical slot the scene's new current frame is built into.  The old current slot is
// reused unless the deque holds it (then the trees live on as a keyframe and the lowest free slot is taken).
__device__ __forceinline__ void kf_alloc_scene(int S, int P, int s, bool has_frame, const double *__restrict__ Twc_in16,
                                               int *__restrict__ cur_slot, const int *__restrict__ kf_n, const int *__restrict__ kf_slots,
                                               int *__restrict__ need, double *__restrict__ Twc, int *__restrict__ out_scene_i,
                                               int *__restrict__ fmap) {
    if (!has_frame) {   // ProcessDepth left no obstacle point: AddVertex returns before anything changes (:39-41)
        *out_scene_i = -1;
        return;
    }
    const int *dq = kf_slots + (size_t)s * P;
    const int nk = kf_n[s], cur = cur_slot[s];
    bool cur_held = cur < 0;
    for (int j = 0; j < nk; ++j) cur_held = cur_held || dq[j] == cur;
    int slot = cur;
    if (cur_held) {
        for (slot = 0; slot < P; ++slot) {
            bool used = false;
            for (int j = 0; j < nk; ++j) used = used || dq[j] == slot;
            if (!used) break;
        }
    }
    cur_slot[s] = slot;           // (slot < P: the deque holds at most P - 1 slots)
    *out_scene_i = slot * S + s;
    fmap[s] = slot * S + s;       // SetCurPtCloud -> UpdateQueryVector (:53-58): the query vector's first frame is the new one at once
                                  // (the keyframe rows do not change here: the deque is the worker's, amk_kfmap_update)
    need[s] = 1;                  // mbNeedProcessPtCloud = true (:51)
    for (int e = 0; e < 16; ++e) Twc[(size_t)s * 16 + e] = Twc_in16[e];   // mCurFrame.Twc = mat4Twb * mParamTbc (:50)
}
__global__ __launch_bounds__(256) void kf_alloc_kernel(int S, int P, int first, int n, const int *__restrict__ counts,
                                                       const double *__restrict__ Twc_in, int *__restrict__ cur_slot,
                                                       const int *__restrict__ kf_n, const int *__restrict__ kf_slots,
                                                       int *__restrict__ need, double *__restrict__ Twc, int *__restrict__ out_scene,
                                                       int *__restrict__ fmap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    kf_alloc_scene(S, P, first + i, !(counts && counts[i] <= 0), Twc_in + (size_t)i * 16, cur_slot, kf_n, kf_slots, need, Twc, out_scene + i, fmap);
}
// the same for the G frames of a pipeline gang at once: frame f holds the map's scenes [f * frame_scenes, (f + 1) * frame_scenes)
struct KfGangIn { const int *counts[AMK_PIPELINE_MAX_GANG]; const double *twc[AMK_PIPELINE_MAX_GANG]; };
__global__ __launch_bounds__(256) void kf_alloc_gang_kernel(int S, int P, int n_frames, int frame_scenes, KfGangIn in,
                                                            int *__restrict__ cur_slot, const int *__restrict__ kf_n,
                                                            const int *__restrict__ kf_slots, int *__restrict__ need,
                                                            double *__restrict__ Twc, int *__restrict__ out_scene, int *__restrict__ fmap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_frames * frame_scenes) return;
    const int f = i / frame_scenes, j = i - f * frame_scenes;
    const int *counts = in.counts[f];
    kf_alloc_scene(S, P, i, !(counts && counts[j] <= 0), in.twc[f] + (size_t)j * 16, cur_slot, kf_n, kf_slots, need, Twc, out_scene + i, fmap);
}

// KeyframeThreadWorker's body up to the sweep (:443-462), one wavefront per scene.
__global__ __launch_bounds__(64) void kf_pop_kernel(int S, int P, GridPtrs pool, const int *__restrict__ pool_size,
                                                    const int *__restrict__ cur_slot, int *__restrict__ kf_n,
                                                    int *__restrict__ kf_slots, int *__restrict__ need,
                                                    const double *__restrict__ Twc, const double *__restrict__ tinv,
                                                    int max_frame_count, double depth_min, int *__restrict__ kf_list,
                                                    int *__restrict__ cur_list) {
#pragma clang fp contract(off)   // Twb = Twc * Tbc^-1 and ptb.x as the CPU restatement forms them (tests/_kfmap.py), sum in index order
    __shared__ GridWaveLds wl;
    const int s = blockIdx.x, lane = threadIdx.x;
    int *dq = kf_slots + (size_t)s * P;
    if (lane == 0) { kf_list[s] = -1; cur_list[s] = -1; }
    if (!need[s]) return;                 // no new frame since the last pass (:440-442)
    const int cur = cur_slot[s];
    int nk = kf_n[s];
    if (lane == 0) need[s] = 0;
    if (nk == 0) {                        // InsertKeyFrame (:446-449)
        if (lane == 0) { dq[0] = cur; kf_n[s] = 1; }
        return;
    }
    // the deque in registers for the duration of the pop loop: entry j lives in lane j % 64, register j / 64 (P <= 128)
    int d0 = lane < nk ? dq[lane] : -1, d1 = lane + 64 < nk ? dq[lane + 64] : -1;
    // the drone in the world: Twb = mCurFrame.Twc * mParamTbc.inverse() (:235-238); twb and the body x axis (first row of Rbw)
    const double *T = Twc + (size_t)s * 16;
    double twb[3], bx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a3 = 0.0, a0 = 0.0;
        for (int k = 0; k < 4; ++k) { a3 += T[4 * i + k] * tinv[4 * k + 3]; a0 += T[4 * i + k] * tinv[4 * k + 0]; }
        twb[i] = a3; bx[i] = a0;
    }
    const int nk0 = nk;
    while (nk > 0) {                      // :450-459
        bool pop = nk > max_frame_count;
        if (!pop) {                       // DroneBehindPts(mCurFrame.Twc, oldest) (:233-252)
            const int m = __shfl(d0, 0) * S + s;
            const int size = pool_size[m];
            const int cnt = size < 10 ? size : 10;
            bool behind = true;
            if (size > cnt) {             // SearchForNearest(cnt) yields nothing when the cloud holds exactly cnt points (kd_tree_two.h:119-124)
                const GridScene gs = pool.scene(m);
                double ld;
                int li, lpos;
                grid_knn(gs, twb[0], twb[1], twb[2], cnt, ld, li, lpos, &wl);
                bool bad = false;
                if (lane < cnt && li != kNoIndex) {
                    const float4 r = gs.pt[lpos];
                    const double ptbx = (bx[0] * ((double)r.x - twb[0]) + bx[1] * ((double)r.y - twb[1])) + bx[2] * ((double)r.z - twb[2]);
                    bad = ptbx <= depth_min;
                }
                behind = __ballot(bad) == 0ull;
            }
            pop = !behind;
        }
        if (!pop) break;
        // RemoveOldVertex: pop_front
        const int n0 = __shfl_down(d0, 1), n1 = __shfl_down(d1, 1), carry = __shfl(d1, 0);
        d0 = lane == 63 ? carry : n0;
        d1 = lane == 63 ? -1 : n1;
        --nk;
    }
    if (nk != nk0) {                      // (only lanes that hold an entry of the old deque write)
        if (lane < nk0) dq[lane] = lane < nk ? d0 : -1;
        if (lane + 64 < nk0) dq[lane + 64] = lane + 64 < nk ? d1 : -1;
        if (lane == 0) kf_n[s] = nk;
    }
    if (nk == 0) return;                  // :460-462
    const int back = nk - 1 < 64 ? __shfl(d0, nk - 1) : __shfl(d1, nk - 1 - 64);
    if (back == cur) return;              // (the newest keyframe IS the current frame: a tree swept against itself has no outlier)
    if (lane == 0) { kf_list[s] = back * S + s; cur_list[s] = cur * S + s; }
}

// InsertKeyFrame for the scenes whose sweep rebuilt the newest keyframe (:486), then UpdateQueryVector (:64-74)
__global__ __launch_bounds__(256) void kf_insert_kernel(int S, int P, int F, const int *__restrict__ cur_slot, int *__restrict__ kf_n,
                                                        int *__restrict__ kf_slots, const int *__restrict__ rebuilt,
                                                        int *__restrict__ fmap) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    int *dq = kf_slots + (size_t)s * P;
    int nk = kf_n[s];
    const int cur = cur_slot[s];
    if (rebuilt && rebuilt[s]) {
        dq[nk] = cur;
        kf_n[s] = ++nk;
    }
    fmap[s] = cur < 0 ? -1 : cur * S + s;                                                   // mCurFrame
    for (int f = 1; f < F; ++f) fmap[(size_t)f * S + s] = (f - 1 < nk - 1) ? dq[f - 1] * S + s : -1;   // all keyframes but the newest
}
}  // namespace

extern "C" {

// Device bytes the map's pools will hold once the first sweep has run (the formula of the header): per pool scene the bucket
// records (16 B per point of capacity), the tile directories, and for the obstacle pool the index-ordered planes (12 B) and the
// sweep's outlier flags (1 B); per SCENE (sweep row) two generations of the hashed grid of the current frame (16 B per point + bucket starts each).
int amk_kfmap_pool_bytes(int n_scenes, int max_points, int max_edge_points, int max_frame_count, long long *bytes_out) {
    if (!bytes_out || n_scenes <= 0 || max_points <= 0 || max_edge_points <= 0 || max_frame_count < 1) return AMK_ERR_INVALID_ARG;
    const long long P = max_frame_count + 2, S = n_scenes, F = max_frame_count + 1;
    auto cap = [](int mp) { return (long long)amk::round_up(mp, 256) + 1024; };
    auto index_bytes = [&](int mp) {   // amk_kd_create
        return 16 * cap(mp) + 4ll * amk::grid_tiles(mp) * (amk::kGridMaxCells + 2) + 8 * amk::kGridParamDoubles + 24 + 8;
    };
    long long b = P * S * (index_bytes(max_points) + index_bytes(max_edge_points));
    b += P * S * (12 + 1) * cap(max_points);                                      // x / y / z planes + flags (kd_sweep_mapped)
    int nb = 1024;
    while (nb < 16384 && nb < 2 * max_points) nb *= 2;
    b += 2 * S * (16 * cap(max_points) + 4ll * (nb + 1)) + 12 * S;                // the sweep's grids of the current frame: two generations per row (kd_sweep_mapped)
    b += S * (4 * (8 + P + F) + 8 * 16);                                          // the deques, lists and Twc
    *bytes_out = b;
    return AMK_OK;
}

int amk_kfmap_create(int n_scenes, int max_points, int max_edge_points, const amk_kfmap_params *prm, amk_kfmap **out) {
    if (!out || !prm || n_scenes <= 0 || max_points <= 0 || max_edge_points <= 0 || prm->keyframe_th_count < 1 ||
        !(prm->keyframe_th_dist >= 0.0))
        return AMK_ERR_INVALID_ARG;
    *out = nullptr;
    if (prm->max_frame_count < 1 || prm->max_frame_count + 1 > AMK_MAX_MAP_FRAMES) return AMK_ERR_UNSUPPORTED;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    {   // the pools are allocated eagerly at full capacity: say what they need before the allocator fails half-way with a bare
        // out-of-memory (max_frame_count = 100 with 512 scenes x 50 k points is 86 GB per map)
        long long need = 0;
        size_t free_b = 0, total_b = 0;
        if (amk_kfmap_pool_bytes(n_scenes, max_points, max_edge_points, prm->max_frame_count, &need) == AMK_OK &&
            hipMemGetInfo(&free_b, &total_b) == hipSuccess && (unsigned long long)need > (unsigned long long)free_b) {
            fprintf(stderr, "amk_kfmap_create: the pools of %d scenes x (max_frame_count %d + 2) slots x (%d + %d) points need %.2f GiB, "
                            "%.2f GiB are free on the device (amk_kfmap_pool_bytes)\n", n_scenes, prm->max_frame_count, max_points,
                    max_edge_points, need / 1073741824.0, free_b / 1073741824.0);
            return AMK_ERR_UNSUPPORTED;
        }
    }
    amk_kfmap *m = new amk_kfmap();
    m->S = n_scenes; m->P = prm->max_frame_count + 2; m->F = prm->max_frame_count + 1;
    m->prm = *prm;
    {   // rigid inverse of Tbc: [R' | -R' t]  (Eigen's general inverse is not restated: DESIGN.md section 10)
        const double *T = prm->Tbc;
        double *I = m->Tbc_inv;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) I[4 * i + j] = T[4 * j + i];
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += I[4 * i + k] * T[4 * k + 3];
            I[4 * i + 3] = -acc;
        }
        I[12] = I[13] = I[14] = 0.0; I[15] = 1.0;
    }
    int st = AMK_OK;
    const long long pool = (long long)m->P * n_scenes;
    if (pool > 0x3fffffff) st = AMK_ERR_UNSUPPORTED;
    if (st == AMK_OK) st = amk_kd_create((int)pool, max_points, &m->obs);
    if (st == AMK_OK) st = amk_kd_create((int)pool, max_edge_points, &m->edge);
    if (st == AMK_OK) st = amk::kd_pool_reserve(m->obs, n_scenes);   // planes, flags, the sweep's grids: everything now, so that memory runs out HERE
    hipError_t e = hipSuccess;
    const size_t S = n_scenes;
    if (st == AMK_OK &&
        ((e = m->cur_slot.alloc(S)) != hipSuccess || (e = m->kf_n.alloc(S)) != hipSuccess || (e = m->kf_slots.alloc(S * m->P)) != hipSuccess ||
         (e = m->need.alloc(S)) != hipSuccess || (e = m->Twc.alloc(S * 16)) != hipSuccess || (e = m->tinv.alloc(16)) != hipSuccess ||
         (e = m->out_scene.alloc(S)) != hipSuccess || (e = m->kf_list.alloc(S)) != hipSuccess || (e = m->cur_list.alloc(S)) != hipSuccess ||
         (e = m->outliers.alloc(S)) != hipSuccess || (e = m->rebuilt.alloc(S)) != hipSuccess || (e = m->fmap.alloc(S * m->F)) != hipSuccess ||
         (e = hipMemcpy(m->tinv.p, m->Tbc_inv, sizeof m->Tbc_inv, hipMemcpyHostToDevice)) != hipSuccess ||
         (e = hipMemset(m->outliers.p, 0, sizeof(int) * S)) != hipSuccess || (e = hipMemset(m->rebuilt.p, 0, sizeof(int) * S)) != hipSuccess))
        st = amk::hip_fail(e);
    if (st == AMK_OK) {
        hipLaunchKernelGGL(kf_init_kernel, dim3((n_scenes + 255) / 256), dim3(256), 0, nullptr, n_scenes, m->P, m->cur_slot.p, m->kf_n.p,
                           m->kf_slots.p, m->need.p, m->Twc.p, m->fmap.p, m->F, 0, n_scenes);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) st = amk::hip_fail(e);
    }
    if (st != AMK_OK) {
        amk_kfmap_destroy(m);
        return st;
    }
    *out = m;
    return AMK_OK;
}

int amk_kfmap_destroy(amk_kfmap *m) {
    if (!m) return AMK_ERR_INVALID_ARG;
    if (m->obs) amk_kd_destroy(m->obs);
    if (m->edge) amk_kd_destroy(m->edge);
    delete m;
    return AMK_OK;
}

// A new FrameKDMap for scenes [first_scene, first_scene + n_scenes): no current frame, no keyframe, Twc = identity (the state
// after the constructor, FrameKDMap.cpp:6-33) -- a robot that starts over.  Stream-ordered.
int amk_kfmap_reset(amk_kfmap *m, int first_scene, int n_scenes, void *stream) {
    if (!m || first_scene < 0 || n_scenes < 1 || first_scene + n_scenes > m->S) return AMK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kf_init_kernel, dim3((n_scenes + 255) / 256), dim3(256), 0, (hipStream_t)stream, m->S, m->P, m->cur_slot.p,
                       m->kf_n.p, m->kf_slots.p, m->need.p, m->Twc.p, m->fmap.p, m->F, first_scene, n_scenes);
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_kfmap_scenes(const amk_kfmap *m) { return m ? m->S : -1; }
int amk_kfmap_frames(const amk_kfmap *m) { return m ? m->F : -1; }
const double *amk_kfmap_twc(const amk_kfmap *m) { return m ? m->Twc.p : nullptr; }

int amk_kfmap_add_vertex(amk_kfmap *m, int first_scene, int n_scenes, const float *d_xyz, const int *d_counts,
                         const float *d_edge_xyz, const int *d_edge_counts, int point_stride, const double *d_Twc, void *stream_) {
    if (!m || !d_xyz || !d_edge_xyz || !d_Twc || first_scene < 0 || n_scenes < 1 || first_scene + n_scenes > m->S ||
        (point_stride != 3 && point_stride != 4))
        return AMK_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(kf_alloc_kernel, dim3((n_scenes + 255) / 256), dim3(256), 0, stream, m->S, m->P, first_scene, n_scenes, d_counts,
                       d_Twc, m->cur_slot.p, m->kf_n.p, m->kf_slots.p, m->need.p, m->Twc.p, m->out_scene.p + first_scene, m->fmap.p);
    AMK_HIP(hipGetLastError());
    return amk::kd_build_mapped(m->obs, m->edge, n_scenes, d_xyz, d_counts, d_edge_xyz, d_edge_counts, point_stride,
                                m->out_scene.p + first_scene, stream);
}

}  // extern "C"
namespace amk {
// AddVertex for the G frames of a pipeline gang (frame f = the map's scenes [f * frame_scenes, (f + 1) * frame_scenes)): one bookkeeping launch
// and one build launch for all of them (amk_kfmap_add_vertex per position: 2 G launches)
int kfmap_add_vertex_gang(amk_kfmap *m, int n_frames, int frame_scenes, const float *const *d_xyz, const int *const *d_counts,
                          const float *const *d_edge_xyz, const int *const *d_edge_counts, int point_stride, const double *const *d_Twc,
                          hipStream_t stream) {
    if (!m || n_frames < 1 || n_frames > AMK_PIPELINE_MAX_GANG || frame_scenes < 1 || n_frames * frame_scenes > m->S ||
        (point_stride != 3 && point_stride != 4))
        return AMK_ERR_INVALID_ARG;
    KfGangIn in{};
    for (int f = 0; f < n_frames; ++f) {
        if (!d_xyz[f] || !d_edge_xyz[f] || !d_Twc[f]) return AMK_ERR_INVALID_ARG;
        in.counts[f] = d_counts[f]; in.twc[f] = d_Twc[f];
    }
    const int n = n_frames * frame_scenes;
    hipLaunchKernelGGL(kf_alloc_gang_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, m->S, m->P, n_frames, frame_scenes, in, m->cur_slot.p,
                       m->kf_n.p, m->kf_slots.p, m->need.p, m->Twc.p, m->out_scene.p, m->fmap.p);
    AMK_HIP(hipGetLastError());
    return kd_build_mapped_gang(m->obs, m->edge, n_frames, frame_scenes, d_xyz, d_counts, d_edge_xyz, d_edge_counts, point_stride,
                                m->out_scene.p, stream);
}
}  // namespace amk
extern "C" {

int amk_kfmap_update(amk_kfmap *m, void *stream_) {
    if (!m) return AMK_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const GridPtrs pool{m->obs->gpt.p, m->obs->cell_start.p, m->obs->gparams.p, m->obs->cap, m->obs->ntiles};
    hipLaunchKernelGGL(kf_pop_kernel, dim3(m->S), dim3(64), 0, stream, m->S, m->P, pool, m->obs->size.p, m->cur_slot.p, m->kf_n.p,
                       m->kf_slots.p, m->need.p, m->Twc.p, m->tinv.p, m->prm.max_frame_count, m->prm.depth_min, m->kf_list.p,
                       m->cur_list.p);
    AMK_HIP(hipGetLastError());
    const int st = amk::kd_sweep_mapped(m->obs, m->S, m->kf_list.p, m->cur_list.p, m->prm.keyframe_th_dist, m->prm.keyframe_th_count,
                                        m->outliers.p, m->rebuilt.p, stream);
    if (st != AMK_OK) return st;
    hipLaunchKernelGGL(kf_insert_kernel, dim3((m->S + 255) / 256), dim3(256), 0, stream, m->S, m->P, m->F, m->cur_slot.p, m->kf_n.p,
                       m->kf_slots.p, m->rebuilt.p, m->fmap.p);
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_kfmap_step(amk_kfmap *m, const amk_frame_camera *cam, amk_mpc *mpc, const amk_step_params *prm, const double *d_state_quad,
                   const double *d_pos_x, double *d_ref_path, double *d_u, double *d_x0array, int *d_flags, void *stream_) {
    if (!m || !mpc || mpc->S != m->S) return AMK_ERR_INVALID_ARG;
    return amk::step_batch_map(m->obs, m->edge, m->F, m->fmap.p, cam ? m->Twc.p : nullptr, cam, mpc, prm, d_state_quad, d_pos_x,
                               d_ref_path, d_u, d_x0array, d_flags, (hipStream_t)stream_);
}

// Introspection (tests, diagnostics): per scene the number of keyframes, the number of frames of the query vector, the
// outliers of the last sweep (0 when none ran), and -- when h_frame_sizes is given, [S][n_frames] -- the obstacle-cloud size of
// every query frame (-1 beyond the scene's map).  Synchronises the device.
int amk_kfmap_state_host(amk_kfmap *m, int *h_n_keyframes, int *h_n_query_frames, int *h_last_outliers, int *h_frame_sizes) {
    if (!m) return AMK_ERR_INVALID_ARG;
    AMK_HIP(hipDeviceSynchronize());
    const size_t S = m->S;
    std::vector<int> nk(S), fm(S * m->F), out(S), sz((size_t)m->P * S);
    AMK_HIP(hipMemcpy(nk.data(), m->kf_n.p, sizeof(int) * S, hipMemcpyDeviceToHost));
    AMK_HIP(hipMemcpy(fm.data(), m->fmap.p, sizeof(int) * S * m->F, hipMemcpyDeviceToHost));
    AMK_HIP(hipMemcpy(out.data(), m->outliers.p, sizeof(int) * S, hipMemcpyDeviceToHost));
    AMK_HIP(hipMemcpy(sz.data(), m->obs->size.p, sizeof(int) * sz.size(), hipMemcpyDeviceToHost));
    for (size_t s = 0; s < S; ++s) {
        if (h_n_keyframes) h_n_keyframes[s] = nk[s];
        if (h_last_outliers) h_last_outliers[s] = out[s];
        int nq = 0;
        for (int f = 0; f < m->F; ++f) {
            const int p = fm[(size_t)f * S + s];
            if (p >= 0) ++nq;
            if (h_frame_sizes) h_frame_sizes[s * m->F + f] = p >= 0 ? sz[p] : -1;
        }
        if (h_n_query_frames) h_n_query_frames[s] = nq;
    }
    return AMK_OK;
}

}  // extern "C"
