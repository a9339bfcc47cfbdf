// Internal helpers shared by the HIP translation units of libavoid_mpc_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/avoid_mpc_amd.h"

namespace amk {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return AMK_ERR_HIP;
}

#define AMK_HIP(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return amk::hip_fail(_e); \
    } while (0)

constexpr int kWave = 64;  // CDNA4 wavefront

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// A device buffer that frees itself (handles own their storage; no allocation on the hot path).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

// value of lane `src` (wave-uniform) broadcast to every lane through scalar registers
__device__ __forceinline__ double readlane_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; internal (amk__timing_* are not part of the C ABI in include/avoid_mpc_amd.h).
enum KernelClass { KC_COMPACT = 0, KC_SCAN_OBS, KC_SCAN_EDGE, KC_PLAN, KC_PACK, KC_SOLVE, KC_BEGIN, KC_GRID, KC_COUNT };
struct Timing;
Timing &timing();
struct TimedLaunch {  // RAII: records an event before and after the enclosed launch when enabled
    TimedLaunch(int kclass, hipStream_t stream);
    ~TimedLaunch();
    int slot;
    hipStream_t stream;
};

}  // namespace amk

struct amk_kd;
namespace amk {
// kd_index.hip: the frames of a pipeline gang built by one launch (see there)
int kd_build_gang(amk_kd *obstacle, amk_kd *edge, int n_frames, int frame_scenes, const float *const *d_xyz,
                  const int *const *d_counts, const float *const *d_edge_xyz, const int *const *d_edge_counts, int point_stride,
                  hipStream_t stream, const int *const *d_keep_if_zero = nullptr);
// kd_index.hip, for the keyframe map (kfmap.hip): builds and sweeps on a POOL handle (scene = physical slot * S + scene)
int kd_build_mapped(amk_kd *obs_pool, amk_kd *edge_pool, int n_in, const float *d_xyz, const int *d_counts, const float *d_edge_xyz,
                    const int *d_edge_counts, int point_stride, const int *d_out_scene, hipStream_t stream);
int kd_pool_reserve(amk_kd *pool, int n_rows);
int kd_build_mapped_gang(amk_kd *obs_pool, amk_kd *edge_pool, int n_frames, int frame_scenes, const float *const *d_xyz,
                         const int *const *d_counts, const float *const *d_edge_xyz, const int *const *d_edge_counts, int point_stride,
                         const int *d_out_scene, hipStream_t stream);
int kd_sweep_mapped(amk_kd *pool, int n_rows, const int *d_kf_list, const int *d_cur_list, double th_dist, int th_count,
                    int *d_outliers, int *d_rebuilt, hipStream_t stream);
}  // namespace amk

// ------------------------------------------------------------------------------------------------
// handle layouts (shared between kd_index.hip, mpc_solve.hip and step.hip)
// ------------------------------------------------------------------------------------------------
struct amk_kd {
    int n_scenes = 0;
    int max_points = 0;
    int cap = 0;  // per-scene SoA capacity, multiple of 256 with >= 1024 floats of NaN padding
    amk::DevBuf<float> x, y, z;  // [S][cap] filtered points (order preserved), NaN padded -- made ON DEMAND from the
                                 // bucket records (ensure_soa, kd_index.hip): the build does not write them
    int soa_valid = 0;           // host flag: x/y/z match the current index
    amk::DevBuf<int> size;       // [S] cloud.pts.size() after the NaN-x filter
    amk::DevBuf<float> pmax;     // [S] max |coordinate| of the kept points (bounds the fp32 pre-filter error)
    // bucketed index (kd_grid.h): bucket-contiguous copy of the points, their cloud indices, bucket starts
    amk::DevBuf<float4> gpt;         // [S][cap]  (x, y, z, cloud index)
    amk::DevBuf<float> bbox;         // [S][6]    min xyz, max xyz of the finite points
    amk::DevBuf<int> cell_start;     // [S][ntiles][kGridMaxCells + 2]  bucket starts of every tile of the record array (kd_grid.h)
    int ntiles = 1;                  // ceil(max_points / kTilePoints)
    amk::DevBuf<double> gparams;     // [S][8]
    int mode = 0;                    // 0: grid search (default), 1: streaming scan (cross-check)
    // opt-in "nanoflann tie order" (amk_kd_set_tie_order, kd_exact.h): the reference's own tree, built beside the bucketed index
    int tie_order = 0;
    int ex_valid = 0;   // host flag: the exact tree was built from the cloud the bucketed index currently holds
    int ex_max_nodes = 0;
    amk::DevBuf<unsigned> ex_vind, ex_left, ex_right, ex_sa, ex_sb;
    amk::DevBuf<float> ex_pc;   // [3][S][cap] the coordinates in vAcc_ order (leaves are contiguous runs)
    amk::DevBuf<int> ex_feat, ex_child, ex_nn;
    amk::DevBuf<double> ex_low, ex_high, ex_nbbox, ex_root;
    amk::DevBuf<unsigned char> flags; // [S][cap] keyframe sweep: 1 = outlier
    amk::DevBuf<int> sweep_cnt;       // [S][2]   {outliers, rebuilt}
    // the keyframe map's pool only (kd_sweep_mapped): the sweep's target, per sweep ROW -- the current frame's points once more, sorted
    // into a fine hashed grid (cells of 2.5 th), rebuilt before every sweep
    // Two generations (this sweep's and the one before: a row's newest keyframe is usually the frame it swept against last period,
    // and its points are then taken in that grid's order -- the lanes of a wavefront ask for the same few buckets)
    amk::DevBuf<float4> sw_gpt;       // [2][rows][cap]         records, bucket by bucket
    amk::DevBuf<int> sw_cs;           // [2][rows][buckets + 1] bucket starts; the last entry = points with finite coordinates
    amk::DevBuf<int> sw_src;          // [2 + 1][rows]          pool scene the row's grid was built from, -1: none (row 2: always -1)
    int sw_rows = 0, sw_flip = 0;
    double sw_inv_h = 0.0;            // the lattice of the grids held (1 / cell edge)
    // staging for the *_host conveniences: a private stream and one pinned host block, so that a single-query
    // SearchForNearest costs one small H2D copy, one launch, one D2H copy and a wait on THIS stream only (a device-wide
    // synchronisation would stall every other stream of the process: a ROS node calls this ~100 times per control period)
    hipStream_t hstream = nullptr;
    void *hpin = nullptr;
    size_t hpin_bytes = 0;
    amk::DevBuf<unsigned char> stage_out;
    int async_pending = 0;  // a stream-ordered build / sweep was enqueued since the last synchronisation
    amk::DevBuf<float> stage_xyz;
    amk::DevBuf<int> stage_counts;
    amk::DevBuf<double> stage_q, stage_d2;
    amk::DevBuf<int> stage_idx, stage_cnt;
    amk::DevBuf<float> stage_pts;
};
