// amk_pipeline: several independent control steps in flight on one GPU, behind the C ABI.
//
// A control step of this library (index builds of a fresh depth frame + amk_step_batch) is a chain of dependent launches
// whose dominant kernel is latency-bound: one stream fills ~1/8 of the chip.  Consecutive frames of
// a fleet of robots (or the batches of a sweep) are independent, so the throughput configuration keeps n_slots launches in
// flight, each on its own HIP stream with its own handles -- until round 2 that orchestration lived only in bench.py.
// A slot = {stream, obstacle index, edge index, MPC batch (warm start, workspace), input / output buffers, events}.
// The reference has one robot and one step in flight (AM/src/mpc_obstacle_avoidance_node.cpp:8,
// AvoidanceStateMachine.cpp:322-355); this is the batched counterpart of its per-frame sequence
// FrameKDMap::AddVertex (FrameKDMap.cpp:34-52) -> AvoidanceStateMachine::Step.
//
// Gang (amk_pipeline_config.gang = G > 1): G consecutively submitted frames share ONE set of launches -- the slot's handles
// hold G x n_scenes scenes, frame g occupies scenes [g S, (g + 1) S).  More than ~28 streams collapse on this runtime (32
// hardware queues), so a gang is how MORE scenes are kept in flight than 20-odd slots of one frame hold, in fewer and
// fuller launches: on the bench workload (256-scene frames) 10 slots x 4 frames reach 519 k scene-steps/s in the steady
// state against 466 k for 20 x 1, and 410-422 k against 378 k over a 20-frame burst (same box).
// A frame is only STAGED by submit() until its gang is full; wait() / drain() launch a partly filled gang.
#include "mpc_handle.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace amk {
int kfmap_add_vertex_gang(amk_kfmap *m, int n_frames, int frame_scenes, const float *const *d_xyz, const int *const *d_counts,
                          const float *const *d_edge_xyz, const int *const *d_edge_counts, int point_stride, const double *const *d_Twc,
                          hipStream_t stream);   // kfmap.hip
}

struct amk_pipeline {
    amk_pipeline_config cfg;
    struct Staged {  // a frame of the open gang
        const float *cloud, *edge;
        const int *cloud_counts, *edge_counts;
        const double *state_quad, *pos_x;   // read directly when gang == 1, gathered at launch otherwise
        const double *ref_path_init;
        double *u_out;
        int keep_warm_start;
        hipEvent_t input_ready;             // the caller's event (or null): the slot's stream waits for it before reading the inputs
        const double *odom;                 // TASK mode (amk_pipeline_frame.d_odom): prologue / epilogue on the device
        double odom_age;
        double *cmd_out;
        const void *depth;                  // raw depth image (amk_pipeline_frame.d_depth): ProcessDepth / BuildEdgeCloud on the device
        int depth_type, depth_rows, depth_cols;
        const double *Twb;
        // multi-frame map (gang 1 only): the caller's keyframe handles behind the slot's own indices of this frame
        std::vector<amk_kd *> kf_obstacle, kf_edge;
        const double *Twc_cur;
        bool has_camera;
        amk_frame_camera camera;
        const double *global_goal;          // TASK mode, task global_goal: mStateGlobalGoal per scene (or null: {0, 0, height})
    };
    struct Slot {
        hipStream_t stream = nullptr;
        std::vector<hipEvent_t> done;   // ring of queue_depth events: done[k % depth] marks the end of the slot's k-th launch
        long long count = 0;            // gang launches issued on this slot
        long long waited = 0;           // launches known to have finished
        amk_kd *obstacle = nullptr, *edge = nullptr;
        amk_mpc *mpc = nullptr;
        amk::DevBuf<double> ref_path, u, x0array, state_quad, pos_x;   // [G S][...]: the gang's contiguous inputs / outputs
        amk::DevBuf<int> flags;
        std::vector<Staged> open;       // frames staged since the last launch (< gang)
        int point_stride = 3;
        // frames that start at the depth image: the slot's own clouds + counts, and mCurFrame.Twc of every scene (persistent)
        amk::DevBuf<float> dcloud, dedge;      // [G S][max_points][3], [G S][max_edge_points][3]
        amk::DevBuf<int> dcount, decount;      // [G S]
        amk::DevBuf<double> Twc;               // [G S][16]
        bool ref_inited[AMK_PIPELINE_MAX_GANG] = {};   // TASK mode: position g holds an mRefPath (a frame with d_ref_path_init ran
                                        // there, and no launch has failed half-way since)
        amk_kfmap *map = nullptr;       // amk_pipeline_config.keyframes.max_frame_count > 0: the slot's keyframe map (gang x n_scenes scenes)
        int fail_status = AMK_OK;       // the slot's newest launch failed half-way with this status: its frames were dropped, their
                                        // results are undefined; reported by that submit() and by wait() / drain() until the next launch
    };
    int depth = 1, gang = 1;
    std::vector<Slot> slots;
    int next = 0;
    long long submitted = 0;
    int inject_failure = 0;             // tests only (amk__pipeline_inject_failure): the next launch fails at this stage
};

namespace {
// The small inputs of the frames of a gang into the slot's contiguous buffers, and their fresh warm starts, in ONE launch
// (as 3 copies + 1 memset per frame they were 16 tiny dispatches in front of every launch of a gang of 4, each of which
// queued behind whatever the other slots were dispatching: 7 % of the summed kernel time with 10 launches in flight).
struct GatherArgs {
    const double *sq[AMK_PIPELINE_MAX_GANG], *px[AMK_PIPELINE_MAX_GANG], *ref[AMK_PIPELINE_MAX_GANG];
    int keep_warm_start[AMK_PIPELINE_MAX_GANG];
    double *sq_dst, *px_dst, *ref_dst, *w0;   // [G][n_*]; sq_dst / px_dst NULL: not copied (gang 1 reads the caller's)
    int n_sq, n_px, n_ref, n_w0;              // doubles per frame
};
__global__ __launch_bounds__(256) void pipeline_gather_kernel(const GatherArgs a) {
    const int g = blockIdx.y;
    const int stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    if (a.sq_dst && a.sq[g]) {
        for (int i = t0; i < a.n_sq; i += stride) a.sq_dst[(size_t)g * a.n_sq + i] = a.sq[g][i];
        for (int i = t0; i < a.n_px; i += stride) a.px_dst[(size_t)g * a.n_px + i] = a.px[g][i];
    }
    if (a.ref[g])   // (a TASK-mode frame's path is the prologue kernel's business)
        for (int i = t0; i < a.n_ref; i += stride) a.ref_dst[(size_t)g * a.n_ref + i] = a.ref[g][i];
    if (!a.keep_warm_start[g])
        for (int i = t0; i < a.n_w0; i += stride) a.w0[(size_t)g * a.n_w0 + i] = 0.0;
}
// ... and the controls of every frame to where its caller wants them; for TASK-mode frames also what the node publishes:
// PubCmd(u) when isSafety, else PubSlowDownCmd (AvoidanceStateMachine.cpp:345-350,369-397)
struct ScatterArgs {
    double *dst[AMK_PIPELINE_MAX_GANG];   // NULL: stays in the slot's buffer only
    double *cmd[AMK_PIPELINE_MAX_GANG];   // NULL: no command wanted
    const double *odom[AMK_PIPELINE_MAX_GANG];
    const double *u;
    const int *flags;
    int n_u, S;
    double kp, kd, a_max_xy, a_max_z;
};
__global__ __launch_bounds__(256) void pipeline_scatter_kernel(const ScatterArgs a) {
#pragma clang fp contract(off)   // the command is compared bit for bit with the host twin (avoid_mpc_amd/flight.py: command)
    const int g = blockIdx.y;
    if (a.dst[g] && a.dst[g] != a.u + (size_t)g * a.n_u)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n_u; i += gridDim.x * 256) a.dst[g][i] = a.u[(size_t)g * a.n_u + i];
    if (a.cmd[g]) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.S * 3; i += gridDim.x * 256) {
            const int s = i / 3, c = i - 3 * s;
            const size_t gs = (size_t)g * a.S + s;
            double v;
            if (a.flags[gs * 4 + 0]) v = a.u[gs * 4 + c];                                  // PubCmd :369-378
            else {                                                                         // PubSlowDownCmd :379-397
                const double *o = a.odom[g] + (size_t)s * 10;
                v = -o[4 + c] * a.kp - o[7 + c] * a.kd + (c == 2 ? 9.8 : 0.0);
                const double lim = c == 2 ? a.a_max_z : a.a_max_xy;
                v = fmax(-lim, fmin(lim, v));
            }
            a.cmd[g][i] = v;
        }
    }
}

// mCurFrame.Twc = mat4Twb * mParamTbc (FrameKDMap.cpp:50) for the scenes whose frame produced an obstacle cloud (:39-41 returns
// before it otherwise); Twc == nullptr-initialised slots start from the identity (pipeline_twc_init_kernel)
__global__ __launch_bounds__(256) void pipeline_twc_update_kernel(const double *__restrict__ Twb, const int *__restrict__ counts,
                                                                  double *__restrict__ Twc, int S, const amk_depth_params prm) {
#pragma clang fp contract(off)   // the product the host twin forms (include/avoid_mpc_amd/frame_kd_map.hpp: AddVertex), sum in k order
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S * 16) return;
    const int s = e / 16, i = (e % 16) / 4, j = e % 4;
    if (counts[s] == 0) return;
    double acc = 0.0;
    for (int k = 0; k < 4; ++k) acc += Twb[s * 16 + 4 * i + k] * prm.Tbc[4 * k + j];
    Twc[e] = acc;
}
__global__ __launch_bounds__(256) void pipeline_twc_init_kernel(double *__restrict__ Twc, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n * 16) Twc[e] = (e % 16) % 5 == 0 ? 1.0 : 0.0;
}

// TASK-mode prologue (AvoidanceStateMachine.cpp:24-54,183-203,322-330): one wavefront per scene runs GetInitPath on the slot's
// own mRefPath (optionally re-initialised from the caller's first) and GetCurStateQuad for every re-plan pass.
struct PrologueArgs {
    const double *odom[AMK_PIPELINE_MAX_GANG];       // NULL: not a TASK-mode frame
    const double *ref_init[AMK_PIPELINE_MAX_GANG];   // NULL: the slot's own path
    double age[AMK_PIPELINE_MAX_GANG];
    const double *goal[AMK_PIPELINE_MAX_GANG];       // [S][3] or NULL: mStateGlobalGoal (task global_goal)
    double *sq_dst, *px_dst, *ref_dst;               // [G][S][mi][10], [G][S], [G][S][N][10]
    int S, N, mi;
    double decay, iter_time, farest, height, speed, T, dt;
    int use_odom_est, task;
};
__global__ __launch_bounds__(256) void pipeline_task_prologue_kernel(const PrologueArgs a) {
#pragma clang fp contract(off)   // bit-identical to the host twins (avoid_mpc_amd/fsm.py, include/avoid_mpc_amd/avoidance_step.hpp)
    const int g = blockIdx.y;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool live = a.odom[g] && s < a.S;
    const size_t gs = (size_t)g * a.S + (live ? s : 0);
    const double *o = live ? a.odom[g] + (size_t)s * 10 : nullptr;
    double *ref = a.ref_dst + gs * a.N * 10;
    const double *src = (live && a.ref_init[g]) ? a.ref_init[g] + (size_t)s * a.N * 10 : ref;
    // GetInitPath: mRefPath[i] = mRefPath[i + 1] with z = height, i < N - 1; the last point is the goal.  Every element is read
    // before any is written (the shift runs in place).
    const int n_shift = (a.N - 1) * 10;
    double v[5];   // N <= AMK_MAX_HORIZON = 32: 310 elements <= 5 x 64
    for (int k = 0; k < 5; ++k) {
        const int e = k * 64 + lane;
        v[k] = (live && e < n_shift) ? src[e + 10] : 0.0;
    }
    // the goal (:26-45), from the path as it is BEFORE the shift.  forward: speed * T ahead of the odometry position, capped;
    // global_goal: from the path's last point towards mStateGlobalGoal by at most speed * dt (dPos.normalized() * min(|dPos|, .):
    // Eigen's normalized() divides by the norm when it is positive and returns the vector unchanged otherwise)
    double goalx = 0.0, goaly = 0.0, goalz = a.height;
    if (live) {
        if (a.task == AMK_TASK_GLOBAL_GOAL) {
            const double *gg = a.goal[g] ? a.goal[g] + (size_t)s * 3 : nullptr;
            const double g0 = gg ? gg[0] : 0.0, g1 = gg ? gg[1] : 0.0, g2 = gg ? gg[2] : a.height;
            const double *last = src + n_shift;
            const double d0 = g0 - last[0], d1 = g1 - last[1], d2 = g2 - last[2];
            const double z = (d0 * d0 + d1 * d1) + d2 * d2, nrm = sqrt(z);
            const double step = fmin(nrm, a.speed * a.dt);
            const double e0 = z > 0.0 ? d0 / nrm : d0, e1 = z > 0.0 ? d1 / nrm : d1, e2 = z > 0.0 ? d2 / nrm : d2;
            goalx = last[0] + e0 * step; goaly = last[1] + e1 * step; goalz = last[2] + e2 * step;
        } else {
            goalx = fmin(a.speed * a.T + o[0], a.farest);
        }
    }
    __syncthreads();
    if (live) {
        for (int k = 0; k < 5; ++k) {
            const int e = k * 64 + lane;
            if (e < n_shift) ref[e] = (e % 10 == 2) ? goalz : v[k];
        }
        if (lane < 10) ref[n_shift + lane] = lane == 0 ? goalx : (lane == 1 ? goaly : (lane == 2 ? goalz : (lane == 4 ? a.speed : 0.0)));
        // GetCurStateQuad(start_i + decay_i) (:329-330,343): pass 0 extrapolates by odom_age + decay; pass i >= 1 starts i passes
        // later and extrapolates by the MEASURED duration of pass i - 1 -- with the clock model "every pass takes iter_time":
        // odom_age + (i + 1) * iter_time (written iter_time + i * iter_time so that the default iter_time = decay
        // keeps the bits of rounds 3-4)
        for (int e = lane; e < a.mi * 10; e += 64) {
            const int i = e / 10, j = e - 10 * i;
            const double dt = a.age[g] + (i == 0 ? a.decay : a.iter_time) + i * a.iter_time;
            double r;
            if (j < 3) r = a.use_odom_est ? o[j] + o[4 + j] * dt + 0.5 * o[7 + j] * dt * dt : o[j];
            else if (j == 3) r = o[3];
            else if (j < 7) r = a.use_odom_est ? o[j] + o[j + 3] * dt : o[j];
            else r = o[j];
            a.sq_dst[(gs * a.mi + i) * 10 + j] = r;
        }
        if (lane == 0) a.px_dst[gs] = o[0];   // mPos.x() of GetRefStates (:251): the odometry position, not the extrapolated one
    }
}

// Waits for a slot's launch.  hipEventSynchronize parks the thread on an HSA signal; with AMK_PIPELINE_SPIN=1 the thread polls
// hipEventQuery instead (diagnostics: tools/experiments/rccl_presence.py).
int wait_event(hipEvent_t ev) {
    static const bool spin = [] { const char *e = std::getenv("AMK_PIPELINE_SPIN"); return e && e[0] == '1'; }();
    if (!spin) {
        AMK_HIP(hipEventSynchronize(ev));
        return AMK_OK;
    }
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return AMK_OK;
        if (e != hipErrorNotReady) return amk::hip_fail(e);
    }
}

// Launches the slot's open gang: both index builds of every staged frame in one launch, one control step over all of them.
// A gang that is not full (wait / drain before the G-th submit) runs on the leading scenes of the slot's handles only.
int launch_gang(amk_pipeline *p, amk_pipeline::Slot &s) {
    if (s.open.empty()) return AMK_OK;
    const amk_pipeline_config &c = p->cfg;
    const int G = p->gang, S = c.n_scenes, N = amk_mpc_horizon(s.mpc), mi = c.step.mpc_max_iter;
    const int filled = (int)s.open.size();
    if (filled > G || filled > AMK_PIPELINE_MAX_GANG) return AMK_ERR_INVALID_ARG;   // (submit() never stages more than a gang)
    hipStream_t st = s.stream;
    for (const auto &f : s.open)   // inputs produced on the caller's streams: ordered on the device, not by the host
        if (f.input_ready) AMK_HIP(hipStreamWaitEvent(st, f.input_ready, 0));
    // a robot that starts over (a TASK frame with d_ref_path_init: InitCircleState's role) also starts with a new FrameKDMap:
    // no keyframes, mCurFrame.Twc = identity -- before the depth stage, whose edge cloud goes through that (stale) pose (:209)
    for (int g = 0; g < (int)s.open.size(); ++g) {
        const amk_pipeline::Staged &f = s.open[g];
        if (!(f.odom && f.ref_path_init)) continue;
        if (s.map) {
            const int rr = amk_kfmap_reset(s.map, g * S, S, st);
            if (rr != AMK_OK) return rr;
        }
        if (s.map && s.Twc.p) {   // (without a map the slot's Twc keeps its round-4 behaviour: it persists)
            hipLaunchKernelGGL(pipeline_twc_init_kernel, dim3((S * 16 + 255) / 256), dim3(256), 0, st, s.Twc.p + (size_t)g * S * 16, S);
            AMK_HIP(hipGetLastError());
        }
    }
    const float *cl[AMK_PIPELINE_MAX_GANG], *ed[AMK_PIPELINE_MAX_GANG];
    const int *cc[AMK_PIPELINE_MAX_GANG], *ec[AMK_PIPELINE_MAX_GANG];
    GatherArgs ga{};
    PrologueArgs pa{};
    bool any_task = false;
    for (int g = 0; g < filled; ++g) {
        const amk_pipeline::Staged &f = s.open[g];
        cl[g] = f.cloud; ed[g] = f.edge; cc[g] = f.cloud_counts; ec[g] = f.edge_counts;
        const bool task = f.odom != nullptr;
        any_task = any_task || task;
        ga.sq[g] = task ? nullptr : f.state_quad; ga.px[g] = task ? nullptr : f.pos_x; ga.ref[g] = task ? nullptr : f.ref_path_init;
        pa.odom[g] = f.odom; pa.ref_init[g] = f.ref_path_init; pa.age[g] = f.odom_age; pa.goal[g] = f.global_goal;
        // fresh frame: mRefPath after GetInitPath, zero warm start unless the caller carries it over (HighLvlMpc.cpp:26-27,35,129)
        ga.keep_warm_start[g] = f.keep_warm_start;
    }
    const bool own_inputs = G > 1 || any_task;   // the step reads the slot's contiguous copies (gang 1 without TASK mode: the caller's)
    ga.sq_dst = own_inputs ? s.state_quad.p : nullptr; ga.px_dst = own_inputs ? s.pos_x.p : nullptr;
    ga.ref_dst = s.ref_path.p; ga.w0 = s.mpc->w0.p;
    ga.n_sq = S * mi * 10; ga.n_px = S; ga.n_ref = S * N * 10; ga.n_w0 = S * s.mpc->nx;
    {
        const int longest = ga.n_w0 > ga.n_ref ? ga.n_w0 : ga.n_ref;
        int bx = (longest + 1023) / 1024;
        bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
        hipLaunchKernelGGL(pipeline_gather_kernel, dim3(bx, filled), dim3(256), 0, st, ga);
        AMK_HIP(hipGetLastError());
    }
    if (any_task) {
        const amk_task_params &t = c.task;
        pa.sq_dst = s.state_quad.p; pa.px_dst = s.pos_x.p; pa.ref_dst = s.ref_path.p;
        pa.S = S; pa.N = N; pa.mi = mi;
        pa.decay = t.decay; pa.iter_time = t.iter_time > 0 ? t.iter_time : t.decay; pa.farest = t.farest_point; pa.height = t.height;
        pa.speed = c.step.speed; pa.T = c.T; pa.dt = c.dt; pa.use_odom_est = t.use_odom_est; pa.task = t.task;
        hipLaunchKernelGGL(pipeline_task_prologue_kernel, dim3((S + 3) / 4, filled), dim3(256), 0, st, pa);
        AMK_HIP(hipGetLastError());
    }
    int rc;
    if (p->inject_failure == 1) { p->inject_failure = 0; return AMK_ERR_HIP; }
    // frames that start at the depth image: ProcessDepth + BuildEdgeCloud (FrameKDMap.cpp:90-130,176-214) into the slot's clouds
    const int *keep[AMK_PIPELINE_MAX_GANG];
    bool any_depth = false;
    for (int g = 0; g < filled; ++g) {
        const amk_pipeline::Staged &f = s.open[g];
        keep[g] = nullptr;
        if (!f.depth) continue;
        if (s.point_stride != 3) return AMK_ERR_INVALID_ARG;   // the slot's own clouds are packed xyz
        if (!s.dcloud.p || !s.dedge.p || !s.dcount.p || !s.decount.p || !s.Twc.p) {   // first depth frame of this slot (or an
            // earlier attempt ran out of memory half-way: ADVICE r4 -- all five buffers or none)
            const size_t GS = (size_t)G * S;
            hipError_t e;
            if ((e = s.dcloud.alloc(GS * c.max_points * 3)) != hipSuccess || (e = s.dedge.alloc(GS * c.max_edge_points * 3)) != hipSuccess ||
                (e = s.dcount.alloc(GS)) != hipSuccess || (e = s.decount.alloc(GS)) != hipSuccess || (e = s.Twc.alloc(GS * 16)) != hipSuccess) {
                s.dcloud.release(); s.dedge.release(); s.dcount.release(); s.decount.release(); s.Twc.release();
                return amk::hip_fail(e);
            }
            hipLaunchKernelGGL(pipeline_twc_init_kernel, dim3((unsigned)((GS * 16 + 255) / 256)), dim3(256), 0, st, s.Twc.p, (int)GS);
            AMK_HIP(hipGetLastError());
        }
        any_depth = true;
        const size_t o = (size_t)g * S;
        float *dc = s.dcloud.p + o * c.max_points * 3, *de = s.dedge.p + o * c.max_edge_points * 3;
        const long long img = (long long)f.depth_rows * f.depth_cols;
        rc = amk_depth_to_cloud(f.depth, f.depth_type, f.depth_rows, f.depth_cols, img, S, &c.depth, f.Twb, dc, 3,
                                (long long)c.max_points * 3, s.dcount.p + o, st);
        if (rc != AMK_OK) return rc;
        rc = amk_depth_to_edge_cloud(f.depth, f.depth_type, f.depth_rows, f.depth_cols, img, S, &c.depth, s.Twc.p + o * 16, de, 3,
                                     (long long)c.max_edge_points * 3, s.decount.p + o, st);
        if (rc != AMK_OK) return rc;
        hipLaunchKernelGGL(pipeline_twc_update_kernel, dim3((S * 16 + 255) / 256), dim3(256), 0, st, f.Twb, s.dcount.p + o, s.Twc.p + o * 16, S, c.depth);
        AMK_HIP(hipGetLastError());
        cl[g] = dc; ed[g] = de; cc[g] = s.dcount.p + o; ec[g] = s.decount.p + o;
        keep[g] = s.dcount.p + o;
    }
    // FrameKDMap::AddVertex: obstacle index and edge index of every frame (FrameKDMap.cpp:44-47)
    static const bool trace_host = [] { const char *e = std::getenv("AMK_PIPELINE_TRACE"); return e && e[0] == '1'; }();
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_h0 = trace_host ? now_us() : 0.0;
    double t_h1 = 0.0, t_h2 = 0.0;
    if (s.map) {   // ... into the slot's keyframe map: the scenes' own physical slots, mCurFrame.Twc, then KeyframeThreadWorker's body
        const double *twcs[AMK_PIPELINE_MAX_GANG] = {};
        for (int g = 0; g < filled; ++g) {
            const amk_pipeline::Staged &f = s.open[g];
            twcs[g] = f.depth ? s.Twc.p + (size_t)g * S * 16 : f.Twc_cur;
            if (!twcs[g]) return AMK_ERR_INVALID_ARG;   // (submit() checked)
        }
        // AddVertex of every position of the gang: one bookkeeping launch + one build launch (both trees of all frames)
        rc = amk::kfmap_add_vertex_gang(s.map, filled, S, cl, cc, ed, ec, s.point_stride, twcs, st);
        if (rc != AMK_OK) return rc;
        if (trace_host) t_h1 = now_us();
        rc = amk_kfmap_update(s.map, st);
        if (trace_host) t_h2 = now_us();
    } else if (G == 1 && !any_depth) rc = amk_kd_build_pair(s.obstacle, cl[0], cc[0], s.edge, ed[0], ec[0], s.point_stride, st);
    else rc = amk::kd_build_gang(s.obstacle, s.edge, filled, S, cl, cc, ed, ec, s.point_stride, st, any_depth ? keep : nullptr);
    if (rc != AMK_OK) return rc;
    if (p->inject_failure == 2) { p->inject_failure = 0; return AMK_ERR_HIP; }
    double *u = (G == 1 && s.open[0].u_out && !any_task) ? s.open[0].u_out : s.u.p;
    const double *sq = own_inputs ? s.state_quad.p : s.open[0].state_quad, *px = own_inputs ? s.pos_x.p : s.open[0].pos_x;
    s.mpc->run_scenes = filled * S;
    if (s.map) {   // the step over the slot's own map: mVecQueryVector = [current, keyframes but the newest] of every scene
        const amk_pipeline::Staged &f = s.open[0];
        amk_frame_camera cam{};
        bool has_cam = f.has_camera;
        if (has_cam) cam = f.camera;
        if (f.depth) {   // the FrameKDMap constructor's down-scaled intrinsics and ProcessDepth's image size (:21-24,106-107)
            int w = 0, h = 0;
            rc = amk_depth_out_size(f.depth_rows, f.depth_cols, c.depth.resize_scale, &w, &h);
            if (rc != AMK_OK) return rc;
            cam.fx = c.depth.fx / c.depth.resize_scale; cam.fy = c.depth.fy / c.depth.resize_scale;
            cam.cx = c.depth.cx / c.depth.resize_scale; cam.cy = c.depth.cy / c.depth.resize_scale;
            cam.depth_max = c.depth.depth_max; cam.width = w; cam.height = h;
            has_cam = true;
        }
        rc = amk_kfmap_step(s.map, has_cam ? &cam : nullptr, s.mpc, &c.step, sq, px, s.ref_path.p, u, s.x0array.p, s.flags.p, st);
    } else if (!s.open[0].kf_obstacle.empty()) {   // mVecQueryVector = [this frame, keyframes ...] (gang 1: submit() checked)
        const amk_pipeline::Staged &f = s.open[0];
        std::vector<amk_kd *> ob{s.obstacle}, ed2{s.edge};
        ob.insert(ob.end(), f.kf_obstacle.begin(), f.kf_obstacle.end());
        ed2.insert(ed2.end(), f.kf_edge.begin(), f.kf_edge.end());
        const double *twc = f.Twc_cur ? f.Twc_cur : (f.depth ? s.Twc.p : nullptr);
        rc = amk_step_batch_frames(ob.data(), ed2.data(), (int)ob.size(), twc, f.has_camera ? &f.camera : nullptr, s.mpc, &c.step, sq, px,
                                   s.ref_path.p, u, s.x0array.p, s.flags.p, st);
    } else {
        rc = amk_step_batch(s.obstacle, s.edge, s.mpc, &c.step, sq, px, s.ref_path.p, u, s.x0array.p, s.flags.p, st);
    }
    s.mpc->run_scenes = 0;
    if (trace_host)   // diagnostics (AMK_PIPELINE_TRACE=1): host microseconds spent enqueueing the stages of this launch
        fprintf(stderr, "amk_pipeline launch: add_vertex %.0f us, map update %.0f us, step %.0f us (host enqueue time)\n", t_h1 - t_h0,
                t_h2 - t_h1, now_us() - (s.map ? t_h2 : t_h0));
    if (rc != AMK_OK) return rc;
    if (G > 1 || any_task) {
        ScatterArgs sa{};
        bool any = false;
        for (int g = 0; g < filled; ++g) {
            sa.dst[g] = s.open[g].u_out; sa.cmd[g] = s.open[g].odom ? s.open[g].cmd_out : nullptr; sa.odom[g] = s.open[g].odom;
            any = any || sa.dst[g] || sa.cmd[g];
        }
        sa.u = s.u.p; sa.flags = s.flags.p; sa.n_u = S * 4; sa.S = S;
        sa.kp = c.task.slow_down_kp; sa.kd = c.task.slow_down_kd; sa.a_max_xy = c.task.a_max_xy; sa.a_max_z = c.task.a_max_z;
        if (any) {
            hipLaunchKernelGGL(pipeline_scatter_kernel, dim3((sa.n_u + 255) / 256 > 16 ? 16 : (sa.n_u + 255) / 256, filled), dim3(256), 0, st, sa);
            AMK_HIP(hipGetLastError());
        }
    }
    AMK_HIP(hipEventRecord(s.done[s.count % p->depth], st));
    ++s.count;
    for (int g = 0; g < filled; ++g)
        if (s.open[g].odom && s.open[g].ref_path_init) s.ref_inited[g] = true;
    s.open.clear();
    return AMK_OK;
}

// launch_gang with the slot left consistent whatever happens: on a failure half-way (a build or a step that could not be
// launched) the staged frames are DROPPED -- a later submit() starts a new gang instead of re-launching a half-consumed one,
// and wait() / drain() have an event to wait for (what did get enqueued stays ordered on the slot's stream) and report the
// failure.  The round robin moves on either way.
int launch_slot(amk_pipeline *p, amk_pipeline::Slot &s) {
    if (s.open.empty()) return AMK_OK;
    static const bool trace_host = [] { const char *e = std::getenv("AMK_PIPELINE_TRACE"); return e && e[0] == '1'; }();
    const auto t0 = std::chrono::steady_clock::now();
    const int st = launch_gang(p, s);
    if (trace_host)
        fprintf(stderr, "amk_pipeline launch_gang total: %.0f us (host)\n",
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    if (st != AMK_OK) {
        // What had been enqueued before the failure stays enqueued: a TASK frame's GetInitPath may already have shifted the slot's
        // mRefPath, its warm start may have been reset, a depth frame's Twc may have moved on.  The slot's persistent
        // TASK state is therefore declared lost: the next TASK frame at any position of this slot must bring d_ref_path_init
        // (submit() rejects it otherwise) -- re-submitting the failed period as it was would shift the path twice.
        for (bool &b : s.ref_inited) b = false;
        s.open.clear();
        s.mpc->run_scenes = 0;
        (void)hipEventRecord(s.done[s.count % p->depth], s.stream);
        ++s.count;
    }
    s.fail_status = st;
    if (&s == &p->slots[p->next]) p->next = (p->next + 1) % (int)p->slots.size();
    return st;
}
}  // namespace

extern "C" {

int amk_pipeline_create(const amk_pipeline_config *cfg, amk_pipeline **out) {
    if (!cfg || !out || cfg->n_slots <= 0 || cfg->n_slots > AMK_PIPELINE_MAX_SLOTS || cfg->n_scenes <= 0 ||
        cfg->max_points <= 0 || cfg->max_edge_points <= 0 || cfg->gang < 0 || cfg->gang > AMK_PIPELINE_MAX_GANG ||
        cfg->step.mpc_max_iter < 1 || cfg->step.mpc_max_iter > AMK_MAX_OUTER_ITER || cfg->keyframes.max_frame_count < 0 ||
        (cfg->task.task != AMK_TASK_FORWARD && cfg->task.task != AMK_TASK_GLOBAL_GOAL))
        return AMK_ERR_INVALID_ARG;
    *out = nullptr;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    amk_pipeline *p = new amk_pipeline();
    p->cfg = *cfg;
    p->depth = cfg->queue_depth > 0 ? cfg->queue_depth : AMK_PIPELINE_DEFAULT_DEPTH;
    if (p->depth > AMK_PIPELINE_MAX_DEPTH) p->depth = AMK_PIPELINE_MAX_DEPTH;
    p->gang = cfg->gang > 0 ? cfg->gang : 1;
    p->slots.resize(cfg->n_slots);
    const int GS = p->gang * cfg->n_scenes;
    int st = AMK_OK;
    for (auto &s : p->slots) {
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess) {
            st = amk::hip_fail(e);
            break;
        }
        s.done.assign(p->depth, nullptr);
        for (auto &ev : s.done)
            if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) break;
        if (e != hipSuccess) {
            st = amk::hip_fail(e);
            break;
        }
        if ((st = amk_kd_create(GS, cfg->max_points, &s.obstacle)) != AMK_OK) break;
        if ((st = amk_kd_create(GS, cfg->max_edge_points, &s.edge)) != AMK_OK) break;
        if ((st = amk_mpc_create(cfg->T, cfg->dt, cfg->nearest_point_num, GS, &s.mpc)) != AMK_OK) break;
        if (cfg->keyframes.max_frame_count > 0) {
            amk_kfmap_params kp = cfg->keyframes;
            auto all_zero = [](const double *T) { bool z = true; for (int e = 0; e < 16; ++e) z = z && T[e] == 0.0; return z; };
            // mParamTbc: the map's own field when it is set, else the depth configuration's (one extrinsic serves both in the
            // reference, FrameKDMap.cpp:15), else camera frame = body frame
            if (all_zero(kp.Tbc)) std::memcpy(kp.Tbc, cfg->depth.Tbc, sizeof kp.Tbc);
            if (all_zero(kp.Tbc)) kp.Tbc[0] = kp.Tbc[5] = kp.Tbc[10] = kp.Tbc[15] = 1.0;
            if ((st = amk_kfmap_create(GS, cfg->max_points, cfg->max_edge_points, &kp, &s.map)) != AMK_OK) break;
        }
        const size_t S = GS, N = amk_mpc_horizon(s.mpc);
        if ((e = s.ref_path.alloc(S * N * 10)) != hipSuccess || (e = s.u.alloc(S * 4)) != hipSuccess ||
            (e = s.x0array.alloc(S * N * 14)) != hipSuccess || (e = s.flags.alloc(S * 4)) != hipSuccess) {
            st = amk::hip_fail(e);
            break;
        }
        if ((e = s.state_quad.alloc(S * cfg->step.mpc_max_iter * 10)) != hipSuccess || (e = s.pos_x.alloc(S)) != hipSuccess) {
            st = amk::hip_fail(e);
            break;
        }
    }
    if (st != AMK_OK) {
        amk_pipeline_destroy(p);
        return st;
    }
    *out = p;
    return AMK_OK;
}

int amk_pipeline_destroy(amk_pipeline *p) {
    if (!p) return AMK_ERR_INVALID_ARG;
    for (auto &s : p->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.obstacle) amk_kd_destroy(s.obstacle);
        if (s.edge) amk_kd_destroy(s.edge);
        if (s.mpc) amk_mpc_destroy(s.mpc);
        if (s.map) amk_kfmap_destroy(s.map);
        for (auto ev : s.done)
            if (ev) (void)hipEventDestroy(ev);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete p;
    return AMK_OK;
}

int amk_pipeline_slots(const amk_pipeline *p) { return p ? (int)p->slots.size() : -1; }
int amk_pipeline_gang(const amk_pipeline *p) { return p ? p->gang : -1; }

amk_mpc *amk_pipeline_mpc(amk_pipeline *p, int slot) {
    return (p && slot >= 0 && slot < (int)p->slots.size()) ? p->slots[slot].mpc : nullptr;
}
amk_kfmap *amk_pipeline_kfmap(amk_pipeline *p, int slot) {
    return (p && slot >= 0 && slot < (int)p->slots.size()) ? p->slots[slot].map : nullptr;
}
amk_kd *amk_pipeline_kd(amk_pipeline *p, int slot, int which) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return nullptr;
    return which == 0 ? p->slots[slot].obstacle : (which == 1 ? p->slots[slot].edge : nullptr);
}
void *amk_pipeline_stream(amk_pipeline *p, int slot) {
    return (p && slot >= 0 && slot < (int)p->slots.size()) ? (void *)p->slots[slot].stream : nullptr;
}

int amk_pipeline_submit(amk_pipeline *p, const amk_pipeline_frame *f, int *ticket_out) {
    if (!p || !f) return AMK_ERR_INVALID_ARG;
    if (f->d_depth) {   // a frame that starts at the depth image
        if (!f->d_Twb || f->depth_rows <= 0 || f->depth_cols <= 0 || (f->depth_type != AMK_DEPTH_U16 && f->depth_type != AMK_DEPTH_F32))
            return AMK_ERR_INVALID_ARG;
        int w = 0, h = 0;
        if (amk_depth_out_size(f->depth_rows, f->depth_cols, p->cfg.depth.resize_scale, &w, &h) != AMK_OK) return AMK_ERR_INVALID_ARG;
        if ((long long)w * h > p->cfg.max_points || (long long)w * h > p->cfg.max_edge_points) return AMK_ERR_INVALID_ARG;
        if ((long long)w * h > AMK_EDGE_MAX_PIXELS) return AMK_ERR_UNSUPPORTED;
    } else if (!f->d_cloud || !f->d_edge) return AMK_ERR_INVALID_ARG;
    if (!f->d_odom && (!f->d_state_quad || !f->d_pos_x || !f->d_ref_path_init)) return AMK_ERR_INVALID_ARG;
    const int si = p->next, ns = (int)p->slots.size();
    auto &s = p->slots[si];
    if (f->n_keyframes < 0 || f->n_keyframes > AMK_MAX_FRAMES - 1) return AMK_ERR_INVALID_ARG;
    // the slot's own map: no caller keyframes; cloud frames bring mCurFrame.Twc AND the camera model -- the reference always runs
    // PtIsInFrame (FrameKDMap.cpp:339-345); without a camera every reference point would count as inside the current frame and
    // the keyframes would never be merged
    if (s.map && (f->n_keyframes > 0 || (!f->d_depth && (!f->d_Twc_cur || !f->camera)))) return AMK_ERR_INVALID_ARG;
    if (s.map && !s.open.empty()) {   // the map step takes ONE camera model per launch (position 0's): a gang must agree on it
        const amk_pipeline::Staged &o = s.open[0];
        if ((o.depth != nullptr) != (f->d_depth != nullptr)) return AMK_ERR_INVALID_ARG;
        if (f->d_depth && (o.depth_rows != f->depth_rows || o.depth_cols != f->depth_cols)) return AMK_ERR_INVALID_ARG;
        if (!f->d_depth && std::memcmp(&o.camera, f->camera, sizeof(amk_frame_camera)) != 0) return AMK_ERR_INVALID_ARG;
    }
    if (f->n_keyframes > 0) {
        if (p->gang != 1) return AMK_ERR_UNSUPPORTED;   // keyframe handles hold n_scenes scenes, a gang's handles gang x n_scenes
        if (!f->kf_obstacle || !f->kf_edge) return AMK_ERR_INVALID_ARG;
        for (int i = 0; i < f->n_keyframes; ++i)
            if (!f->kf_obstacle[i] || !f->kf_edge[i]) return AMK_ERR_INVALID_ARG;
    }
    // a TASK frame shifts the mRefPath the slot keeps for its position: there must be one (InitCircleState's role, :14-23)
    if (f->d_odom && !f->d_ref_path_init && !s.ref_inited[s.open.size() < AMK_PIPELINE_MAX_GANG ? s.open.size() : 0]) return AMK_ERR_INVALID_ARG;
    const int stride = f->d_depth ? 3 : (f->point_stride ? f->point_stride : 3);
    if (stride != 3 && stride != 4) return AMK_ERR_INVALID_ARG;
    if (!s.open.empty() && stride != s.point_stride) return AMK_ERR_INVALID_ARG;   // one point layout per gang
    if ((int)s.open.size() >= p->gang) return AMK_ERR_INVALID_ARG;                  // (cannot happen: a full gang is launched at once)
    // Flow control.  A slot's launches are ordered by its stream, so queuing the next one behind a running one is safe (same
    // handles, same workspaces, in order); submit() only blocks when `depth` launches of this slot are still unfinished --
    // the event about to be re-recorded belongs to the launch issued `depth` launches ago.  depth 1 = at most one launch per
    // slot on the device (every launch then pays the host's reaction time between its predecessor's end and its own start).
    if (s.open.empty() && s.count - s.waited >= p->depth) {
        static const bool trace_host = [] { const char *e = std::getenv("AMK_PIPELINE_TRACE"); return e && e[0] == '1'; }();
        const auto t0 = std::chrono::steady_clock::now();
        const int ws = wait_event(s.done[s.count % p->depth]);
        if (trace_host)
            fprintf(stderr, "amk_pipeline submit: waited %.0f us for the launch issued %d launches ago on slot %d\n",
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), p->depth, si);
        if (ws != AMK_OK) return ws;
        s.waited = s.count - p->depth + 1;
    }
    const int g = (int)s.open.size();
    s.point_stride = stride;
    // staged: the frame's inputs are read when its gang is launched (they stay the caller's until then)
    s.open.push_back(amk_pipeline::Staged{f->d_cloud, f->d_edge, f->d_cloud_counts, f->d_edge_counts, f->d_state_quad, f->d_pos_x,
                                           f->d_ref_path_init, f->d_u_out, f->keep_warm_start, (hipEvent_t)f->input_ready,
                                           f->d_odom, f->odom_age, f->d_cmd_out, f->d_depth, f->depth_type, f->depth_rows, f->depth_cols, f->d_Twb,
                                           {}, {}, f->d_Twc_cur, f->camera != nullptr, f->camera ? *f->camera : amk_frame_camera{}, f->d_global_goal});
    if (f->n_keyframes > 0) {
        s.open.back().kf_obstacle.assign(f->kf_obstacle, f->kf_obstacle + f->n_keyframes);
        s.open.back().kf_edge.assign(f->kf_edge, f->kf_edge + f->n_keyframes);
    }
    if (ticket_out) *ticket_out = g * ns + si;
    ++p->submitted;
    if ((int)s.open.size() == p->gang) return launch_slot(p, s);
    return AMK_OK;
}

int amk_pipeline_wait(amk_pipeline *p, int ticket) {
    if (!p || ticket < 0 || ticket >= (int)p->slots.size() * p->gang) return AMK_ERR_INVALID_ARG;
    auto &s = p->slots[ticket % (int)p->slots.size()];
    if (!s.open.empty()) {   // a partly filled gang: launch it now
        const int st = launch_slot(p, s);
        if (st != AMK_OK) return st;
    }
    if (s.waited < s.count) {   // the newest event implies all earlier ones (in-order stream)
        const int ws = wait_event(s.done[(s.count - 1) % p->depth]);
        if (ws != AMK_OK) return ws;
        s.waited = s.count;
    }
    return s.fail_status;   // AMK_OK unless the slot's newest launch failed half-way (its frames were dropped)
}

int amk_pipeline_wait_stream(amk_pipeline *p, int ticket, void *stream) {
    if (!p || ticket < 0 || ticket >= (int)p->slots.size() * p->gang) return AMK_ERR_INVALID_ARG;
    auto &s = p->slots[ticket % (int)p->slots.size()];
    if (!s.open.empty()) {
        const int st = launch_slot(p, s);
        if (st != AMK_OK) return st;
    }
    if (s.count > 0) AMK_HIP(hipStreamWaitEvent((hipStream_t)stream, s.done[(s.count - 1) % p->depth], 0));
    return s.fail_status;
}

int amk_pipeline_query(amk_pipeline *p, int ticket) {  // 1 = finished (or idle), 0 = still running or staged, -1 error
    if (!p || ticket < 0 || ticket >= (int)p->slots.size() * p->gang) return -1;
    auto &s = p->slots[ticket % (int)p->slots.size()];
    if (!s.open.empty()) return 0;
    if (s.fail_status != AMK_OK) return -1;   // the slot's newest launch failed half-way: its frames were dropped (wait() tells why)
    if (s.waited >= s.count) return 1;
    const hipError_t e = hipEventQuery(s.done[(s.count - 1) % p->depth]);
    if (e == hipSuccess) { s.waited = s.count; return 1; }
    return e == hipErrorNotReady ? 0 : -1;
}

int amk_pipeline_drain(amk_pipeline *p) {
    if (!p) return AMK_ERR_INVALID_ARG;
    int first_error = AMK_OK;   // every slot is launched and waited for even when one fails: drain() leaves nothing in flight
    for (auto &s : p->slots) {   // every open gang first: a launch must not wait for the slots before it to finish
        if (s.open.empty()) continue;
        const int st = launch_slot(p, s);
        if (st != AMK_OK && first_error == AMK_OK) first_error = st;
    }
    for (int i = 0; i < (int)p->slots.size(); ++i) {
        const int st = amk_pipeline_wait(p, i);
        if (st != AMK_OK && first_error == AMK_OK) first_error = st;
    }
    return first_error;
}

// tests only: the next launch of this pipeline fails at stage 1 (after the gather of the small inputs, before the index
// builds) or 2 (after the builds, before the control step) with AMK_ERR_HIP
int amk__pipeline_inject_failure(amk_pipeline *p, int stage) {
    if (!p || stage < 0 || stage > 2) return AMK_ERR_INVALID_ARG;
    p->inject_failure = stage;
    return AMK_OK;
}

int amk_pipeline_outputs(amk_pipeline *p, int ticket, double **d_u, double **d_x0array, int **d_flags, double **d_ref_path) {
    if (!p || ticket < 0 || ticket >= (int)p->slots.size() * p->gang) return AMK_ERR_INVALID_ARG;
    const int ns = (int)p->slots.size(), g = ticket / ns;
    auto &s = p->slots[ticket % ns];
    const size_t S = p->cfg.n_scenes, N = amk_mpc_horizon(s.mpc), o = (size_t)g * S;
    if (d_u) *d_u = s.u.p + o * 4;
    if (d_x0array) *d_x0array = s.x0array.p + o * N * 14;
    if (d_flags) *d_flags = s.flags.p + o * 4;
    if (d_ref_path) *d_ref_path = s.ref_path.p + o * N * 10;
    return AMK_OK;
}

}  // extern "C"
