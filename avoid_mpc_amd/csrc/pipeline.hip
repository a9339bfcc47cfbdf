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

#include <cstdlib>
#include <cstring>
#include <vector>

struct amk_pipeline {
    amk_pipeline_config cfg;
    struct Staged {  // a frame of the open gang
        const float *cloud, *edge;
        const int *cloud_counts, *edge_counts;
        const double *state_quad, *pos_x;   // read directly when gang == 1, gathered at launch otherwise
        const double *ref_path_init;
        double *u_out;
        int keep_warm_start;
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
    };
    int depth = 1, gang = 1;
    std::vector<Slot> slots;
    int next = 0;
    long long submitted = 0;
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
    if (a.sq_dst) {
        for (int i = t0; i < a.n_sq; i += stride) a.sq_dst[(size_t)g * a.n_sq + i] = a.sq[g][i];
        for (int i = t0; i < a.n_px; i += stride) a.px_dst[(size_t)g * a.n_px + i] = a.px[g][i];
    }
    for (int i = t0; i < a.n_ref; i += stride) a.ref_dst[(size_t)g * a.n_ref + i] = a.ref[g][i];
    if (!a.keep_warm_start[g])
        for (int i = t0; i < a.n_w0; i += stride) a.w0[(size_t)g * a.n_w0 + i] = 0.0;
}
// ... and the controls of every frame to where its caller wants them
struct ScatterArgs {
    double *dst[AMK_PIPELINE_MAX_GANG];   // NULL: stays in the slot's buffer only
    const double *u;
    int n_u;
};
__global__ __launch_bounds__(256) void pipeline_scatter_kernel(const ScatterArgs a) {
    const int g = blockIdx.y;
    if (!a.dst[g]) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n_u; i += gridDim.x * 256) a.dst[g][i] = a.u[(size_t)g * a.n_u + i];
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
    hipStream_t st = s.stream;
    const float *cl[AMK_PIPELINE_MAX_GANG], *ed[AMK_PIPELINE_MAX_GANG];
    const int *cc[AMK_PIPELINE_MAX_GANG], *ec[AMK_PIPELINE_MAX_GANG];
    GatherArgs ga{};
    for (int g = 0; g < filled; ++g) {
        const amk_pipeline::Staged &f = s.open[g];
        cl[g] = f.cloud; ed[g] = f.edge; cc[g] = f.cloud_counts; ec[g] = f.edge_counts;
        ga.sq[g] = f.state_quad; ga.px[g] = f.pos_x; ga.ref[g] = f.ref_path_init;
        // fresh frame: mRefPath after GetInitPath, zero warm start unless the caller carries it over (HighLvlMpc.cpp:26-27,35,129)
        ga.keep_warm_start[g] = f.keep_warm_start;
    }
    ga.sq_dst = G > 1 ? s.state_quad.p : nullptr; ga.px_dst = G > 1 ? s.pos_x.p : nullptr;
    ga.ref_dst = s.ref_path.p; ga.w0 = s.mpc->w0.p;
    ga.n_sq = S * mi * 10; ga.n_px = S; ga.n_ref = S * N * 10; ga.n_w0 = S * s.mpc->nx;
    {
        const int longest = ga.n_w0 > ga.n_ref ? ga.n_w0 : ga.n_ref;
        int bx = (longest + 1023) / 1024;
        bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
        hipLaunchKernelGGL(pipeline_gather_kernel, dim3(bx, filled), dim3(256), 0, st, ga);
        AMK_HIP(hipGetLastError());
    }
    int rc;
    // FrameKDMap::AddVertex: obstacle index and edge index of every frame (FrameKDMap.cpp:44-47)
    if (G == 1) rc = amk_kd_build_pair(s.obstacle, cl[0], cc[0], s.edge, ed[0], ec[0], s.point_stride, st);
    else rc = amk::kd_build_gang(s.obstacle, s.edge, filled, S, cl, cc, ed, ec, s.point_stride, st);
    if (rc != AMK_OK) return rc;
    double *u = (G == 1 && s.open[0].u_out) ? s.open[0].u_out : s.u.p;
    const double *sq = G == 1 ? s.open[0].state_quad : s.state_quad.p, *px = G == 1 ? s.open[0].pos_x : s.pos_x.p;
    s.mpc->run_scenes = filled * S;
    rc = amk_step_batch(s.obstacle, s.edge, s.mpc, &c.step, sq, px, s.ref_path.p, u, s.x0array.p, s.flags.p, st);
    s.mpc->run_scenes = 0;
    if (rc != AMK_OK) return rc;
    if (G > 1) {
        ScatterArgs sa{};
        bool any = false;
        for (int g = 0; g < filled; ++g) { sa.dst[g] = s.open[g].u_out; any = any || sa.dst[g]; }
        sa.u = s.u.p; sa.n_u = S * 4;
        if (any) {
            hipLaunchKernelGGL(pipeline_scatter_kernel, dim3((sa.n_u + 255) / 256 > 16 ? 16 : (sa.n_u + 255) / 256, filled), dim3(256), 0, st, sa);
            AMK_HIP(hipGetLastError());
        }
    }
    AMK_HIP(hipEventRecord(s.done[s.count % p->depth], st));
    ++s.count;
    s.open.clear();
    return AMK_OK;
}
}  // namespace

extern "C" {

int amk_pipeline_create(const amk_pipeline_config *cfg, amk_pipeline **out) {
    if (!cfg || !out || cfg->n_slots <= 0 || cfg->n_slots > AMK_PIPELINE_MAX_SLOTS || cfg->n_scenes <= 0 ||
        cfg->max_points <= 0 || cfg->max_edge_points <= 0 || cfg->gang < 0 || cfg->gang > AMK_PIPELINE_MAX_GANG ||
        cfg->step.mpc_max_iter < 1 || cfg->step.mpc_max_iter > AMK_MAX_OUTER_ITER)
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
        const size_t S = GS, N = amk_mpc_horizon(s.mpc);
        if ((e = s.ref_path.alloc(S * N * 10)) != hipSuccess || (e = s.u.alloc(S * 4)) != hipSuccess ||
            (e = s.x0array.alloc(S * N * 14)) != hipSuccess || (e = s.flags.alloc(S * 4)) != hipSuccess) {
            st = amk::hip_fail(e);
            break;
        }
        if (p->gang > 1 && ((e = s.state_quad.alloc(S * cfg->step.mpc_max_iter * 10)) != hipSuccess || (e = s.pos_x.alloc(S)) != hipSuccess)) {
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
amk_kd *amk_pipeline_kd(amk_pipeline *p, int slot, int which) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return nullptr;
    return which == 0 ? p->slots[slot].obstacle : (which == 1 ? p->slots[slot].edge : nullptr);
}
void *amk_pipeline_stream(amk_pipeline *p, int slot) {
    return (p && slot >= 0 && slot < (int)p->slots.size()) ? (void *)p->slots[slot].stream : nullptr;
}

int amk_pipeline_submit(amk_pipeline *p, const amk_pipeline_frame *f, int *ticket_out) {
    if (!p || !f || !f->d_cloud || !f->d_edge || !f->d_state_quad || !f->d_pos_x || !f->d_ref_path_init)
        return AMK_ERR_INVALID_ARG;
    const int si = p->next, ns = (int)p->slots.size();
    auto &s = p->slots[si];
    const int stride = f->point_stride ? f->point_stride : 3;
    if (!s.open.empty() && stride != s.point_stride) return AMK_ERR_INVALID_ARG;   // one point layout per gang
    // Flow control.  A slot's launches are ordered by its stream, so queuing the next one behind a running one is safe (same
    // handles, same workspaces, in order); submit() only blocks when `depth` launches of this slot are still unfinished --
    // the event about to be re-recorded belongs to the launch issued `depth` launches ago.  depth 1 = at most one launch per
    // slot on the device (every launch then pays the host's reaction time between its predecessor's end and its own start).
    if (s.open.empty() && s.count - s.waited >= p->depth) {
        const int ws = wait_event(s.done[s.count % p->depth]);
        if (ws != AMK_OK) return ws;
        s.waited = s.count - p->depth + 1;
    }
    const int g = (int)s.open.size();
    s.point_stride = stride;
    // staged: the frame's inputs are read when its gang is launched (they stay the caller's until then)
    s.open.push_back(amk_pipeline::Staged{f->d_cloud, f->d_edge, f->d_cloud_counts, f->d_edge_counts, f->d_state_quad, f->d_pos_x,
                                           f->d_ref_path_init, f->d_u_out, f->keep_warm_start});
    if (ticket_out) *ticket_out = g * ns + si;
    ++p->submitted;
    const bool launch = (int)s.open.size() == p->gang;
    if (launch) {
        const int st = launch_gang(p, s);
        if (st != AMK_OK) return st;
        p->next = (si + 1) % ns;
    }
    return AMK_OK;
}

int amk_pipeline_wait(amk_pipeline *p, int ticket) {
    if (!p || ticket < 0 || ticket >= (int)p->slots.size() * p->gang) return AMK_ERR_INVALID_ARG;
    auto &s = p->slots[ticket % (int)p->slots.size()];
    if (!s.open.empty()) {   // a partly filled gang: launch it now
        const int st = launch_gang(p, s);
        if (st != AMK_OK) return st;
        if (&s == &p->slots[p->next]) p->next = (p->next + 1) % (int)p->slots.size();
    }
    if (s.waited < s.count) {   // the newest event implies all earlier ones (in-order stream)
        const int ws = wait_event(s.done[(s.count - 1) % p->depth]);
        if (ws != AMK_OK) return ws;
        s.waited = s.count;
    }
    return AMK_OK;
}

int amk_pipeline_query(amk_pipeline *p, int ticket) {  // 1 = finished (or idle), 0 = still running or staged, -1 error
    if (!p || ticket < 0 || ticket >= (int)p->slots.size() * p->gang) return -1;
    auto &s = p->slots[ticket % (int)p->slots.size()];
    if (!s.open.empty()) return 0;
    if (s.waited >= s.count) return 1;
    const hipError_t e = hipEventQuery(s.done[(s.count - 1) % p->depth]);
    if (e == hipSuccess) { s.waited = s.count; return 1; }
    return e == hipErrorNotReady ? 0 : -1;
}

int amk_pipeline_drain(amk_pipeline *p) {
    if (!p) return AMK_ERR_INVALID_ARG;
    for (auto &s : p->slots) {   // every open gang first: a launch must not wait for the slots before it to finish
        if (s.open.empty()) continue;
        const int st = launch_gang(p, s);
        if (st != AMK_OK) return st;
        if (&s == &p->slots[p->next]) p->next = (p->next + 1) % (int)p->slots.size();
    }
    for (int i = 0; i < (int)p->slots.size(); ++i) {
        const int st = amk_pipeline_wait(p, i);
        if (st != AMK_OK) return st;
    }
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
