// amk_pipeline: several independent control steps in flight on one GPU, behind the C ABI.
//
// A control step of this library (index builds of a fresh depth frame + amk_step_batch) is a chain of dependent launches
// whose dominant kernel is latency-bound: one stream fills ~1/8 of the chip.  Consecutive frames of
// a fleet of robots (or the batches of a sweep) are independent, so the throughput configuration keeps n_slots steps in
// flight, each on its own HIP stream with its own handles -- until round 2 that orchestration lived only in bench.py.
// A slot = {stream, obstacle index, edge index, MPC batch (warm start, workspace), reference-path buffer, outputs, event}.
// The reference has one robot and one step in flight (AM/src/mpc_obstacle_avoidance_node.cpp:8,
// AvoidanceStateMachine.cpp:322-355); this is the batched counterpart of its per-frame sequence
// FrameKDMap::AddVertex (FrameKDMap.cpp:34-52) -> AvoidanceStateMachine::Step.
#include "mpc_handle.h"

#include <cstdlib>
#include <cstring>
#include <vector>

struct amk_pipeline {
    amk_pipeline_config cfg;
    struct Slot {
        hipStream_t stream = nullptr;
        std::vector<hipEvent_t> done;   // ring of queue_depth events: done[k % depth] marks the end of the slot's k-th step
        long long count = 0;            // steps submitted on this slot
        long long waited = 0;           // steps known to have finished
        amk_kd *obstacle = nullptr, *edge = nullptr;
        amk_mpc *mpc = nullptr;
        amk::DevBuf<double> ref_path, u, x0array;
        amk::DevBuf<int> flags;
    };
    int depth = 1;
    std::vector<Slot> slots;
    int next = 0;
    long long submitted = 0;
};

namespace {
// Waits for a slot's step.  hipEventSynchronize parks the thread on an HSA signal; with AMK_PIPELINE_SPIN=1 the thread polls
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
}  // namespace

extern "C" {

int amk_pipeline_create(const amk_pipeline_config *cfg, amk_pipeline **out) {
    if (!cfg || !out || cfg->n_slots <= 0 || cfg->n_slots > AMK_PIPELINE_MAX_SLOTS || cfg->n_scenes <= 0 ||
        cfg->max_points <= 0 || cfg->max_edge_points <= 0)
        return AMK_ERR_INVALID_ARG;
    *out = nullptr;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    amk_pipeline *p = new amk_pipeline();
    p->cfg = *cfg;
    p->depth = cfg->queue_depth > 0 ? cfg->queue_depth : AMK_PIPELINE_DEFAULT_DEPTH;
    if (p->depth > AMK_PIPELINE_MAX_DEPTH) p->depth = AMK_PIPELINE_MAX_DEPTH;
    p->slots.resize(cfg->n_slots);
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
        if ((st = amk_kd_create(cfg->n_scenes, cfg->max_points, &s.obstacle)) != AMK_OK) break;
        if ((st = amk_kd_create(cfg->n_scenes, cfg->max_edge_points, &s.edge)) != AMK_OK) break;
        if ((st = amk_mpc_create(cfg->T, cfg->dt, cfg->nearest_point_num, cfg->n_scenes, &s.mpc)) != AMK_OK) break;
        const size_t S = cfg->n_scenes, N = amk_mpc_horizon(s.mpc);
        if ((e = s.ref_path.alloc(S * N * 10)) != hipSuccess || (e = s.u.alloc(S * 4)) != hipSuccess ||
            (e = s.x0array.alloc(S * N * 14)) != hipSuccess || (e = s.flags.alloc(S * 4)) != hipSuccess) {
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

int amk_pipeline_submit(amk_pipeline *p, const amk_pipeline_frame *f, int *slot_out) {
    if (!p || !f || !f->d_cloud || !f->d_edge || !f->d_state_quad || !f->d_pos_x || !f->d_ref_path_init)
        return AMK_ERR_INVALID_ARG;
    const int si = p->next;
    auto &s = p->slots[si];
    // Flow control.  A slot's steps are ordered by its stream, so queuing the next step behind a running one is safe (same
    // handles, same workspaces, in order); submit() only blocks when `depth` steps of this slot are still unfinished --
    // the event about to be re-recorded belongs to the step submitted `depth` submits ago.  depth 1 = at most one step per
    // slot on the device (every step then pays the host's reaction time between its predecessor's end and its own start).
    if (s.count - s.waited >= p->depth) {
        const int ws = wait_event(s.done[s.count % p->depth]);
        if (ws != AMK_OK) return ws;
        s.waited = s.count - p->depth + 1;
    }
    const amk_pipeline_config &c = p->cfg;
    const int N = amk_mpc_horizon(s.mpc);
    const size_t S = c.n_scenes;
    const int stride = f->point_stride ? f->point_stride : 3;
    // fresh frame: mRefPath after GetInitPath, zero warm start unless the caller carries it over (HighLvlMpc.cpp:26-27,35,129)
    AMK_HIP(hipMemcpyAsync(s.ref_path.p, f->d_ref_path_init, sizeof(double) * S * N * 10, hipMemcpyDeviceToDevice, s.stream));
    int st = AMK_OK;
    if (!f->keep_warm_start && (st = amk_mpc_reset_warm_start(s.mpc, s.stream)) != AMK_OK) return st;
    // FrameKDMap::AddVertex: obstacle index and edge index of the frame (FrameKDMap.cpp:44-47)
    if ((st = amk_kd_build_pair(s.obstacle, f->d_cloud, f->d_cloud_counts, s.edge, f->d_edge, f->d_edge_counts, stride, s.stream)) != AMK_OK)
        return st;
    double *u = f->d_u_out ? f->d_u_out : s.u.p;
    if ((st = amk_step_batch(s.obstacle, s.edge, s.mpc, &c.step, f->d_state_quad, f->d_pos_x, s.ref_path.p, u, s.x0array.p,
                             s.flags.p, s.stream)) != AMK_OK)
        return st;
    AMK_HIP(hipEventRecord(s.done[s.count % p->depth], s.stream));
    ++s.count;
    p->next = (si + 1) % (int)p->slots.size();
    ++p->submitted;
    if (slot_out) *slot_out = si;
    return AMK_OK;
}

int amk_pipeline_wait(amk_pipeline *p, int slot) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return AMK_ERR_INVALID_ARG;
    auto &s = p->slots[slot];
    if (s.waited < s.count) {   // the newest event implies all earlier ones (in-order stream)
        const int ws = wait_event(s.done[(s.count - 1) % p->depth]);
        if (ws != AMK_OK) return ws;
        s.waited = s.count;
    }
    return AMK_OK;
}

int amk_pipeline_query(amk_pipeline *p, int slot) {  // 1 = finished (or idle), 0 = still running
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return -1;
    auto &s = p->slots[slot];
    if (s.waited >= s.count) return 1;
    const hipError_t e = hipEventQuery(s.done[(s.count - 1) % p->depth]);
    if (e == hipSuccess) { s.waited = s.count; return 1; }
    return e == hipErrorNotReady ? 0 : -1;
}

int amk_pipeline_drain(amk_pipeline *p) {
    if (!p) return AMK_ERR_INVALID_ARG;
    for (int i = 0; i < (int)p->slots.size(); ++i) {
        const int st = amk_pipeline_wait(p, i);
        if (st != AMK_OK) return st;
    }
    return AMK_OK;
}

int amk_pipeline_outputs(amk_pipeline *p, int slot, double **d_u, double **d_x0array, int **d_flags, double **d_ref_path) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return AMK_ERR_INVALID_ARG;
    auto &s = p->slots[slot];
    if (d_u) *d_u = s.u.p;
    if (d_x0array) *d_x0array = s.x0array.p;
    if (d_flags) *d_flags = s.flags.p;
    if (d_ref_path) *d_ref_path = s.ref_path.p;
    return AMK_OK;
}

}  // extern "C"
