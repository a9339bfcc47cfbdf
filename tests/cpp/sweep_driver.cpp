// C++ multi-process driver of the batched-scenes sweep (BASELINE.json configs[3]): one process per
// GPU, scenes block-partitioned over the ranks, every rank keeps n_slots control steps in flight through amk_pipeline_*,
// ONE exchange step at the end of the sweep (amk_shard_gather: ncclAllGather of the controls over RCCL / xGMI), wall time
// = max over ranks.  The reference runs a single instance (AM/src/mpc_obstacle_avoidance_node.cpp:8); this is the host side
// of its batched counterpart, in C++ as north_star asks.
//
//   sweep_driver <in.bin> <out.bin> <rank> <world> <rendezvous file> [n_slots] [repeat] [gang]
//
// in.bin : int32 total_scenes, n, ne, N, K, mpc_max_iter ; double T, dt, speed, safety_distance ;
//          double weights[25], tau[4], gains[4], limits[5] (aMinZ aMaxZ aMaxXy aMaxYawDot radius) ;
//          per scene: float cloud[n*3], edge[ne*3] ; double state_quad[mpc_max_iter*10], pos_x, ref_path[N*10]
// out.bin (rank 0): int32 total_scenes ; double seconds ; double u[total_scenes][4] ; int32 flags[local scenes of rank 0][4]
// The `repeat` argument re-submits the rank's batch that many times (distinct slots, same frames) to measure steps/s.
//
// -DAMK_SWEEP_STUB builds the same partition / exchange / timing logic over a host-only transport (a shared file in place of
// RCCL, a closed-form "control" in place of the GPU step) so that the world-size-2 path runs where there is no GPU
// (tests/test_sweep_driver.py); without the macro the transport is amk_shard_* and the work is amk_pipeline_* (GPU box,
// world size 1 on a one-GPU box: tests/test_sweep_driver_gpu.py).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "avoid_mpc_amd.h"

#ifndef AMK_SWEEP_STUB
#include <hip/hip_runtime.h>
#endif

template <class T>
static void rd(FILE *f, T *p, size_t n) {
    if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}
#define CHECK(expr)                                                                          \
    do {                                                                                     \
        int _s = (expr);                                                                     \
        if (_s != AMK_OK) { fprintf(stderr, "%s -> %s (hip %d)\n", #expr, amk_status_string(_s), amk_last_hip_error()); exit(3); } \
    } while (0)

// ---- rendezvous through a file: rank 0 publishes a blob, the others wait for it (a real deployment would use MPI / TCP)
static void publish(const std::string &path, const void *p, size_t n) {
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    fwrite(p, 1, n, f);
    fclose(f);
    rename(tmp.c_str(), path.c_str());
}
static void fetch(const std::string &path, void *p, size_t n) {
    for (int tries = 0; tries < 6000; ++tries) {
        FILE *f = fopen(path.c_str(), "rb");
        if (f) {
            const size_t got = fread(p, 1, n, f);
            fclose(f);
            if (got == n) return;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    fprintf(stderr, "rendezvous timed out on %s\n", path.c_str());
    exit(4);
}

#ifdef AMK_SWEEP_STUB
// host-only transport: every rank writes its block to <rendezvous>.<rank>, then reads all of them
struct Transport {
    int rank, world;
    std::string base;
    int round = 0;
    Transport(int r, int w, const std::string &b) : rank(r), world(w), base(b) {}
    void gather(const double *local, long long n, double *all) {
        const std::string tag = base + ".g" + std::to_string(round++) + ".";
        publish(tag + std::to_string(rank), local, sizeof(double) * n);
        for (int r = 0; r < world; ++r) fetch(tag + std::to_string(r), all + (size_t)r * n, sizeof(double) * n);
    }
    double max(double v) {
        std::vector<double> all(world);
        gather(&v, 1, all.data());
        double m = all[0];
        for (double x : all) m = std::fmax(m, x);
        return m;
    }
};
#else
struct Transport {
    amk_shard *sh = nullptr;
    double *d_tmp = nullptr;
    Transport(int rank, int world, const std::string &base) {
        char id[AMK_SHARD_ID_BYTES];
        if (rank == 0) {
            CHECK(amk_shard_unique_id(id));
            publish(base + ".id", id, sizeof id);
        } else {
            fetch(base + ".id", id, sizeof id);
        }
        CHECK(amk_shard_create(id, rank, world, &sh));
        hipMalloc((void **)&d_tmp, sizeof(double));
    }
    double max(double v) {
        hipMemcpy(d_tmp, &v, sizeof v, hipMemcpyHostToDevice);
        CHECK(amk_shard_max(sh, d_tmp, 1, nullptr));
        hipDeviceSynchronize();
        hipMemcpy(&v, d_tmp, sizeof v, hipMemcpyDeviceToHost);
        return v;
    }
    ~Transport() {
        if (sh) amk_shard_destroy(sh);
        if (d_tmp) hipFree(d_tmp);
    }
};
#endif

int main(int argc, char **argv) {
    if (argc < 6) return 1;
    const int rank = atoi(argv[3]), world = atoi(argv[4]);
    const std::string rdv = argv[5];
    const int n_slots = argc > 6 ? atoi(argv[6]) : 4;
    const int repeat = argc > 7 ? atoi(argv[7]) : 1;
    const int gang = argc > 8 ? atoi(argv[8]) : 0;   // frames per launch (amk_pipeline_config.gang)
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int hdr[6];
    rd(f, hdr, 6);
    const int total = hdr[0], n = hdr[1], ne = hdr[2], N = hdr[3], K = hdr[4], max_iter = hdr[5];
    double sc[4];
    rd(f, sc, 4);
    std::vector<double> weights(25), tau(4), gains(4), lim(5);
    rd(f, weights.data(), 25); rd(f, tau.data(), 4); rd(f, gains.data(), 4); rd(f, lim.data(), 5);
    int first = 0, count = 0;
    CHECK(amk_shard_scene_range(rank, world, total, &first, &count));
    const int S = amk_shard_padded_count(world, total);  // equal shards for the collective (ncclAllGather): the last ranks pad with their last scene
    const size_t per_scene = sizeof(float) * 3 * ((size_t)n + ne) + sizeof(double) * ((size_t)max_iter * 10 + 1 + (size_t)N * 10);
    std::vector<float> cl((size_t)S * n * 3), ed((size_t)S * ne * 3);
    std::vector<double> sq((size_t)S * max_iter * 10), posx(S), ref((size_t)S * N * 10);
    const long data0 = ftell(f);
    for (int s = 0; s < S; ++s) {
        const int g = first + (s < count ? s : count - 1);
        fseek(f, data0 + (long)(per_scene * (size_t)g), SEEK_SET);
        rd(f, cl.data() + (size_t)s * n * 3, (size_t)n * 3);
        rd(f, ed.data() + (size_t)s * ne * 3, (size_t)ne * 3);
        rd(f, sq.data() + (size_t)s * max_iter * 10, (size_t)max_iter * 10);
        rd(f, posx.data() + s, 1);
        rd(f, ref.data() + (size_t)s * N * 10, (size_t)N * 10);
    }
    fclose(f);
    std::vector<double> u_local((size_t)S * 4), u_all((size_t)S * 4 * world);
    std::vector<int> flags((size_t)S * 4, 0);
    double seconds = 0.0;

#ifdef AMK_SWEEP_STUB
    Transport tp(rank, world, rdv);
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < S; ++s) {   // stands where the GPU step stands: a closed form of the scene's inputs
        const double *r0 = ref.data() + (size_t)s * N * 10;
        for (int i = 0; i < 4; ++i) u_local[4 * s + i] = cl[(size_t)s * n * 3 + i] + 10.0 * posx[s] + r0[i] + 100.0 * sq[(size_t)s * max_iter * 10 + i];
    }
    tp.gather(u_local.data(), 4LL * S, u_all.data());
    seconds = tp.max(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
#else
    int ndev = 0;
    hipGetDeviceCount(&ndev);
    if (ndev <= 0) { fprintf(stderr, "no GPU\n"); return 5; }
    hipSetDevice(rank % ndev);
    Transport tp(rank, world, rdv);
    amk_pipeline_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.n_slots = n_slots; cfg.n_scenes = S; cfg.max_points = n; cfg.max_edge_points = ne;
    cfg.T = sc[0]; cfg.dt = sc[1]; cfg.nearest_point_num = K; cfg.queue_depth = 0; cfg.gang = gang;
    cfg.step.speed = sc[2]; cfg.step.safety_distance = sc[3]; cfg.step.mpc_max_iter = max_iter;
    amk_pipeline *pl = nullptr;
    CHECK(amk_pipeline_create(&cfg, &pl));
    for (int i = 0; i < n_slots; ++i) {   // what the FSM's constructor does with the yaml (AvoidanceStateMachine.cpp:62-70)
        amk_mpc *m = amk_pipeline_mpc(pl, i);
        CHECK(amk_mpc_setup_weights(m, weights.data()));
        CHECK(amk_mpc_setup_tau(m, tau.data()));
        CHECK(amk_mpc_setup_gains(m, gains.data()));
        CHECK(amk_mpc_set_drone_radius(m, lim[4]));
        CHECK(amk_mpc_set_drone_accel_limits(m, lim[0], lim[1], lim[2], lim[3]));
    }
    float *d_cl, *d_ed;
    double *d_sq, *d_px, *d_ref, *d_u, *d_all;
    hipMalloc((void **)&d_cl, cl.size() * sizeof(float)); hipMalloc((void **)&d_ed, ed.size() * sizeof(float));
    hipMalloc((void **)&d_sq, sq.size() * 8); hipMalloc((void **)&d_px, posx.size() * 8); hipMalloc((void **)&d_ref, ref.size() * 8);
    hipMalloc((void **)&d_u, (size_t)repeat * S * 4 * 8); hipMalloc((void **)&d_all, (size_t)repeat * S * 4 * 8 * world);
    hipMemcpy(d_cl, cl.data(), cl.size() * sizeof(float), hipMemcpyHostToDevice);
    hipMemcpy(d_ed, ed.data(), ed.size() * sizeof(float), hipMemcpyHostToDevice);
    hipMemcpy(d_sq, sq.data(), sq.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_px, posx.data(), posx.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_ref, ref.data(), ref.size() * 8, hipMemcpyHostToDevice);
    amk_pipeline_frame fr;
    std::memset(&fr, 0, sizeof fr);
    fr.d_cloud = d_cl; fr.d_edge = d_ed; fr.point_stride = 3; fr.d_state_quad = d_sq; fr.d_pos_x = d_px; fr.d_ref_path_init = d_ref;
    // warm-up: every slot allocates its step workspace once
    for (int i = 0; i < n_slots * (gang > 1 ? gang : 1); ++i) { fr.d_u_out = nullptr; CHECK(amk_pipeline_submit(pl, &fr, nullptr)); }
    CHECK(amk_pipeline_drain(pl));
    tp.max(0.0);   // barrier
    const auto t0 = std::chrono::steady_clock::now();
    int last_slot = 0;   // the last frame's ticket (= its slot without a gang)
    for (int r = 0; r < repeat; ++r) {
        fr.d_u_out = d_u + (size_t)r * S * 4;
        CHECK(amk_pipeline_submit(pl, &fr, &last_slot));
    }
    CHECK(amk_pipeline_drain(pl));
    // the one exchange step of the sweep: every rank's controls to every rank
    CHECK(amk_shard_gather(tp.sh, d_u, 4LL * S * repeat, d_all, nullptr));
    hipDeviceSynchronize();
    seconds = tp.max(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    std::vector<double> all((size_t)repeat * S * 4 * world);
    hipMemcpy(all.data(), d_all, all.size() * 8, hipMemcpyDeviceToHost);
    for (int r = 0; r < world; ++r)   // first repetition of every rank's block
        std::memcpy(u_all.data() + (size_t)r * S * 4, all.data() + (size_t)r * repeat * S * 4, sizeof(double) * S * 4);
    int *d_flags = nullptr;
    CHECK(amk_pipeline_outputs(pl, last_slot, nullptr, nullptr, &d_flags, nullptr));
    hipMemcpy(flags.data(), d_flags, flags.size() * sizeof(int), hipMemcpyDeviceToHost);
    amk_pipeline_destroy(pl);
    hipFree(d_cl); hipFree(d_ed); hipFree(d_sq); hipFree(d_px); hipFree(d_ref); hipFree(d_u); hipFree(d_all);
#endif
    if (rank == 0) {
        // rank r's block holds scenes [first_r, first_r + count_r) followed by padding: write them in global scene order
        std::vector<double> u_glob((size_t)total * 4);
        for (int r = 0; r < world; ++r) {
            int fr_ = 0, cn = 0;
            CHECK(amk_shard_scene_range(r, world, total, &fr_, &cn));
            std::memcpy(u_glob.data() + (size_t)fr_ * 4, u_all.data() + (size_t)r * S * 4, sizeof(double) * 4 * cn);
        }
        FILE *o = fopen(argv[2], "wb");
        fwrite(&total, sizeof(int), 1, o);
        fwrite(&seconds, sizeof(double), 1, o);
        fwrite(u_glob.data(), sizeof(double), u_glob.size(), o);
        fwrite(flags.data(), sizeof(int), (size_t)count * 4, o);
        fclose(o);
        printf("{\"world\": %d, \"scenes\": %d, \"slots\": %d, \"repeat\": %d, \"seconds\": %.6f, \"scene_steps_per_s\": %.1f}\n", world,
               total, n_slots, repeat, seconds, (double)total * repeat / seconds);
    }
    return 0;
}
