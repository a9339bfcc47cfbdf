// A C++ fleet host on the C ABI's TASK mode: what a maintainer's node-side code looks like when MANY robots share one GPU.
// Per control period and batch of robots it hands the pipeline what the reference's callbacks hand the state machine -- the
// frame (DepthCallback -> AddVertex, AM/src/AvoidanceStateMachine.cpp:153-164) and the odometry (:118-152) -- and gets the
// command back (PubCmd / PubSlowDownCmd, :345-350,369-397); GetInitPath, GetCurStateQuad, the re-plan loop and the warm start
// live in the slot (include/avoid_mpc_amd.h: amk_pipeline_frame.d_odom).  The vehicle here is the test's: the MPC's own model, RK4 x 4 (avoid_mpc_amd/flight.py: plant_step, statement by statement).
// Driven by tests/test_flight_gpu.py, which flies the same flights through the Python driver and compares bit for bit.
//   flight_driver <in.bin> <out.bin> [max_frame_count]
// With max_frame_count > 0 every robot flies with FrameKDMap's keyframe list in the slot (amk_pipeline_config.keyframes: the
// reference's default regime, FrameKDMap.cpp:29-32): cloud frames then bring mCurFrame.Twc = Twb * T_b_c -- Twb = [I | odometry position],
// T_b_c the yaml's extrinsic (mpc_parameters.yaml:67-71: the camera looks along the body's +x) -- and PtIsInFrame's camera (the yaml's
// 640 x 480 / 10 sensor).
// in.bin : int32 S, P, n, ne, K, max_iter ; double prm[9] = T dt speed safety decay height farest kp kd ; weights[25] tau[4] gains[4]
//          lim[5] (aMinZ aMaxZ aMaxXy aMaxYawDot radius) ; x0[S*10] ; ref0[S*N*10] ;
//          per period: float cloud[S*n*3], edge[S*ne*3]
// out.bin: per period: double x[S*10] (after the vehicle step), cmd[S*3], int32 flags[S*4]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "avoid_mpc_amd.h"

#define CHECK(x) do { int st_ = (x); if (st_ != AMK_OK) { fprintf(stderr, "%s: %s\n", #x, amk_status_string(st_)); return 3; } } while (0)
template <class T>
static void rd(FILE *f, T *p, size_t n) {
    if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

static const double kTbc[16] = {0.0, 0.0, 1.0, 0.05, -1.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0, 0.01, 0.0, 0.0, 0.0, 1.0};   // mpc_parameters.yaml:67-71

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int hdr[6];
    rd(f, hdr, 6);
    const int S = hdr[0], P = hdr[1], n = hdr[2], ne = hdr[3], K = hdr[4], max_iter = hdr[5];
    double prm[9];
    rd(f, prm, 9);
    std::vector<double> weights(25), tau(4), gains(4), lim(5);
    rd(f, weights.data(), 25); rd(f, tau.data(), 4); rd(f, gains.data(), 4); rd(f, lim.data(), 5);
    const int N = (int)(prm[0] / prm[1]);
    std::vector<double> x((size_t)S * 10), ref0((size_t)S * N * 10);
    rd(f, x.data(), x.size()); rd(f, ref0.data(), ref0.size());

    amk_pipeline_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.n_slots = 1; cfg.n_scenes = S; cfg.max_points = n; cfg.max_edge_points = ne;
    cfg.T = prm[0]; cfg.dt = prm[1]; cfg.nearest_point_num = K; cfg.queue_depth = 1; cfg.gang = 1;
    cfg.step.speed = prm[2]; cfg.step.safety_distance = prm[3]; cfg.step.mpc_max_iter = max_iter;
    cfg.task.decay = prm[4]; cfg.task.iter_time = 0.0; cfg.task.height = prm[5]; cfg.task.farest_point = prm[6];
    cfg.task.slow_down_kp = prm[7]; cfg.task.slow_down_kd = prm[8]; cfg.task.a_max_xy = lim[2]; cfg.task.a_max_z = lim[1];
    cfg.task.use_odom_est = 1;
    const int max_frames = argc > 3 ? atoi(argv[3]) : 0;
    if (max_frames > 0) {   // mpc_parameters.yaml:66,71-73
        cfg.keyframes.max_frame_count = max_frames; cfg.keyframes.keyframe_th_count = 10; cfg.keyframes.keyframe_th_dist = 0.1;
        cfg.keyframes.depth_min = 0.1;
        for (int e = 0; e < 16; ++e) cfg.keyframes.Tbc[e] = kTbc[e];   // mParamTbc
    }
    amk_frame_camera cam;
    cam.fx = 32.0; cam.fy = 32.0; cam.cx = 32.0; cam.cy = 24.0; cam.depth_max = 100.0; cam.width = 64; cam.height = 48;
    amk_pipeline *pl = nullptr;
    CHECK(amk_pipeline_create(&cfg, &pl));
    amk_mpc *m = amk_pipeline_mpc(pl, 0);   // SetupMPC (AvoidanceStateMachine.cpp:55-70)
    CHECK(amk_mpc_setup_weights(m, weights.data())); CHECK(amk_mpc_setup_tau(m, tau.data())); CHECK(amk_mpc_setup_gains(m, gains.data()));
    CHECK(amk_mpc_set_drone_radius(m, lim[4])); CHECK(amk_mpc_set_drone_accel_limits(m, lim[0], lim[1], lim[2], lim[3]));

    float *d_cl, *d_ed;
    double *d_x, *d_ref0, *d_cmd;
    hipMalloc((void **)&d_cl, sizeof(float) * S * n * 3); hipMalloc((void **)&d_ed, sizeof(float) * S * ne * 3);
    hipMalloc((void **)&d_x, sizeof(double) * S * 10); hipMalloc((void **)&d_ref0, sizeof(double) * S * N * 10);
    hipMalloc((void **)&d_cmd, sizeof(double) * S * 3);
    double *d_Twc = nullptr;
    std::vector<double> Twc((size_t)S * 16, 0.0);
    if (max_frames > 0) hipMalloc((void **)&d_Twc, sizeof(double) * S * 16);
    hipMemcpy(d_ref0, ref0.data(), sizeof(double) * ref0.size(), hipMemcpyHostToDevice);
    std::vector<float> cl((size_t)S * n * 3), ed((size_t)S * ne * 3);
    std::vector<double> cmd((size_t)S * 3), xn((size_t)S * 10);
    std::vector<int> flags((size_t)S * 4);
    FILE *o = fopen(argv[2], "wb");
    for (int t = 0; t < P; ++t) {
        rd(f, cl.data(), cl.size()); rd(f, ed.data(), ed.size());
        hipMemcpy(d_cl, cl.data(), sizeof(float) * cl.size(), hipMemcpyHostToDevice);       // the frame ...
        hipMemcpy(d_ed, ed.data(), sizeof(float) * ed.size(), hipMemcpyHostToDevice);
        hipMemcpy(d_x, x.data(), sizeof(double) * x.size(), hipMemcpyHostToDevice);         // ... and the odometry of every robot
        amk_pipeline_frame fr;
        std::memset(&fr, 0, sizeof fr);
        fr.d_cloud = d_cl; fr.d_edge = d_ed; fr.point_stride = 3;
        fr.d_odom = d_x; fr.d_cmd_out = d_cmd; fr.keep_warm_start = t > 0;
        fr.d_ref_path_init = t == 0 ? d_ref0 : nullptr;     // InitCircleState's role; afterwards the slot's own mRefPath
        if (max_frames > 0) {   // mCurFrame.Twc of the frame (what DepthCallback's pose would give): [I | odometry position] * T_b_c
            for (int s = 0; s < S; ++s) {
                double *T = &Twc[16 * (size_t)s];
                for (int e = 0; e < 16; ++e) T[e] = kTbc[e];
                T[3] += x[10 * s]; T[7] += x[10 * s + 1]; T[11] += x[10 * s + 2];
            }
            hipMemcpy(d_Twc, Twc.data(), sizeof(double) * Twc.size(), hipMemcpyHostToDevice);
            fr.d_Twc_cur = d_Twc; fr.camera = &cam;
        }
        int ticket = -1;
        CHECK(amk_pipeline_submit(pl, &fr, &ticket));
        CHECK(amk_pipeline_wait(pl, ticket));
        int *d_flags = nullptr;
        CHECK(amk_pipeline_outputs(pl, ticket, nullptr, nullptr, &d_flags, nullptr));
        hipMemcpy(cmd.data(), d_cmd, sizeof(double) * cmd.size(), hipMemcpyDeviceToHost);
        hipMemcpy(flags.data(), d_flags, sizeof(int) * flags.size(), hipMemcpyDeviceToHost);
        for (int s = 0; s < S; ++s) {   // the vehicle over one control period (yaw_dot = 0: Command.yaw = 0 holds the heading)
            const double u[4] = {cmd[3 * s], cmd[3 * s + 1], cmd[3 * s + 2], 0.0};
            auto fx = [&](const double *xx, double *d) {   // mpc_obstacle_casadi.py:106-122
                d[0] = xx[4]; d[1] = xx[5]; d[2] = xx[6]; d[3] = u[3]; d[4] = xx[7]; d[5] = xx[8]; d[6] = xx[9];
                d[7] = (u[0] - xx[7]) * tau[0]; d[8] = (u[1] - xx[8]) * tau[1]; d[9] = (u[2] - 9.81 - xx[9]) * tau[2];
            };
            double X[10], k1[10], k2[10], k3[10], k4[10], tmp[10];
            std::memcpy(X, &x[10 * s], sizeof X);
            const double h = prm[1] / 4;
            for (int r = 0; r < 4; ++r) {   // :338-357
                fx(X, k1); for (int i = 0; i < 10; ++i) { k1[i] = h * k1[i]; tmp[i] = X[i] + 0.5 * k1[i]; }
                fx(tmp, k2); for (int i = 0; i < 10; ++i) { k2[i] = h * k2[i]; tmp[i] = X[i] + 0.5 * k2[i]; }
                fx(tmp, k3); for (int i = 0; i < 10; ++i) { k3[i] = h * k3[i]; tmp[i] = X[i] + k3[i]; }
                fx(tmp, k4); for (int i = 0; i < 10; ++i) { k4[i] = h * k4[i]; X[i] = X[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6; }
            }
            std::memcpy(&xn[10 * s], X, sizeof X);
        }
        x = xn;
        fwrite(x.data(), sizeof(double), x.size(), o); fwrite(cmd.data(), sizeof(double), cmd.size(), o);
        fwrite(flags.data(), sizeof(int), flags.size(), o);
    }
    fclose(o); fclose(f);
    CHECK(amk_pipeline_destroy(pl));
    return 0;
}
