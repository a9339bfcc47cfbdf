// SURVEY.md section 8 row f4, end to end in ONE process: message-shaped sensor_msgs/Image (16UC1 with row padding or 32FC1),
// nav_msgs/Odometry and sensor_msgs/Imu go through include/avoid_mpc_amd/ros_glue.hpp into the reference-shaped classes and
// come out as a quadrotor_msgs/Command -- the sequence the reference's node runs per control period:
//   OdomCallback / IMUCallback (AM/src/AvoidanceStateMachine.cpp:118-152) -> DepthCallback -> FrameKDMap::AddVertex (:153-164;
//   AM/src/FrameKDMap.cpp:34-52: ProcessDepth + BuildEdgeCloud + two InitializeNew, all on the device) -> Step, TASK branch
//   (:322-355: GetInitPath, <= mpc_max_iter re-plan passes) -> PubCmd / PubSlowDownCmd (:345-350,369-397).
// Driven by tests/test_f4_e2e_gpu.py, which runs the oracle chain (depth_oracle -> kd_oracle -> step_oracle) on the same
// messages and compares what is written here.
//   f4_e2e <in.bin> <out.bin>
// in.bin : int32 n_frames, rows, cols, depth_type (0 = 16UC1, 1 = 32FC1), row_padding_bytes, max_iter, K ;
//          double prm[14] = T dt speed safety decay height pixel2meter depth_min depth_max resize_scale fx fy cx cy ;
//          double Tbc[16], weights[25], tau[4], gains[4], lim[5] (aMinZ aMaxZ aMaxXy aMaxYawDot radius), kpkd[2], farest ;
//          double ref_path0[N*10] ;
//          per frame: double odom_pos[3], odom_quat[4] (w x y z), odom_vel[3], t_odom, accb[3], t_imu, t_depth, t_step ;
//                     uint8 image bytes [rows * step]
// out.bin: per frame: int32 have_frame, n_cloud, n_edge ; int32 flags[4] ; double u[4] ; int32 cmd_mode ; double cmd_acc[3], cmd_yaw ;
//          double ref_path[N*10] ; double Twb[16]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "avoid_mpc_amd/ros_glue.hpp"

using namespace avoid_mpc_amd;

struct Vec3 { double x = 0, y = 0, z = 0; };
struct Quat { double x = 0, y = 0, z = 0, w = 1; };
struct Header { double stamp = 0; };
struct Command {   // quadrotor_msgs/Command
    Header header;
    uint8_t mode = 0;
    Vec3 position, velocity, angularVel, acceleration, jerk;
    Quat quat;
    double yaw = -1, yaw_dot = 0, thrust = 0;
    enum { POSITION_MODE = 0, ACCELERATION_MODE = 1, ANGULAR_MODE = 2, QUAT_MODE = 3 };
};
struct Image {     // sensor_msgs/Image
    Header header;
    uint32_t height = 0, width = 0;
    std::string encoding;
    uint8_t is_bigendian = 0;
    uint32_t step = 0;
    std::vector<uint8_t> data;
};
struct Odometry { struct { struct { Vec3 position; Quat orientation; } pose; } pose; struct { struct { Vec3 linear; } twist; } twist; };
struct Imu { Quat orientation; Vec3 linear_acceleration; };

template <class T>
static void rd(FILE *f, T *p, size_t n) {
    if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}
template <class T>
static void wr(FILE *f, const T *p, size_t n) { fwrite(p, sizeof(T), n, f); }

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int hdr[7];
    rd(f, hdr, 7);
    const int n_frames = hdr[0], rows = hdr[1], cols = hdr[2], dtype = hdr[3], pad = hdr[4], max_iter = hdr[5], K = hdr[6];
    double sc[14];
    rd(f, sc, 14);
    ros_glue::Params p;
    p.T = sc[0]; p.dt = sc[1]; p.speed = sc[2]; p.safetyDistance = sc[3]; p.decay = sc[4]; p.height = sc[5];
    p.pixel2Meter = sc[6]; p.depthMin = sc[7]; p.depthMax = sc[8]; p.resizeScale = sc[9]; p.fx = sc[10]; p.fy = sc[11]; p.cx = sc[12]; p.cy = sc[13];
    p.maxIter = max_iter; p.nearestPointNum = K;
    rd(f, p.Tbc, 16);
    p.weights.resize(25); p.taus.resize(4); p.gains.resize(4);
    rd(f, p.weights.data(), 25); rd(f, p.taus.data(), 4); rd(f, p.gains.data(), 4);
    double lim[5], kpkd[2];
    rd(f, lim, 5); rd(f, kpkd, 2); rd(f, &p.farestPoint, 1);
    p.aMinZ = lim[0]; p.aMaxZ = lim[1]; p.aMaxXy = lim[2]; p.aMaxYawDot = lim[3]; p.droneRadius = lim[4];
    p.slowDownKp = kpkd[0]; p.slowDownKd = kpkd[1];
    const int N = (int)(p.T / p.dt);

    // the objects the node owns (AvoidanceStateMachine's members): the map, the MPC + TASK step, the pose the callbacks keep
    FrameKDMap map;
    map.depthParams = ros_glue::DepthParams(p);
    AvoidanceTaskStep task(p.T, p.dt, p.nearestPointNum, p.maxIter, p.speed, p.safetyDistance, p.decay, p.height);
    ros_glue::ConfigureMpc(task.mpc(), p);
    task.useOdomEstimate = p.useOdomEst;
    rd(f, task.RefPath().data(), (size_t)N * 10);   // InitCircleState's role (:14-23): mRefPath before the first GetInitPath
    ros_glue::Pose pose;

    FILE *o = fopen(argv[2], "wb");
    const int bpp = dtype == 0 ? 2 : 4;
    for (int fr = 0; fr < n_frames; ++fr) {
        double m[17];   // odom_pos[3] quat[4] vel[3] t_odom accb[3] t_imu t_depth t_step
        rd(f, m, 17);
        Image img;
        img.height = rows; img.width = cols; img.encoding = dtype == 0 ? "16UC1" : "32FC1";
        img.step = (uint32_t)(cols * bpp + pad);
        img.data.resize((size_t)img.step * rows);
        rd(f, img.data.data(), img.data.size());
        // ---- the callbacks, in the order the messages arrive
        Odometry od;
        od.pose.pose.position = {m[0], m[1], m[2]};
        od.pose.pose.orientation.w = m[3]; od.pose.pose.orientation.x = m[4]; od.pose.pose.orientation.y = m[5]; od.pose.pose.orientation.z = m[6];
        od.twist.twist.linear = {m[7], m[8], m[9]};
        ros_glue::OnOdometry(pose, od, m[10], p.onlyTrustVel);
        Imu imu;
        const double accb[3] = {m[11], m[12], m[13]};
        ros_glue::OnImu(pose, imu, accb, m[14], p.useOdomEst, p.onlyTrustVel);
        double Twb[16];
        ros_glue::DepthPose(pose, m[15], p.useOdomEst, Twb);
        ros_glue::OnDepth(map, pose, img, m[15], p.useOdomEst);       // Image -> AddVertex: depth -> both clouds -> both indices
        // ---- Step, TASK branch
        const auto &cur = map.CurFrame();
        int have = cur.pointCloud ? 1 : 0;
        int counts[2] = {have ? (int)cur.pointCloud->GetPointCloud().pts.size() : 0, have && cur.edgeCloud ? (int)cur.edgeCloud->GetPointCloud().pts.size() : 0};
        int flags[4] = {0, 0, -1, 0};
        std::vector<double> u(4, 0.0);
        std::vector<std::vector<double>> x0Array;
        Command cmd;
        if (have) {
            task.GetInitPath(pose.odom, p.farestPoint);
            task.Step(map, pose.odom, u, x0Array, -1.0, m[16]);
            std::memcpy(flags, task.Flags(), sizeof flags);
            ros_glue::FillStepCmd(cmd, flags, u, pose.odom, p);
        }
        wr(o, &have, 1); wr(o, counts, 2); wr(o, flags, 4); wr(o, u.data(), 4);
        int mode = cmd.mode;
        double acc[4] = {cmd.acceleration.x, cmd.acceleration.y, cmd.acceleration.z, cmd.yaw};
        wr(o, &mode, 1); wr(o, acc, 4);
        wr(o, task.RefPath().data(), (size_t)N * 10);
        wr(o, Twb, 16);
    }
    fclose(o);
    fclose(f);
    return 0;
}
