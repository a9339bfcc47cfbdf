// CPU test of include/avoid_mpc_amd/ros_glue.hpp against message-shaped structs (field names of sensor_msgs/Image,
// nav_msgs/Odometry, sensor_msgs/Imu, quadrotor_msgs/Command); prints "key value" lines that tests/test_ros_glue.py checks
// against closed forms of the reference's code (AvoidanceStateMachine.cpp:118-164,369-397, FrameKDMap.cpp:90-101,
// ParameterManager.cpp:59-104).  No GPU call is made: only the field mapping is exercised (FillStepCmd, ImageToDepth,
// ReadParams, DepthPose, OnOdometry, OnImu).
#include <cstdio>
#include <map>
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

int main(int argc, char **argv) {
    // ---- parameters: a getter over a key/value table, as the test's yaml
    std::map<std::string, double> kv;
    for (int i = 1; i + 1 < argc; i += 2) kv[argv[i]] = atof(argv[i + 1]);
    auto get = [&](const std::string &k, double &v) { auto it = kv.find(k); if (it == kv.end()) return false; v = it->second; return true; };
    const ros_glue::Params p = ros_glue::ReadParams(get);
    printf("T %.17g\ndt %.17g\nmaxIter %d\nK %d\nspeed %.17g\nradius %.17g\nsafety %.17g\n", p.T, p.dt, p.maxIter, p.nearestPointNum,
           p.speed, p.droneRadius, p.safetyDistance);
    printf("limits %.17g %.17g %.17g %.17g\n", p.aMinZ, p.aMaxZ, p.aMaxXy, p.aMaxYawDot);
    printf("weights"); for (double w : p.weights) printf(" %.17g", w); printf("\n");
    printf("taus"); for (double w : p.taus) printf(" %.17g", w); printf("\n");
    printf("gains"); for (double w : p.gains) printf(" %.17g", w); printf("\n");
    printf("Tbc"); for (double w : p.Tbc) printf(" %.17g", w); printf("\n");
    const amk_depth_params dp = ros_glue::DepthParams(p);
    printf("depth %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", dp.pixel2meter, dp.depth_min, dp.depth_max, dp.resize_scale, dp.fx, dp.fy, dp.cx, dp.cy);
    const amk_step_params sp = ros_glue::StepParams(p);
    printf("step %.17g %.17g %d\n", sp.speed, sp.safety_distance, sp.mpc_max_iter);

    // ---- commands
    ros_glue::Pose s;
    s.odom.vel[0] = 9.0; s.odom.vel[1] = -40.0; s.odom.vel[2] = 0.5;
    s.odom.acc[0] = 1.0; s.odom.acc[1] = 2.0; s.odom.acc[2] = -30.0;
    Command c;
    const std::vector<double> u = {1.25, -2.5, 9.81, 0.3};
    int flags_ok[4] = {1, 3, 0, 47}, flags_unsafe[4] = {0, 3, 0, 47}, flags_cap[4] = {1, 3, 1, 300};
    bool ok = ros_glue::FillStepCmd(c, flags_ok, u, s.odom, p);
    printf("cmd_ok %d %d %.17g %.17g %.17g %.17g\n", (int)ok, (int)c.mode, c.acceleration.x, c.acceleration.y, c.acceleration.z, c.yaw);
    c = Command();
    ok = ros_glue::FillStepCmd(c, flags_unsafe, u, s.odom, p);
    printf("cmd_unsafe %d %d %.17g %.17g %.17g %.17g\n", (int)ok, (int)c.mode, c.acceleration.x, c.acceleration.y, c.acceleration.z, c.yaw);
    ok = ros_glue::FillStepCmd(c, flags_cap, u, s.odom, p);
    printf("cmd_cap_default %d %.17g\n", (int)ok, c.acceleration.x);
    ok = ros_glue::FillStepCmd(c, flags_cap, u, s.odom, p, true);
    printf("cmd_cap_fallback %d %.17g\n", (int)ok, c.acceleration.x);

    // ---- images: 16UC1 with row padding, big-endian 32FC1, an unsupported encoding
    Image im;
    im.height = 2; im.width = 3; im.encoding = "16UC1"; im.step = 8; im.data.assign(16, 0xEE);
    const uint16_t px[6] = {100, 200, 300, 400, 500, 65535};
    for (int r = 0; r < 2; ++r) memcpy(im.data.data() + 8 * r, px + 3 * r, 6);
    ros_glue::DepthView v = ros_glue::ImageToDepth(im);
    printf("img16 %d %d %d %d", v.type, v.rows, v.cols, (int)!v.owned.empty());
    for (int i = 0; i < 6; ++i) printf(" %u", (unsigned)((const uint16_t *)v.data)[i]);
    printf("\n");
    im.step = 6; im.data.resize(12); memcpy(im.data.data(), px, 12);
    v = ros_glue::ImageToDepth(im);
    printf("img16_packed %d %d\n", (int)v.owned.empty(), (int)(v.data == im.data.data()));
    Image fm;
    fm.height = 1; fm.width = 2; fm.encoding = "32FC1"; fm.step = 8; fm.is_bigendian = 1; fm.data.resize(8);
    const float fv[2] = {1.5f, -0.25f};
    for (int i = 0; i < 2; ++i) for (int b = 0; b < 4; ++b) fm.data[4 * i + b] = ((const uint8_t *)&fv[i])[3 - b];
    v = ros_glue::ImageToDepth(fm);
    printf("img32be %d %.9g %.9g\n", v.type, ((const float *)v.data)[0], ((const float *)v.data)[1]);
    fm.encoding = "rgb8";
    try { ros_glue::ImageToDepth(fm); printf("imgbad accepted\n"); } catch (const std::exception &e) { printf("imgbad rejected\n"); }

    // ---- callbacks
    Odometry od;
    od.pose.pose.position = {1.0, 2.0, 3.0};
    od.pose.pose.orientation.w = 0.9238795325112867; od.pose.pose.orientation.z = 0.3826834323650898;   // yaw = 45 deg
    od.twist.twist.linear = {4.0, 5.0, 6.0};
    ros_glue::Pose q;
    ros_glue::OnOdometry(q, od, 10.0, false);
    printf("odom %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", q.odom.pos[0], q.odom.pos[1], q.odom.pos[2], q.odom.vel[0], q.odom.vel[1], q.odom.vel[2], q.odom.yaw, q.odom.stamp);
    Imu imu;
    const double accb[3] = {1.0, 0.0, 9.81};
    ros_glue::OnImu(q, imu, accb, 10.1, true, false);
    printf("imu %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", q.odom.pos[0], q.odom.vel[0], q.odom.acc[0], q.odom.acc[1], q.odom.acc[2], q.odom.stamp, q.odom.pos[2]);
    double Twb[16];
    ros_glue::DepthPose(q, 10.3, true, Twb);
    printf("Twb"); for (double w : Twb) printf(" %.17g", w); printf("\n");
    return 0;
}
