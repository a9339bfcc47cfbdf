// ROS1 glue of the hot path: the wire formats on either side of the MI355X path, header-only
// and templated on the message types so that it compiles against real ROS messages where ROS exists and against
// message-shaped structs where it does not (tests/cpp/ros_glue_test.cpp; there is no ROS in the build image).
//
//   in   sensor_msgs/Image (32FC1 / 16UC1 depth)  -> raw depth buffer for FrameKDMap::AddVertex      AM/src/FrameKDMap.cpp:34-52,90-105
//        nav_msgs/Odometry, sensor_msgs/Imu       -> OdomState (the callbacks)                        AM/src/AvoidanceStateMachine.cpp:118-152
//        the yaml / ROS parameter keys            -> amk_* setters, amk_depth_params, step params      AM/src/ParameterManager.cpp:13-104,
//                                                                                                      AM/config/mpc_parameters.yaml:1-84
//   out  quadrotor_msgs/Command, ACCELERATION_MODE: PubCmd / PubSlowDownCmd                            AM/src/AvoidanceStateMachine.cpp:369-397
//        (betaflight_ctrl/quadrotor_msgs/msg/Command.msg: uint8 mode, Vector3 acceleration, float64 yaw, ACCELERATION_MODE = 1)
// AM = roswrapper/ros/src/avoid_mpc of the reference tree.  Nothing here touches the GPU; it only maps fields.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "avoidance_step.hpp"

namespace avoid_mpc_amd {
namespace ros_glue {

// ---------------------------------------------------------------------------------------------------------------------------
// parameters: the keys of AM/config/mpc_parameters.yaml as ParameterManager reads them
// ---------------------------------------------------------------------------------------------------------------------------
struct Params {
    // SetupConParam (ParameterManager.cpp:59-104)
    double T = 1.0, dt = 0.033, conDt = 0.02;
    int maxIter = 3, nearestPointNum = 3;
    std::vector<double> weights, taus, gains;   // 25 / 4 / 4 in the weightsName / tausName / gainsName order (:63-85)
    double speed = 10.0, droneRadius = 0.5, safetyDistance = 0.2;
    double aMinZ = 5.0, aMaxZ = 15.0, aMaxXy = 10.0, aMaxYawDot = 10.0;
    double decay = 0.015;
    bool useOdomEst = true, onlyTrustVel = false;
    double slowDownKp = 0.3, slowDownKd = 0.3;
    // SetupPerceptionParam (:13-57)
    double fx = 320, fy = 320, cx = 320, cy = 240, pixel2Meter = 1, depthMax = 100, depthMin = 0.1, resizeScale = 10;
    double Tbc[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double keyframeDistanceTh = 0.1;
    int keyframeCountTh = 10, maxFrameCount = 100;
    // SetupTaskParam (:105-110)
    double height = 1.5, farestPoint = 500;
};

inline const std::vector<std::string> &WeightNames() {  // ParameterManager.cpp:63-68
    static const std::vector<std::string> n = {"goal_p_x", "goal_p_y", "goal_p_z", "goal_yaw", "goal_v_x", "goal_v_y", "goal_v_z",
                                                "goal_a_x", "goal_a_y", "goal_a_z", "path_p_x", "path_p_y", "path_p_z", "path_yaw",
                                                "path_v_x", "path_v_y", "path_v_z", "path_a_x", "path_a_y", "path_a_z", "u_a_x",
                                                "u_a_y", "u_a_z", "u_yaw_dot", "collide_lambda"};
    return n;
}

// `get(key, double&) -> bool` stands where ros::NodeHandle::getParam stands; a missing key keeps the default, as the
// reference's unchecked getParam calls do (its defaults are whatever the struct held).  T_b_c: 16 keys "T_b_c/r/c" or, when
// the source has a matrix accessor, fill Params::Tbc directly.
template <class Getter>
Params ReadParams(Getter &&get) {
    Params p;
    auto num = [&](const char *k, double &v) { double t; if (get(std::string(k), t)) v = t; };
    auto integer = [&](const char *k, int &v) { double t; if (get(std::string(k), t)) v = (int)t; };
    auto flag = [&](const char *k, bool &v) { double t; if (get(std::string(k), t)) v = t != 0.0; };
    num("mpc_dt", p.dt); num("mpc_T", p.T); num("con_dt", p.conDt);
    integer("mpc_max_iter", p.maxIter); integer("nearest_point_num", p.nearestPointNum);
    for (const std::string &k : WeightNames()) { double w = 0; get(k, w); p.weights.push_back(w); }
    for (const char *k : {"tau_a_x", "tau_a_y", "tau_a_z", "tau_yaw_dot"}) { double w = 0; get(std::string(k), w); p.taus.push_back(w); }
    for (const char *k : {"gain_a_x", "gain_a_y", "gain_a_z", "gain_yaw_dot"}) { double w = 0; get(std::string(k), w); p.gains.push_back(w); }
    num("speed", p.speed); num("drone_radius", p.droneRadius); num("safety_distance", p.safetyDistance);
    num("a_min_z", p.aMinZ); num("a_max_z", p.aMaxZ); num("a_max_xy", p.aMaxXy); num("a_max_yaw_dot", p.aMaxYawDot);
    num("decay", p.decay); flag("use_odom_est", p.useOdomEst); flag("only_trust_vel", p.onlyTrustVel);
    num("slow_down_kp", p.slowDownKp); num("slow_down_kd", p.slowDownKd);
    num("fx", p.fx); num("fy", p.fy); num("cx", p.cx); num("cy", p.cy); num("pixel2meter", p.pixel2Meter);
    num("depth_max", p.depthMax); num("depth_min", p.depthMin); num("resize_scale", p.resizeScale);
    num("keyframe_th_dist", p.keyframeDistanceTh); integer("keyframe_th_count", p.keyframeCountTh);
    integer("max_frame_count", p.maxFrameCount);
    num("height", p.height); num("goal_x", p.farestPoint);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double t;
            if (get("T_b_c/" + std::to_string(r) + "/" + std::to_string(c), t)) p.Tbc[4 * r + c] = t;
        }
    return p;
}

// SetupMPC (AvoidanceStateMachine.cpp:55-70) on a handle of the C ABI.
inline void ConfigureMpc(amk_mpc *mpc, const Params &p) {
    if (p.weights.size() != 25 || p.taus.size() != 4 || p.gains.size() != 4) throw std::runtime_error("ros_glue: weights/taus/gains sizes");
    amk_throw(amk_mpc_setup_weights(mpc, p.weights.data()), "SetupWeights");
    amk_throw(amk_mpc_setup_tau(mpc, p.taus.data()), "SetupTau");
    amk_throw(amk_mpc_setup_gains(mpc, p.gains.data()), "SetupGains");
    amk_throw(amk_mpc_set_drone_radius(mpc, p.droneRadius), "SetDroneRadius");
    amk_throw(amk_mpc_set_drone_accel_limits(mpc, p.aMinZ, p.aMaxZ, p.aMaxXy, p.aMaxYawDot), "SetDroneAccelLimits");
}
inline amk_step_params StepParams(const Params &p) {
    amk_step_params s;
    s.speed = p.speed; s.safety_distance = p.safetyDistance; s.mpc_max_iter = p.maxIter; s.reserved = 0;
    return s;
}
inline amk_depth_params DepthParams(const Params &p) {  // FrameKDMap's constructor (FrameKDMap.cpp:6-25): full-resolution intrinsics
    amk_depth_params d;
    d.pixel2meter = p.pixel2Meter; d.depth_min = p.depthMin; d.depth_max = p.depthMax; d.resize_scale = p.resizeScale;
    d.fx = p.fx; d.fy = p.fy; d.cx = p.cx; d.cy = p.cy;
    std::memcpy(d.Tbc, p.Tbc, sizeof d.Tbc);
    return d;
}

// ---------------------------------------------------------------------------------------------------------------------------
// in: sensor_msgs/Image -> what FrameKDMap::AddVertex(Twb, depth buffer) takes
// ---------------------------------------------------------------------------------------------------------------------------
struct DepthView {
    const void *data = nullptr;   // tightly packed rows x cols, host byte order
    int type = -1;                // AMK_DEPTH_U16 / AMK_DEPTH_F32
    int rows = 0, cols = 0;
    std::vector<uint8_t> owned;   // filled when the message had to be repacked (row padding or foreign endianness)
};

// Image: anything with .height, .width, .encoding (string), .is_bigendian, .step, .data (byte vector) -- sensor_msgs/Image.
// The reference goes through cv_bridge::toCvCopy and accepts CV_16UC1 and CV_32FC1 only (FrameKDMap.cpp:92-101):
// encodings "16UC1" / "mono16" and "32FC1"; anything else is the reference's "depth image type not supported".
template <class Image>
DepthView ImageToDepth(const Image &img) {
    DepthView v;
    const std::string enc = img.encoding;
    int bpp = 0;
    if (enc == "16UC1" || enc == "mono16") { v.type = AMK_DEPTH_U16; bpp = 2; }
    else if (enc == "32FC1") { v.type = AMK_DEPTH_F32; bpp = 4; }
    else throw std::runtime_error("ros_glue: depth image type not supported: " + enc);
    v.rows = (int)img.height; v.cols = (int)img.width;
    const size_t row_bytes = (size_t)v.cols * bpp, step = (size_t)img.step;
    if (step < row_bytes || img.data.size() < step * (size_t)v.rows) throw std::runtime_error("ros_glue: short depth image");
    const uint16_t probe = 1;
    const bool host_big = *reinterpret_cast<const uint8_t *>(&probe) == 0;
    const bool swap = (img.is_bigendian != 0) != host_big;
    if (step == row_bytes && !swap) {
        v.data = img.data.data();
        return v;
    }
    v.owned.resize(row_bytes * (size_t)v.rows);
    for (int r = 0; r < v.rows; ++r) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(img.data.data()) + step * (size_t)r;
        uint8_t *dst = v.owned.data() + row_bytes * (size_t)r;
        if (!swap) std::memcpy(dst, src, row_bytes);
        else
            for (size_t i = 0; i < row_bytes; i += bpp)
                for (int b = 0; b < bpp; ++b) dst[i + b] = src[i + bpp - 1 - b];
    }
    v.data = v.owned.data();
    return v;
}

// yaw of a quaternion, GetYawFromPuat (AvoidanceStateMachine.cpp:112-116)
inline double YawFromQuat(double qw, double qx, double qy, double qz) {
    return std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
}
inline void QuatToRot(double w, double x, double y, double z, double R[9]) {  // Eigen::Quaterniond::toRotationMatrix
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x,
                 txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// What the odometry / IMU callbacks keep (AvoidanceStateMachine.cpp:118-152) beyond OdomState: the attitude.
struct Pose {
    OdomState odom;
    double quat[4] = {1, 0, 0, 0};   // w x y z
};

// OdomCallback (:118-134).  Odometry: nav_msgs/Odometry-shaped.
template <class Odometry>
void OnOdometry(Pose &s, const Odometry &msg, double now, bool onlyTrustVel) {
    s.odom.stamp = now;
    if (!onlyTrustVel) {
        s.odom.pos[0] = msg.pose.pose.position.x; s.odom.pos[1] = msg.pose.pose.position.y; s.odom.pos[2] = msg.pose.pose.position.z;
        s.quat[0] = msg.pose.pose.orientation.w; s.quat[1] = msg.pose.pose.orientation.x;
        s.quat[2] = msg.pose.pose.orientation.y; s.quat[3] = msg.pose.pose.orientation.z;
    } else {
        s.odom.pos[0] = s.odom.pos[1] = s.odom.pos[2] = 0.0;
    }
    s.odom.vel[0] = msg.twist.twist.linear.x; s.odom.vel[1] = msg.twist.twist.linear.y; s.odom.vel[2] = msg.twist.twist.linear.z;
    s.odom.yaw = YawFromQuat(s.quat[0], s.quat[1], s.quat[2], s.quat[3]);
}

// IMUCallback (:136-152) after the COG filter (the filter itself is outside the hot path): accb is the
// filtered body-frame acceleration.
template <class Imu>
void OnImu(Pose &s, const Imu &msg, const double accbFiltered[3], double now, bool useOdomEstimate, bool onlyTrustVel) {
    if (useOdomEstimate) {
        const double dt = now - s.odom.stamp;
        for (int i = 0; i < 3; ++i) {
            s.odom.pos[i] += s.odom.vel[i] * dt + 0.5 * s.odom.acc[i] * dt * dt;
            s.odom.vel[i] += s.odom.acc[i] * dt;
        }
        s.odom.stamp = now;
    }
    if (onlyTrustVel) {
        s.quat[0] = msg.orientation.w; s.quat[1] = msg.orientation.x; s.quat[2] = msg.orientation.y; s.quat[3] = msg.orientation.z;
        s.odom.yaw = YawFromQuat(s.quat[0], s.quat[1], s.quat[2], s.quat[3]);
    }
    double R[9];
    QuatToRot(s.quat[0], s.quat[1], s.quat[2], s.quat[3], R);
    for (int i = 0; i < 3; ++i)
        s.odom.acc[i] = R[3 * i] * accbFiltered[0] + R[3 * i + 1] * accbFiltered[1] + R[3 * i + 2] * accbFiltered[2] - (i == 2 ? 9.81 : 0.0);
}

// DepthCallback (:153-164): Twb (row-major 4x4) from the attitude and the (extrapolated) position, then AddVertex.
inline void DepthPose(const Pose &s, double now, bool useOdomEstimate, double Twb[16]) {
    const double dt = now - s.odom.stamp;
    double R[9];
    QuatToRot(s.quat[0], s.quat[1], s.quat[2], s.quat[3], R);
    for (int i = 0; i < 16; ++i) Twb[i] = 0.0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Twb[4 * r + c] = R[3 * r + c];
        Twb[4 * r + 3] = useOdomEstimate ? s.odom.pos[r] + s.odom.vel[r] * dt + 0.5 * s.odom.acc[r] * dt * dt : s.odom.pos[r];
    }
    Twb[15] = 1.0;
}
template <class Image>
void OnDepth(FrameKDMap &map, const Pose &s, const Image &img, double now, bool useOdomEstimate) {
    double Twb[16];
    DepthPose(s, now, useOdomEstimate, Twb);
    const DepthView v = ImageToDepth(img);
    map.AddVertex(Twb, v.data, v.type, v.rows, v.cols);
}

// ---------------------------------------------------------------------------------------------------------------------------
// out: quadrotor_msgs/Command
// ---------------------------------------------------------------------------------------------------------------------------
// Command: anything with .mode, .acceleration.{x,y,z}, .yaw and the constant ACCELERATION_MODE (quadrotor_msgs/Command).
// The header stamp is the caller's (ros::Time::now(), :371,389).
template <class Command>
void FillCmd(Command &cmd, const std::vector<double> &u) {  // PubCmd, :369-378
    cmd.mode = Command::ACCELERATION_MODE;
    cmd.acceleration.x = u[0];
    cmd.acceleration.y = u[1];
    cmd.acceleration.z = u[2];
    cmd.yaw = 0;
}
template <class Command>
void FillSlowDownCmd(Command &cmd, const OdomState &o, const Params &p) {  // PubSlowDownCmd, :379-397
    double a[3];
    for (int i = 0; i < 3; ++i) a[i] = -o.vel[i] * p.slowDownKp - o.acc[i] * p.slowDownKd + (i == 2 ? 9.8 : 0.0);
    cmd.mode = Command::ACCELERATION_MODE;
    cmd.acceleration.x = std::max(-p.aMaxXy, std::min(p.aMaxXy, a[0]));
    cmd.acceleration.y = std::max(-p.aMaxXy, std::min(p.aMaxXy, a[1]));
    cmd.acceleration.z = std::max(-p.aMaxZ, std::min(p.aMaxZ, a[2]));   // the reference clamps z to +-aMaxZ (not aMinZ), :386-387
    cmd.yaw = 0;
}
// The tail of the TASK branch (:345-350): publish u when the step was safe, the PD slow-down otherwise.  `alsoOnSolverFailure`
// extends the fallback to a step whose WORST solver status is non-zero (amk_step_batch flags[2] > 0: iteration cap or
// regularisation overflow) -- the reference cannot see that (it ignores IPOPT's status, HighLvlMpc.cpp:116-122), so it is
// off by default.
template <class Command>
bool FillStepCmd(Command &cmd, const int flags[4], const std::vector<double> &u, const OdomState &o, const Params &p,
                 bool alsoOnSolverFailure = false) {
    const bool ok = flags[0] != 0 && !(alsoOnSolverFailure && flags[2] > 0);
    if (ok) FillCmd(cmd, u);
    else FillSlowDownCmd(cmd, o, p);
    return ok;
}

}  // namespace ros_glue
}  // namespace avoid_mpc_amd
