// The TASK branch of AvoidanceStateMachine::Step (AM/src/AvoidanceStateMachine.cpp:322-355) as a
// host object around amk_step_batch_host: one call per control period does the <= mpc_max_iter
// re-plan passes (dual KD queries, pack, solve, refill) on the GPU.  What a ROS node keeps is the
// odometry/IMU callbacks, GetInitPath and the publishing of u.
#pragma once
#include <cmath>
#include <stdexcept>
#include <vector>

#include "frame_kd_map.hpp"
#include "high_lvl_mpc.hpp"

namespace avoid_mpc_amd {

struct OdomState {  // what the callbacks store (AvoidanceStateMachine.cpp:118-152)
    double pos[3] = {0, 0, 0}, vel[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
    double yaw = 0;
    double stamp = 0;  // mTimePos: time stamp of the odometry message (seconds)
};

class AvoidanceTaskStep {
public:
    // mirrors SetupMPC (AvoidanceStateMachine.cpp:55-85)
    AvoidanceTaskStep(double T, double dt, int nearestPointNum, int mpcMaxIter, double speed, double safetyDistance,
                      double decay, double height)
        : mMpcT(T), mMpcDt(dt), mMpcN((int)(T / dt)), mK(nearestPointNum), mMaxIter(mpcMaxIter), mSpeed(speed),
          mSafety(safetyDistance), mDecay(decay), mHeight(height) {
        int st = amk_mpc_create(T, dt, nearestPointNum, 1, &mpc_);
        if (st != AMK_OK) throw std::runtime_error(std::string("amk_mpc_create: ") + amk_status_string(st));
        mRefPath.assign((size_t)mMpcN * 10, 0.0);
    }
    ~AvoidanceTaskStep() {
        if (mpc_) amk_mpc_destroy(mpc_);
    }
    amk_mpc *mpc() { return mpc_; }
    std::vector<double> &RefPath() { return mRefPath; }  // [N][10], mRefPath
    bool useOdomEstimate = true;                          // mParamIsUseOdomEstimate

    // GetCurStateQuad (:183-203)
    void CurStateQuad(const OdomState &o, double dt, double *sq) const {
        for (int i = 0; i < 3; ++i) {
            sq[i] = useOdomEstimate ? o.pos[i] + o.vel[i] * dt + 0.5 * o.acc[i] * dt * dt : o.pos[i];
            sq[4 + i] = useOdomEstimate ? o.vel[i] + o.acc[i] * dt : o.vel[i];
            sq[7 + i] = o.acc[i];
        }
        sq[3] = o.yaw;
    }
    // GetInitPath (:24-54).  globalGoal == nullptr: task "forward" (:29-33); else task "global_goal" (:34-45): the last point
    // of the path walks towards mStateGlobalGoal by at most mSpeed * mMpcDt (Eigen's normalized(): divide by the norm when it is
    // positive), and its z goes into every shifted point.
    void GetInitPath(const OdomState &o, double farestPoint, const double *globalGoal = nullptr) {
        double goalx = std::fmin(mSpeed * mMpcT + o.pos[0], farestPoint), goaly = 0.0, goalz = mHeight;
        if (globalGoal) {
            const double *last = &mRefPath[10 * (mMpcN - 1)];
            const double d0 = globalGoal[0] - last[0], d1 = globalGoal[1] - last[1], d2 = globalGoal[2] - last[2];
            const double z = (d0 * d0 + d1 * d1) + d2 * d2, nrm = std::sqrt(z);
            const double step = std::fmin(nrm, mSpeed * mMpcDt);
            const double e0 = z > 0.0 ? d0 / nrm : d0, e1 = z > 0.0 ? d1 / nrm : d1, e2 = z > 0.0 ? d2 / nrm : d2;
            goalx = last[0] + e0 * step; goaly = last[1] + e1 * step; goalz = last[2] + e2 * step;
        }
        for (int i = 0; i < mMpcN - 1; ++i) {
            for (int j = 0; j < 10; ++j) mRefPath[10 * i + j] = mRefPath[10 * (i + 1) + j];
            mRefPath[10 * i + 2] = goalz;
        }
        double *l = &mRefPath[10 * (mMpcN - 1)];
        for (int j = 0; j < 10; ++j) l[j] = 0.0;
        l[0] = goalx; l[1] = goaly; l[2] = goalz; l[4] = mSpeed;
    }

    // One TASK step against the current frame of `map`.  now: ros::Time::now() at the start of the step -- the state is
    // extrapolated over the AGE of the odometry plus the compute latency, dt = now + decay - mTimePos (:183-184,329-330);
    // pass now = o.stamp for fresh odometry.  iterTime: assumed duration of one re-plan pass (the reference measures it
    // with ros::Time::now(), :329,343; the device loop has no host round trip, so it is a model -- default: decay).
    // Single-frame map only (mVecQueryVector = [cur]): with keyframes in the map use the reference-shaped
    // FrameKDMap::QueryNearest path or amk_step_batch_frames.  Returns isSafety; u = last solve's control.
    bool Step(FrameKDMap &map, const OdomState &o, std::vector<double> &u, std::vector<std::vector<double>> &x0Array,
              double iterTime = -1.0, double now = -1.0) {
        const auto &fr = map.CurFrame();
        if (!fr.pointCloud || !fr.edgeCloud) throw std::runtime_error("AvoidanceTaskStep::Step: no frame in the map");
        if (iterTime < 0) iterTime = mDecay;
        std::vector<double> sq((size_t)mMaxIter * 10);
        const double age = now >= 0 ? now - o.stamp : 0.0;
        for (int i = 0; i < mMaxIter; ++i) CurStateQuad(o, age + (i == 0 ? mDecay : iterTime) + i * iterTime, &sq[10 * i]);   // pass i >= 1: (i + 1) * iterTime (:329-330,343)
        amk_step_params p;
        p.speed = mSpeed; p.safety_distance = mSafety; p.mpc_max_iter = mMaxIter; p.reserved = 0;
        u.assign(4, 0.0);
        std::vector<double> x0((size_t)14 * mMpcN);
        int st = amk_step_batch_host(fr.pointCloud->handle(), fr.edgeCloud->handle(), mpc_, &p, sq.data(), &o.pos[0],
                                     mRefPath.data(), u.data(), x0.data(), mFlags);
        if (st != AMK_OK) throw std::runtime_error(std::string("amk_step_batch_host: ") + amk_status_string(st));
        x0Array.clear();
        for (int k = 0; k < mMpcN; ++k) x0Array.emplace_back(x0.begin() + 14 * k, x0.begin() + 14 * (k + 1));
        return mFlags[0] != 0;
    }
    const int *Flags() const { return mFlags; }  // {isSafety, solves, last status, ipm iterations}

private:
    double mMpcT, mMpcDt;
    int mMpcN, mK, mMaxIter;
    double mSpeed, mSafety, mDecay, mHeight;
    std::vector<double> mRefPath;
    amk_mpc *mpc_ = nullptr;
    int mFlags[4] = {1, 0, -1, 0};
};

}  // namespace avoid_mpc_amd
