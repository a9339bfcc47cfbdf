// ObstacleAvoidanceMPC with the reference's interface (AM/include/HighLvlMpc.h:4-33,
// AM/src/HighLvlMpc.cpp:5-137) on top of the C ABI.  Header-only; links against libavoid_mpc_amd.so.
//
// The reference constructor takes the path of the CasADi-generated plugin, which bakes N = int(T/dt)
// and K = nearest_point_num (AM/tools/mpc_obstacle_casadi.py:36-37,76-85).  Here `soPath` is accepted
// for source compatibility and ignored; K is either given explicitly or inferred from the first
// Solve() call: len(vecRefStates) = 20 + 10 N + 3 K N  (AvoidanceStateMachine.cpp:236-257).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../avoid_mpc_amd.h"

namespace avoid_mpc_amd {

class ObstacleAvoidanceMPC {
public:
    ObstacleAvoidanceMPC() {}
    ObstacleAvoidanceMPC(double T, double dt, std::string soPath, int nearest_point_num = -1)
        : mT(T), mDt(dt), mN((int)(T / dt)), mK(nearest_point_num), mSoPath(std::move(soPath)) {
        // defaults of HighLvlMpc.cpp:13-16,53-56
        mWeights = {100, 100, 100, 300, 1, 1, 1, 0., 0., 0., 0.0, 10, 10, 30, 0, 1, 1, 0., 0., 0., 1., 1., 1., 1., 1.};
        mTau = {0.01, 0.01, 0.01, 0};
        mGains = {1, 1, 1, 1};
    }
    ~ObstacleAvoidanceMPC() {
        if (mpc_) amk_mpc_destroy(mpc_);
    }
    ObstacleAvoidanceMPC(ObstacleAvoidanceMPC &&o) noexcept { *this = std::move(o); }
    ObstacleAvoidanceMPC &operator=(ObstacleAvoidanceMPC &&o) noexcept {
        if (this != &o) {
            if (mpc_) amk_mpc_destroy(mpc_);
            mT = o.mT; mDt = o.mDt; mN = o.mN; mK = o.mK; mSoPath = o.mSoPath;
            mDroneRadius = o.mDroneRadius; for (int i = 0; i < 3; ++i) mDrag[i] = o.mDrag[i]; mTau = o.mTau; mGains = o.mGains; mWeights = o.mWeights;
            mLimits = o.mLimits; mHaveLimits = o.mHaveLimits; mpc_ = o.mpc_;
            o.mpc_ = nullptr;
        }
        return *this;
    }

    void SetupWeights(const std::vector<double> &weights) { mWeights = weights; if (mpc_) Push(); }
    void SetupTau(const std::vector<double> &tau) { mTau = tau; if (mpc_) Push(); }
    void SetupGains(const std::vector<double> &gains) { mGains = gains; if (mpc_) Push(); }
    void SetDroneRadius(const double droneRadius) { mDroneRadius = droneRadius; if (mpc_) Push(); }
    // Not in the reference's class (the generator bakes it into the plugin: use_drag_coefficient, mpc_obstacle_casadi.py:95-105):
    // v' = a - k .* v, the switch's expression read as matrix products (amk_mpc_set_drag_coefficient); {0, 0, 0} = off, the yaml's default
    void SetDragCoefficient(double kx, double ky, double kz) { mDrag[0] = kx; mDrag[1] = ky; mDrag[2] = kz; if (mpc_) Push(); }
    void SetDroneAccelLimits(const double aMinZ, const double aMaxZ, const double aMaxXy, const double aMaxYawDot) {
        mLimits = {aMinZ, aMaxZ, aMaxXy, aMaxYawDot};
        mHaveLimits = true;
        if (mpc_) Push();
    }

    void Solve(const std::vector<double> &vecRefStates, std::vector<double> &u,
               std::vector<std::vector<double>> &x0Array, bool faster = false) {  // HighLvlMpc.cpp:93-137
        if (!mpc_) Create(vecRefStates.size());
        if ((int)vecRefStates.size() != amk_mpc_ref_len(mpc_))
            throw std::runtime_error("ObstacleAvoidanceMPC::Solve: vecRefStates has the wrong length");
        u.assign(4, 0.0);
        std::vector<double> x0((size_t)14 * mN);
        int st = amk_mpc_solve_host(mpc_, vecRefStates.data(), u.data(), x0.data(), mInfo, faster ? 1 : 0);
        if (st != AMK_OK) throw std::runtime_error(std::string("amk_mpc_solve_host: ") + amk_status_string(st));
        x0Array.clear();  // rows [X_k, U_k], k < N  (.cpp:130-136)
        for (int k = 0; k < mN; ++k) x0Array.emplace_back(x0.begin() + 14 * k, x0.begin() + 14 * (k + 1));
    }
    // beyond the reference: the status the reference never looks at (.cpp:116-122)
    const int *LastSolveInfo() const { return mInfo; }
    // beyond the reference: arithmetic of the solve, 64 (default) or 32 (amk_mpc_set_precision)
    void SetPrecision(int bits) {
        mPrecision = bits;
        if (mpc_ && amk_mpc_set_precision(mpc_, bits) != AMK_OK)
            throw std::runtime_error("ObstacleAvoidanceMPC::SetPrecision: 32 or 64");
    }
    amk_mpc *handle() { return mpc_; }

private:
    void Create(size_t refLen) {
        if (mK < 0) {
            const long rem = (long)refLen - 20 - 10L * mN;
            if (mN <= 0 || rem < 0 || rem % (3L * mN) != 0)
                throw std::runtime_error("ObstacleAvoidanceMPC: cannot infer nearest_point_num from vecRefStates");
            mK = (int)(rem / (3L * mN));
        }
        int st = amk_mpc_create(mT, mDt, mK, 1, &mpc_);
        if (st != AMK_OK) throw std::runtime_error(std::string("amk_mpc_create: ") + amk_status_string(st));
        Push();
    }
    void Push() {
        amk_mpc_setup_weights(mpc_, mWeights.data());
        amk_mpc_setup_tau(mpc_, mTau.data());
        amk_mpc_setup_gains(mpc_, mGains.data());
        amk_mpc_set_drone_radius(mpc_, mDroneRadius);
        amk_mpc_set_drag_coefficient(mpc_, mDrag[0], mDrag[1], mDrag[2]);
        if (mHaveLimits) amk_mpc_set_drone_accel_limits(mpc_, mLimits[0], mLimits[1], mLimits[2], mLimits[3]);
        amk_mpc_set_precision(mpc_, mPrecision);
    }
    double mT = 0, mDt = 0;
    int mN = 0, mK = -1, mPrecision = 64;
    std::string mSoPath;
    double mDroneRadius = 0;
    double mDrag[3] = {0.0, 0.0, 0.0};
    std::vector<double> mTau, mGains, mWeights, mLimits;
    bool mHaveLimits = false;
    amk_mpc *mpc_ = nullptr;
    int mInfo[4] = {0, 0, 0, 0};
};

}  // namespace avoid_mpc_amd
