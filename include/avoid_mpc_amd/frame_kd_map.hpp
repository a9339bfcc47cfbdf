// FrameKDMap's query front end with the reference's interface (AM/include/FrameKDMap.h:60-67,
// AM/src/FrameKDMap.cpp:34-74,254-427) on top of KDTreeTwo (kd_tree_two.hpp -> C ABI).
//
// In scope: per-frame dual trees, the query list [cur, keyframes[0..size-2]]
// (FrameKDMap.cpp:64-74), the fast path / multi-frame merge of QueryNearest (:322-376) and
// GetNearestDistance (:400-427), the keyframe sweep (§8 f1, KeyframeUpdate), ProcessDepth (§8 f2: depth image ->
// world-frame cloud, FrameKDMap.cpp:90-130) and BuildEdgeCloud (§8 f3: quantise, erode, Canny, back-project,
// :176-214), i.e. the reference's AddVertex(Twb, depth image) end to end.  The reference fans frames out over
// std::threads; every per-frame search here is one device call, so the loop is sequential on the host.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <functional>
#include <list>
#include <memory>
#include <mutex>
#include <vector>

#include "kd_tree_two.hpp"

namespace avoid_mpc_amd {

#if __has_include(<Eigen/Core>)
}  // namespace avoid_mpc_amd
#include <Eigen/Core>
namespace avoid_mpc_amd {
using Vector3d = Eigen::Vector3d;
inline double vx(const Vector3d &v) { return v.x(); }
inline double vy(const Vector3d &v) { return v.y(); }
inline double vz(const Vector3d &v) { return v.z(); }
#else
struct Vector3d {
    double x_, y_, z_;
    Vector3d() : x_(0), y_(0), z_(0) {}
    Vector3d(double x, double y, double z) : x_(x), y_(y), z_(z) {}
    double x() const { return x_; }
    double y() const { return y_; }
    double z() const { return z_; }
};
inline double vx(const Vector3d &v) { return v.x(); }
inline double vy(const Vector3d &v) { return v.y(); }
inline double vz(const Vector3d &v) { return v.z(); }
#endif

class FrameKDMap {
public:
    using PtCloudKdPtr = std::shared_ptr<KDTreeTwo<double>>;
    struct Frame {
        PtCloudKdPtr pointCloud;
        PtCloudKdPtr edgeCloud;
    };

    // PtIsInFrame (FrameKDMap.cpp:215-231): the point in the camera frame of `Twc` (row-major 4x4), 0 <= z <= depth_max,
    // projected with the intrinsics DIVIDED by the resize scale (:21-24) inside the down-scaled image [0,W) x [0,H)
    // (mParamWidth / mParamHeight are set by ProcessDepth, :106-107).  Twc is rigid: its inverse is [R' | -R' t] (the
    // reference calls Eigen's general 4x4 inverse on the same matrix).
    bool PtIsInFrame(const Vector3d &ptw, const double *Twc) const {
        const double dx = vx(ptw) - Twc[3], dy = vy(ptw) - Twc[7], dz = vz(ptw) - Twc[11];
        const double x = Twc[0] * dx + Twc[4] * dy + Twc[8] * dz;
        const double y = Twc[1] * dx + Twc[5] * dy + Twc[9] * dz;
        const double z = Twc[2] * dx + Twc[6] * dy + Twc[10] * dz;
        if (z > depthParams.depth_max || z < 0) return false;                                       // :222-224
        const double sc = depthParams.resize_scale;
        const double u = (depthParams.fx / sc) * x / z + depthParams.cx / sc;                       // :225
        const double v = (depthParams.fy / sc) * y / z + depthParams.cy / sc;                       // :226
        if (u < 0 || u >= mParamWidth || v < 0 || v >= mParamHeight) return false;                  // :227-229
        return true;
    }
    // Camera model of the fast-path test when the clouds are supplied directly (no ProcessDepth call has set them).
    void SetImageSize(int rows, int cols) {
        mParamWidth = (int)(cols / depthParams.resize_scale);                                       // :106-107
        mParamHeight = (int)(rows / depthParams.resize_scale);
    }
    // The fast-path predicate of QueryNearest (:339-340).  Default: the reference's PtIsInFrame(point, mCurFrame.Twc)
    // once the image size is known (after the first ProcessDepth / SetImageSize); before that the reference reads
    // uninitialised members -- here every point counts as inside.  Replaceable for hosts with their own camera model.
    std::function<bool(const Vector3d &)> ptIsInCurFrame = [this](const Vector3d &p) {
        return mParamWidth <= 0 ? true : PtIsInFrame(p, mTwc);
    };
    // mCurFrame.Twc (row-major).  The reference leaves it uninitialised until the first AddVertex; identity here.
    double mTwc[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

    // ProcessDepth, obstacle cloud only (FrameKDMap.cpp:90-130): raw depth buffer (AMK_DEPTH_U16 / AMK_DEPTH_F32,
    // rows x cols, tightly packed) + Twb (row-major 4x4) -> points in the reference's row-major pixel order.
    // `Cloud` is anything with a `points` vector of {x, y, z} floats (pcl::PointCloud<pcl::PointXYZ> qualifies).
    amk_depth_params depthParams{};  // perception parameters (ParameterManager.cpp:44-57); fx..cy full-resolution
    template <class Cloud>
    void ProcessDepth(const void *depth, int depthType, int rows, int cols, const double *Twb, Cloud &cloud) {
        int w = 0, h = 0;
        amk_throw(amk_depth_out_size(rows, cols, depthParams.resize_scale, &w, &h), "amk_depth_out_size");
        mParamWidth = w; mParamHeight = h;                                                         // :106-107
        std::vector<float> xyz((size_t)w * h * 3);
        int n = 0;
        amk_throw(amk_depth_to_cloud_host(depth, depthType, rows, cols, (long long)rows * cols, 1, &depthParams, Twb,
                                          xyz.data(), 3, (long long)w * h * 3, &n), "amk_depth_to_cloud_host");
        cloud.points.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            cloud.points[i].x = xyz[3 * i];
            cloud.points[i].y = xyz[3 * i + 1];
            cloud.points[i].z = xyz[3 * i + 2];
        }
    }

    // BuildEdgeCloud (FrameKDMap.cpp:176-214).  Twc is what the reference multiplies with Tbc at :209: mCurFrame.Twc,
    // the PREVIOUS frame's Twb * Tbc.
    template <class Cloud>
    void BuildEdgeCloud(const void *depth, int depthType, int rows, int cols, const double *Twc, Cloud &edgeCloud) {
        int w = 0, h = 0;
        amk_throw(amk_depth_out_size(rows, cols, depthParams.resize_scale, &w, &h), "amk_depth_out_size");
        std::vector<float> xyz((size_t)w * h * 3);
        int n = 0;
        amk_throw(amk_depth_to_edge_cloud_host(depth, depthType, rows, cols, (long long)rows * cols, 1, &depthParams, Twc,
                                               xyz.data(), 3, (long long)w * h * 3, &n), "amk_depth_to_edge_cloud_host");
        edgeCloud.points.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            edgeCloud.points[i].x = xyz[3 * i];
            edgeCloud.points[i].y = xyz[3 * i + 1];
            edgeCloud.points[i].z = xyz[3 * i + 2];
        }
    }

    // AddVertex(mat4Twb, depth) of the reference (FrameKDMap.cpp:34-51), the depth image as a raw buffer:
    // ProcessDepth + BuildEdgeCloud (with the stale pose, as the reference), two fresh trees, swap, Twc = Twb * Tbc.
    struct XYZ { float x, y, z; };
    struct XYZCloud { std::vector<XYZ> points; };
    void AddVertex(const double *Twb, const void *depth, int depthType, int rows, int cols) {
        auto cloud = std::make_shared<XYZCloud>(), edgeCloud = std::make_shared<XYZCloud>();
        ProcessDepth(depth, depthType, rows, cols, Twb, *cloud);
        if (cloud->points.empty()) return;                                         // :39-41
        BuildEdgeCloud(depth, depthType, rows, cols, mTwc, *edgeCloud);            // ProcessDepth's tail, :130
        AddVertex(cloud, edgeCloud);
        for (int i = 0; i < 4; ++i)                                                // mCurFrame.Twc = mat4Twb * mParamTbc, :50
            for (int j = 0; j < 4; ++j) {
                double acc = 0;
                for (int k = 0; k < 4; ++k) acc += Twb[4 * i + k] * depthParams.Tbc[4 * k + j];
                mTwc[4 * i + j] = acc;
            }
    }
    const double *CurTwc() const { return mTwc; }

    // AddVertex after ProcessDepth (FrameKDMap.cpp:39-51): two fresh trees, then swap under the lock.
    template <class CloudPtr>
    void AddVertex(CloudPtr const &cloud, CloudPtr const &edgeCloud) {
        if (cloud->points.empty()) return;  // :39-41
        PtCloudKdPtr kdtree = std::make_shared<KDTreeTwo<double>>();
        kdtree->InitializeNew(cloud);
        PtCloudKdPtr edgeKdtree = std::make_shared<KDTreeTwo<double>>();
        edgeKdtree->InitializeNew(edgeCloud);
        std::lock_guard<std::mutex> lock(mMtxKdTree);
        mCurFrame.pointCloud = kdtree;
        mCurFrame.edgeCloud = edgeKdtree;
        UpdateQueryVector();
    }
    void InsertKeyFrame() {  // FrameKDMap.cpp:428-431
        std::lock_guard<std::mutex> lock(mMtxKdTree);
        mKeyFrameMap.push_back(mCurFrame);
    }
    const Frame &CurFrame() const { return mCurFrame; }

    void QueryNearest(const Vector3d &point, int nearestPointCount, std::vector<Vector3d> &nearestPoints,
                      std::vector<double> &distances, bool queryEdge = false) {  // :322-376
        struct PtDists { Vector3d pt; double dist; };
        std::vector<PtDists> pointsDist;
        std::lock_guard<std::mutex> lock(mMtxKdTree);
        PtCloudKdPtr cur = queryEdge ? mCurFrame.edgeCloud : mCurFrame.pointCloud;
        if (cur) {
            const int first = (int)cur->GetPointCloud().pts.size();
            if (first >= nearestPointCount && ptIsInCurFrame(point)) {  // fast path :339-345
                cur->SearchForNearest(vx(point), vy(point), vz(point), nearestPointCount);
                nearestPoints.clear();
                distances.clear();
                for (size_t i = 0; i < cur->squared_distances.size(); ++i) {
                    nearestPoints.push_back(Vector3d(cur->closest_pts[i].x, cur->closest_pts[i].y, cur->closest_pts[i].z));
                    distances.push_back(cur->squared_distances[i]);
                }
                return;
            }
        }
        for (auto &fr : mVecQueryVector) {  // :347-364 (one worker per frame in the reference)
            PtCloudKdPtr pc = queryEdge ? fr.edgeCloud : fr.pointCloud;
            if (!pc) continue;
            const int queryPointCount = std::min(nearestPointCount, (int)pc->GetPointCloud().pts.size());  // :298
            pc->SearchForNearest(vx(point), vy(point), vz(point), queryPointCount);
            for (size_t j = 0; j < pc->squared_distances.size(); ++j)
                pointsDist.push_back({Vector3d(pc->closest_pts[j].x, pc->closest_pts[j].y, pc->closest_pts[j].z),
                                      pc->squared_distances[j]});
        }
        nearestPoints.clear();
        distances.clear();
        if (pointsDist.empty()) return;
        std::sort(pointsDist.begin(), pointsDist.end(), [](const PtDists &a, const PtDists &b) { return a.dist < b.dist; });
        for (int i = 0; i < nearestPointCount && i < (int)pointsDist.size(); ++i) {  // :372-375
            nearestPoints.push_back(pointsDist[i].pt);
            distances.push_back(pointsDist[i].dist);
        }
    }

    // One pass of KeyframeThreadWorker's body (:443-486) -- the caller owns the 30 ms cadence and the
    // "new frame arrived" flag.  dronePos / bodyX: translation of Twb = Twc * Tbc^-1 and the world
    // direction of the body x axis (first column of Rwb), what DroneBehindPts (:233-252) derives.
    // The n x 1-NN sweep and the rebuild run on the GPU (amk_kd_keyframe_sweep).
    void KeyframeUpdate(const Vector3d &dronePos, const Vector3d &bodyX, double depthMin, double keyframeThDist,
                        int keyframeThCount, int maxFrameCount) {
        std::lock_guard<std::mutex> lock(mMtxKdTree);
        if (!mCurFrame.pointCloud) return;
        if (mKeyFrameMap.empty()) {  // :446-449
            mKeyFrameMap.push_back(mCurFrame);
            UpdateQueryVector();
            return;
        }
        while (!mKeyFrameMap.empty()) {  // :450-459
            Frame &oldest = mKeyFrameMap.front();
            if ((int)mKeyFrameMap.size() > maxFrameCount || !DroneBehindPts(dronePos, bodyX, depthMin, oldest)) {
                oldest.pointCloud->Clear();
                mKeyFrameMap.pop_front();
                UpdateQueryVector();
            } else {
                break;
            }
        }
        if (mKeyFrameMap.empty()) return;
        PtCloudKdPtr last = mKeyFrameMap.back().pointCloud;
        if (last == mCurFrame.pointCloud) return;  // the reference would sweep a tree against itself: no outliers
        int outliers = 0, rebuilt = 0;
        amk_throw(amk_kd_keyframe_sweep_host(last->handle(), mCurFrame.pointCloud->handle(), keyframeThDist,
                                             keyframeThCount, &outliers, &rebuilt), "amk_kd_keyframe_sweep_host");
        mLastSweepOutliers = outliers;
        if (!rebuilt) return;  // :477-479
        last->SyncFromDevice();  // :480-485 rebuilt the tree from the outliers
        mKeyFrameMap.push_back(mCurFrame);  // InsertKeyFrame :486
        UpdateQueryVector();
    }
    int LastSweepOutliers() const { return mLastSweepOutliers; }
    size_t KeyFrameCount() const { return mKeyFrameMap.size(); }

    double GetNearestDistance(const Vector3d &point) {  // :400-427
        double nearestDistance = DBL_MAX;
        std::lock_guard<std::mutex> lock(mMtxKdTree);
        if (mVecQueryVector.empty()) return nearestDistance;
        for (auto &fr : mVecQueryVector) {
            if (!fr.pointCloud || fr.pointCloud->GetPointCloud().pts.empty()) continue;  // :385-387
            fr.pointCloud->SearchForNearest(vx(point), vy(point), vz(point), 1);
            if (!fr.pointCloud->squared_distances.empty())
                nearestDistance = std::min(nearestDistance, fr.pointCloud->squared_distances[0]);
        }
        return std::sqrt(nearestDistance);
    }

private:
    bool DroneBehindPts(const Vector3d &twb, const Vector3d &bodyX, double depthMin, const Frame &frame) {  // :233-252
        const int ptsCount = std::min((int)frame.pointCloud->GetPointCloud().pts.size(), 10);
        frame.pointCloud->SearchForNearest(vx(twb), vy(twb), vz(twb), ptsCount);
        for (const auto &pt : frame.pointCloud->closest_pts) {
            const double ptbx = vx(bodyX) * (pt.x - vx(twb)) + vy(bodyX) * (pt.y - vy(twb)) + vz(bodyX) * (pt.z - vz(twb));
            if (ptbx <= depthMin) return false;
        }
        return true;
    }
    int mLastSweepOutliers = 0;
    int mParamWidth = 0, mParamHeight = 0;
    void UpdateQueryVector() {  // :64-74: current frame + all key frames but the newest
        mVecQueryVector.clear();
        mVecQueryVector.push_back(mCurFrame);
        if (!mKeyFrameMap.empty()) {
            auto it = mKeyFrameMap.begin();
            for (size_t i = 0; i + 1 < mKeyFrameMap.size(); ++i, ++it) mVecQueryVector.push_back(*it);
        }
    }
    std::list<Frame> mKeyFrameMap;
    std::vector<Frame> mVecQueryVector;
    Frame mCurFrame;
    std::mutex mMtxKdTree;
};

}  // namespace avoid_mpc_amd
