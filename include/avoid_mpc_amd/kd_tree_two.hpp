// KDTreeTwo<num_t> with the reference's interface (AM/include/kd_tree_two.h:53-144) on top of the C ABI
// (include/avoid_mpc_amd.h).  Header-only; links against libavoid_mpc_amd.so.  No PCL needed: any
// cloud pointer whose pointee has a `points` container of {x,y,z} floats works, which
// pcl::PointCloud<pcl::PointXYZ>::Ptr satisfies.
#pragma once
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../avoid_mpc_amd.h"

namespace avoid_mpc_amd {

#if __has_include(<pcl/point_types.h>)
}  // namespace avoid_mpc_amd
#include <pcl/point_types.h>
namespace avoid_mpc_amd {
using PointXYZ = pcl::PointXYZ;
#else
struct alignas(16) PointXYZ {  // layout of pcl::PointXYZ: 3 floats + 4 bytes padding
    float x, y, z, pad_ = 1.f;
    PointXYZ() : x(0), y(0), z(0) {}
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
#endif

template <typename T>
struct PointCloudTwo {  // kd_tree_two.h:11-51 (the accessors nanoflann needed are gone with nanoflann)
    std::vector<PointXYZ> pts;
    inline size_t kdtree_get_point_count() const { return pts.size(); }
};

inline void amk_throw(int status, const char *what) {
    if (status != AMK_OK) throw std::runtime_error(std::string(what) + ": " + amk_status_string(status));
}

template <typename num_t>
class KDTreeTwo {
public:
    std::vector<PointXYZ> closest_pts;
    std::vector<num_t> squared_distances;
    std::vector<int> indices;
    std::vector<int> colors;

    KDTreeTwo() {
        for (int i = 0; i < 3; i++) colors.push_back(rand() % 256);  // kd_tree_two.h:69-72
    }
    ~KDTreeTwo() {
        if (kd_) amk_kd_destroy(kd_);
    }
    KDTreeTwo(const KDTreeTwo &) = delete;
    KDTreeTwo &operator=(const KDTreeTwo &) = delete;

    template <class CloudPtr>
    void InitializeNew(CloudPtr const &xyz_cloud_new) { Initialize(xyz_cloud_new, true); }
    template <class CloudPtr>
    void AddToKDTree(CloudPtr const &xyz_cloud_new) { Initialize(xyz_cloud_new, false); }
    void Clear() { cloud.pts.clear(); }
    // Not in the reference: true -> later (re)builds also construct nanoflann's own tree on the device and searches
    // follow its traversal, so that `indices` equal the reference's on clouds with equal squared distances too
    // (amk_kd_set_tie_order, AMK_TIES_NANOFLANN); default: equal distances ordered by index.
    void SetNanoflannTieOrder(bool on) {
        tie_order_ = on ? AMK_TIES_NANOFLANN : AMK_TIES_LOWEST_INDEX;
        if (kd_) amk_throw(amk_kd_set_tie_order(kd_, tie_order_), "amk_kd_set_tie_order");
    }

    // Not in the reference: whether SetNanoflannTieOrder(true) holds for the cloud the tree currently indexes --
    // AMK_EXACT_IN_USE (0): searches follow nanoflann's own tree; AMK_EXACT_GAVE_UP (1) / AMK_EXACT_TOO_DEEP (2): pathological
    // data, the bucketed index answers (same distances, equal distances in index order); AMK_EXACT_OFF (-1): mode off / no build
    int NanoflannTieOrderStatus() const {
        int st = AMK_EXACT_OFF;
        if (kd_) amk_throw(amk_kd_exact_status_host(kd_, &st), "amk_kd_exact_status_host");
        return st;
    }

    template <class CloudPtr>
    void Initialize(CloudPtr const &xyz_cloud_new, bool clear) {  // kd_tree_two.h:88-106
        if (clear) cloud.pts.clear();
        const size_t num_points = xyz_cloud_new->points.size();
        for (size_t i = 0; i < num_points; i++) {
            const auto &p = xyz_cloud_new->points[i];
            if (!(p.x != p.x)) cloud.pts.push_back(PointXYZ(p.x, p.y, p.z));
        }
        Rebuild();
    }

    void SearchForNearest(num_t x, num_t y, num_t z, int n) {  // kd_tree_two.h:108-133
        closest_pts.clear();
        squared_distances.clear();
        indices.clear();
        if (cloud.pts.size() == 0 || n <= 0) return;
        if (n > AMK_MAX_K) throw std::runtime_error("KDTreeTwo::SearchForNearest: n > AMK_MAX_K");
        const double q[3] = {(double)x, (double)y, (double)z};
        std::vector<int> idx(n);
        std::vector<double> d2(n);
        std::vector<float> pts(3 * (size_t)n);
        int count = 0;
        amk_throw(amk_kd_search_host(kd_, q, 1, n, idx.data(), d2.data(), pts.data(), &count), "amk_kd_search_host");
        for (int i = 0; i < count; i++) {
            closest_pts.push_back(cloud.pts[idx[i]]);
            squared_distances.push_back((num_t)d2[i]);
            indices.push_back(idx[i]);
        }
    }
    // Several SearchForNearest calls in ONE device round trip (the reference issues them one at a time: N per re-plan
    // pass in ProcessWaypoints, AvoidanceStateMachine.cpp:210-215; each single call costs a launch + two small copies,
    // tens of microseconds -- see INTEGRATION.md).  queries: q[3*i + {0,1,2}], nq <= AMK_MAX_QUERIES.  Results of query i:
    // out_indices[i], out_sqdist[i] (same contents as `indices` / `squared_distances` after SearchForNearest(q_i, n)).
    void SearchForNearestBatch(const double *queries, int nq, int n, std::vector<std::vector<int>> &out_indices,
                               std::vector<std::vector<num_t>> &out_sqdist) {
        out_indices.assign(nq, {});
        out_sqdist.assign(nq, {});
        if (cloud.pts.size() == 0 || n <= 0 || nq <= 0) return;
        if (n > AMK_MAX_K || nq > AMK_MAX_QUERIES) throw std::runtime_error("KDTreeTwo::SearchForNearestBatch: too many");
        std::vector<int> idx((size_t)nq * n), cnt(nq);
        std::vector<double> d2((size_t)nq * n);
        amk_throw(amk_kd_search_host(kd_, queries, nq, n, idx.data(), d2.data(), nullptr, cnt.data()), "amk_kd_search_host");
        for (int i = 0; i < nq; ++i)
            for (int j = 0; j < cnt[i]; ++j) {
                out_indices[i].push_back(idx[(size_t)i * n + j]);
                out_sqdist[i].push_back((num_t)d2[(size_t)i * n + j]);
            }
    }
    PointCloudTwo<num_t> const &GetPointCloud() { return cloud; }
    std::vector<int> const &GetColors() { return colors; }
    amk_kd *handle() { return kd_; }  // for batched / fused use through the C ABI
    // after a device-side rebuild (amk_kd_keyframe_sweep): refresh the host copy behind GetPointCloud()
    void SyncFromDevice() {
        if (!kd_) return;
        std::vector<float> xyz((size_t)capacity_ * 3);
        int n = 0;
        amk_throw(amk_kd_points_host(kd_, xyz.data(), &n), "amk_kd_points_host");
        cloud.pts.resize(n);
        for (int i = 0; i < n; ++i) cloud.pts[i] = PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    }

private:
    void Rebuild() {
        const int n = (int)cloud.pts.size();
        if (!kd_ || n > capacity_) {
            if (kd_) amk_kd_destroy(kd_);
            kd_ = nullptr;
            capacity_ = n > 1024 ? n + n / 2 : 1024;
            amk_throw(amk_kd_create(1, capacity_, &kd_), "amk_kd_create");
            amk_throw(amk_kd_set_tie_order(kd_, tie_order_), "amk_kd_set_tie_order");
        }
        static_assert(sizeof(PointXYZ) == 16, "PointXYZ must be 16 bytes (pcl layout)");
        static const PointXYZ none(0, 0, 0);  // (a valid address for an empty cloud; only n points are read)
        amk_throw(amk_kd_build_host(kd_, reinterpret_cast<const float *>(n ? cloud.pts.data() : &none), 4, 4LL * capacity_, &n),
                  "amk_kd_build_host");
    }
    PointCloudTwo<num_t> cloud;
    amk_kd *kd_ = nullptr;
    int capacity_ = 0;
    int tie_order_ = AMK_TIES_LOWEST_INDEX;
};

}  // namespace avoid_mpc_amd
