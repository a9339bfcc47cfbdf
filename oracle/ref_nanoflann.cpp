// TEST INFRASTRUCTURE ONLY -- builds into oracle/_ref/ (git-ignored), never shipped, never linked
// by the product library.
//
// Thin C-ABI driver around the REFERENCE's own vendored nanoflann header, compiled IN PLACE from
//   /root/reference/roswrapper/ros/src/avoid_mpc/include/nanoflann_two.hpp
// (include path passed by oracle/Makefile; the header is never copied into this repo).
//
// The reference's adaptor class lives in AM/include/kd_tree_two.h, which cannot be compiled here
// (it includes PCL headers, kd_tree_two.h:5-7; PCL is not in this image).  nanoflann's public API
// is "bring your own dataset adaptor", so this driver supplies one with the SAME semantics as
// PointCloudTwo<double> (kd_tree_two.h:11-51: float32 storage, accessors widen to double) and
// restates the ~40 lines of KDTreeTwo<double> behaviour that surround the nanoflann calls:
//   * Initialize():  copy points whose x is not NaN (kd_tree_two.h:96-101), buildIndex (:105)
//   * ctor:          leaf size 10 (kd_tree_two.h:65-68)
//   * SearchForNearest(): result-count rule incl. the size==n quirk (kd_tree_two.h:119-124),
//                         size_t result buffers, int narrowing of indices (:128)
#include "nanoflann_two.hpp"

#include <cstddef>
#include <cstdint>
#include <vector>

namespace {

struct P3 {
    float x, y, z;
};

struct RefCloud {
    std::vector<P3> pts;
    inline size_t kdtree_get_point_count() const { return pts.size(); }
    inline double kdtree_get_pt(const size_t idx, int dim) const {
        if (dim == 0) return pts[idx].x;
        else if (dim == 1) return pts[idx].y;
        else return pts[idx].z;
    }
    template <class BBOX> bool kdtree_get_bbox(BBOX &) const { return false; }
};

using RefTree = nanoflann::KDTreeSingleIndexAdaptor<
    nanoflann::L2_Simple_Adaptor<double, RefCloud>, RefCloud, 3>;

struct RefKd {
    RefCloud cloud;
    RefTree index;
    RefKd() : cloud(), index(3, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(10)) {}
};

}  // namespace

extern "C" {

void *ref_kd_create(const float *xyz, int n, int stride) {
    RefKd *kd = new RefKd();
    for (int i = 0; i < n; ++i) {
        const float x = xyz[(size_t)i * stride];
        if (!(x != x)) kd->cloud.pts.push_back({x, xyz[(size_t)i * stride + 1], xyz[(size_t)i * stride + 2]});
    }
    kd->index.buildIndex();
    return kd;
}

int ref_kd_size(void *h) { return (int)static_cast<RefKd *>(h)->cloud.pts.size(); }

// Returns num_results as SearchForNearest would expose it (kd_tree_two.h:119-124).
int ref_kd_search(void *h, double x, double y, double z, int n, int *indices, double *sqdist,
                  float *pts_xyz) {
    RefKd *kd = static_cast<RefKd *>(h);
    if (kd->cloud.pts.size() == 0) return 0;
    double q[3] = {x, y, z};
    std::vector<size_t> ret(n > 0 ? n : 1);
    std::vector<double> d(n > 0 ? n : 1);
    nanoflann::KNNResultSet<double> rs(n);
    rs.init(ret.data(), d.data());
    kd->index.findNeighbors(rs, q);
    int num = 0;
    if ((long long)kd->cloud.pts.size() < (long long)n) num = (int)kd->cloud.pts.size();
    else if ((long long)kd->cloud.pts.size() > (long long)n) num = n;
    for (int i = 0; i < num; ++i) {
        indices[i] = (int)ret[i];
        sqdist[i] = d[i];
        if (pts_xyz) {
            pts_xyz[3 * i + 0] = kd->cloud.pts[ret[i]].x;
            pts_xyz[3 * i + 1] = kd->cloud.pts[ret[i]].y;
            pts_xyz[3 * i + 2] = kd->cloud.pts[ret[i]].z;
        }
    }
    return num;
}

// Raw nanoflann answer (no adaptor count rule): fills min(n, size) entries, returns that count.
int ref_kd_search_raw(void *h, double x, double y, double z, int n, int *indices, double *sqdist) {
    RefKd *kd = static_cast<RefKd *>(h);
    if (kd->cloud.pts.size() == 0 || n <= 0) return 0;
    double q[3] = {x, y, z};
    std::vector<size_t> ret(n);
    std::vector<double> d(n);
    nanoflann::KNNResultSet<double> rs(n);
    rs.init(ret.data(), d.data());
    kd->index.findNeighbors(rs, q);
    int num = (int)rs.size();
    for (int i = 0; i < num; ++i) {
        indices[i] = (int)ret[i];
        sqdist[i] = d[i];
    }
    return num;
}

// Build-only timing hook: rebuilds the index `reps` times (for cpu_baseline "reference" kind).
void ref_kd_rebuild(void *h, int reps) {
    RefKd *kd = static_cast<RefKd *>(h);
    for (int r = 0; r < reps; ++r) kd->index.buildIndex();
}

void ref_kd_destroy(void *h) { delete static_cast<RefKd *>(h); }

}  // extern "C"
