/* TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's KD-tree path in plain C.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * library (avoid_mpc_amd/csrc) never links or calls it.
 *
 * What is restated, with the reference lines each function follows
 * (AM = /root/reference/roswrapper/ros/src/avoid_mpc):
 *   kdo_create        KDTreeTwo<double>::Initialize      AM/include/kd_tree_two.h:88-106
 *                     nanoflann buildIndex               AM/include/nanoflann_two.hpp:1518-1541
 *   bbox_all          computeBoundingBox                 nanoflann_two.hpp:1694-1720
 *   divide            divideTree                         nanoflann_two.hpp:1055-1106
 *   middle_split      middleSplit_                       nanoflann_two.hpp:1197-1245
 *   plane_split       planeSplit                         nanoflann_two.hpp:1256-1294
 *   rs_add            KNNResultSet::addPoint             nanoflann_two.hpp:219-246
 *   search_level      searchLevel                        nanoflann_two.hpp:1729-1793
 *   kdo_search_raw    findNeighbors + initial distances  nanoflann_two.hpp:1563-1586,1296-1315
 *   kdo_search        KDTreeTwo::SearchForNearest        kd_tree_two.h:108-133
 *   dist2             PointCloudTwo / L2_Simple_Adaptor  kd_tree_two.h:24-27, nanoflann_two.hpp:590-598
 *
 * Pinned by: oracle/_ref (the reference header itself, compiled in place) on random, tie-heavy and
 * degenerate clouds (tests/test_kd_oracle.py) and by the golden vectors in tests/golden/.
 * Compile with -ffp-contract=off: distances are then the canonical IEEE left-to-right sums.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KDO_LEAF 10 /* kd_tree_two.h:68 */

typedef struct {
    int divfeat;            /* -1 => leaf */
    double divlow, divhigh; /* inner */
    int child1, child2;     /* inner */
    uint32_t left, right;   /* leaf: [left,right) into vind */
} kdo_node;

typedef struct {
    float *pts; /* n x 3, NaN-x points already dropped */
    uint32_t n;
    uint32_t *vind;
    kdo_node *nodes;
    int n_nodes, cap_nodes;
    int root;
    double bbox[3][2];
} kdo_tree;

typedef struct {
    size_t *indices;
    double *dists;
    size_t capacity, count;
} kdo_rs;

static inline double getpt(const kdo_tree *t, uint32_t idx, int dim) { return (double)t->pts[3 * (size_t)idx + dim]; }

static int new_node(kdo_tree *t) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1024;
        t->nodes = (kdo_node *)realloc(t->nodes, sizeof(kdo_node) * (size_t)t->cap_nodes);
    }
    return t->n_nodes++;
}

static void compute_minmax(const kdo_tree *t, uint32_t ind, uint32_t count, int dim, double *mn, double *mx) {
    *mn = getpt(t, t->vind[ind], dim);
    *mx = *mn;
    for (uint32_t i = 1; i < count; ++i) {
        double v = getpt(t, t->vind[ind + i], dim);
        if (v < *mn) *mn = v;
        if (v > *mx) *mx = v;
    }
}

static void plane_split(kdo_tree *t, uint32_t ind, uint32_t count, int cutfeat, double cutval, uint32_t *lim1,
                        uint32_t *lim2) {
    /* size_t arithmetic as in the reference (Offset = vector::size_type) */
    size_t left = 0, right = (size_t)count - 1;
    for (;;) {
        while (left <= right && getpt(t, t->vind[ind + left], cutfeat) < cutval) ++left;
        while (right && left <= right && getpt(t, t->vind[ind + right], cutfeat) >= cutval) --right;
        if (left > right || !right) break;
        uint32_t tmp = t->vind[ind + left];
        t->vind[ind + left] = t->vind[ind + right];
        t->vind[ind + right] = tmp;
        ++left;
        --right;
    }
    *lim1 = (uint32_t)left;
    right = (size_t)count - 1;
    for (;;) {
        while (left <= right && getpt(t, t->vind[ind + left], cutfeat) <= cutval) ++left;
        while (right && left <= right && getpt(t, t->vind[ind + right], cutfeat) > cutval) --right;
        if (left > right || !right) break;
        uint32_t tmp = t->vind[ind + left];
        t->vind[ind + left] = t->vind[ind + right];
        t->vind[ind + right] = tmp;
        ++left;
        --right;
    }
    *lim2 = (uint32_t)left;
}

static void middle_split(kdo_tree *t, uint32_t ind, uint32_t count, uint32_t *index, int *cutfeat, double *cutval,
                         double bbox[3][2]) {
    const double EPS = 0.00001;
    double max_span = bbox[0][1] - bbox[0][0];
    for (int i = 1; i < 3; ++i) {
        double span = bbox[i][1] - bbox[i][0];
        if (span > max_span) max_span = span;
    }
    double max_spread = -1;
    *cutfeat = 0;
    double min_elem = 0, max_elem = 0;
    for (int i = 0; i < 3; ++i) {
        double span = bbox[i][1] - bbox[i][0];
        if (span > (1 - EPS) * max_span) {
            double mn, mx;
            compute_minmax(t, ind, count, i, &mn, &mx);
            double spread = mx - mn;
            if (spread > max_spread) {
                *cutfeat = i;
                max_spread = spread;
                min_elem = mn;
                max_elem = mx;
            }
        }
    }
    double split_val = (bbox[*cutfeat][0] + bbox[*cutfeat][1]) / 2;
    if (split_val < min_elem) *cutval = min_elem;
    else if (split_val > max_elem) *cutval = max_elem;
    else *cutval = split_val;
    uint32_t lim1, lim2;
    plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
    if (lim1 > count / 2) *index = lim1;
    else if (lim2 < count / 2) *index = lim2;
    else *index = count / 2;
}

static int divide(kdo_tree *t, uint32_t left, uint32_t right, double bbox[3][2]) {
    int id = new_node(t);
    if ((right - left) <= KDO_LEAF) {
        kdo_node nd;
        memset(&nd, 0, sizeof nd);
        nd.divfeat = -1;
        nd.child1 = nd.child2 = -1;
        nd.left = left;
        nd.right = right;
        t->nodes[id] = nd;
        for (int i = 0; i < 3; ++i) bbox[i][0] = bbox[i][1] = getpt(t, t->vind[left], i);
        for (uint32_t k = left + 1; k < right; ++k)
            for (int i = 0; i < 3; ++i) {
                double v = getpt(t, t->vind[k], i);
                if (bbox[i][0] > v) bbox[i][0] = v;
                if (bbox[i][1] < v) bbox[i][1] = v;
            }
    } else {
        uint32_t idx;
        int cutfeat;
        double cutval;
        middle_split(t, left, right - left, &idx, &cutfeat, &cutval, bbox);
        double lb[3][2], rb[3][2];
        memcpy(lb, bbox, sizeof lb);
        lb[cutfeat][1] = cutval;
        int c1 = divide(t, left, left + idx, lb);
        memcpy(rb, bbox, sizeof rb);
        rb[cutfeat][0] = cutval;
        int c2 = divide(t, left + idx, right, rb);
        kdo_node nd;
        memset(&nd, 0, sizeof nd);
        nd.divfeat = cutfeat;
        nd.child1 = c1;
        nd.child2 = c2;
        nd.divlow = lb[cutfeat][1];
        nd.divhigh = rb[cutfeat][0];
        t->nodes[id] = nd;
        for (int i = 0; i < 3; ++i) {
            bbox[i][0] = lb[i][0] < rb[i][0] ? lb[i][0] : rb[i][0];
            bbox[i][1] = lb[i][1] > rb[i][1] ? lb[i][1] : rb[i][1];
        }
    }
    return id;
}

static void build_index(kdo_tree *t) {
    for (uint32_t i = 0; i < t->n; ++i) t->vind[i] = i;
    t->n_nodes = 0;
    t->root = -1;
    if (t->n == 0) return;
    for (int i = 0; i < 3; ++i) t->bbox[i][0] = t->bbox[i][1] = getpt(t, t->vind[0], i);
    for (uint32_t k = 1; k < t->n; ++k)
        for (int i = 0; i < 3; ++i) {
            double v = getpt(t, t->vind[k], i);
            if (v < t->bbox[i][0]) t->bbox[i][0] = v;
            if (v > t->bbox[i][1]) t->bbox[i][1] = v;
        }
    t->root = divide(t, 0, t->n, t->bbox);
}

void *kdo_create(const float *xyz, int n, int stride) {
    kdo_tree *t = (kdo_tree *)calloc(1, sizeof(kdo_tree));
    t->pts = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    uint32_t m = 0;
    for (int i = 0; i < n; ++i) {
        float x = xyz[(size_t)i * stride];
        if (!(x != x)) { /* kd_tree_two.h:99 -- only x is tested */
            t->pts[3 * (size_t)m + 0] = x;
            t->pts[3 * (size_t)m + 1] = xyz[(size_t)i * stride + 1];
            t->pts[3 * (size_t)m + 2] = xyz[(size_t)i * stride + 2];
            ++m;
        }
    }
    t->n = m;
    t->vind = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(m > 0 ? m : 1));
    build_index(t);
    return t;
}

void kdo_rebuild(void *h, int reps) {
    kdo_tree *t = (kdo_tree *)h;
    for (int r = 0; r < reps; ++r) build_index(t);
}

int kdo_size(void *h) { return (int)((kdo_tree *)h)->n; }
int kdo_num_nodes(void *h) { return ((kdo_tree *)h)->n_nodes; }

void kdo_destroy(void *h) {
    kdo_tree *t = (kdo_tree *)h;
    free(t->pts);
    free(t->vind);
    free(t->nodes);
    free(t);
}

static inline double dist2(const kdo_tree *t, const double *q, uint32_t idx) {
    double r = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double d = q[i] - getpt(t, idx, i);
        r += d * d;
    }
    return r;
}

static void rs_add(kdo_rs *rs, double dist, size_t index) {
    size_t i;
    for (i = rs->count; i > 0; --i) {
        if (rs->dists[i - 1] > dist) {
            if (i < rs->capacity) {
                rs->dists[i] = rs->dists[i - 1];
                rs->indices[i] = rs->indices[i - 1];
            }
        } else
            break;
    }
    if (i < rs->capacity) {
        rs->dists[i] = dist;
        rs->indices[i] = index;
    }
    if (rs->count < rs->capacity) rs->count++;
}

static void search_level(const kdo_tree *t, kdo_rs *rs, const double *q, int node, double mindist, double dists[3]) {
    const kdo_node *nd = &t->nodes[node];
    if (nd->divfeat < 0) {
        double worst = rs->dists[rs->capacity - 1]; /* cached at leaf entry (:1734) */
        for (uint32_t i = nd->left; i < nd->right; ++i) {
            uint32_t acc = t->vind[i];
            double d = dist2(t, q, acc);
            if (d < worst) rs_add(rs, d, acc);
        }
        return;
    }
    int idx = nd->divfeat;
    double val = q[idx];
    double diff1 = val - nd->divlow;
    double diff2 = val - nd->divhigh;
    int best, other;
    double cut;
    if ((diff1 + diff2) < 0) {
        best = nd->child1;
        other = nd->child2;
        cut = (val - nd->divhigh) * (val - nd->divhigh);
    } else {
        best = nd->child2;
        other = nd->child1;
        cut = (val - nd->divlow) * (val - nd->divlow);
    }
    search_level(t, rs, q, best, mindist, dists);
    double dst = dists[idx];
    mindist = mindist + cut - dst;
    dists[idx] = cut;
    if (mindist * 1.0f <= rs->dists[rs->capacity - 1]) search_level(t, rs, q, other, mindist, dists);
    dists[idx] = dst;
}

/* raw nanoflann answer: min(n, size) entries */
int kdo_search_raw(void *h, double x, double y, double z, int n, int *indices, double *sqdist) {
    kdo_tree *t = (kdo_tree *)h;
    if (t->n == 0 || n <= 0) return 0;
    double q[3] = {x, y, z};
    size_t *ri = (size_t *)malloc(sizeof(size_t) * (size_t)n);
    double *rd = (double *)malloc(sizeof(double) * (size_t)n);
    kdo_rs rs = {ri, rd, (size_t)n, 0};
    rd[n - 1] = DBL_MAX;
    double dists[3] = {0, 0, 0};
    double dist = 0;
    for (int i = 0; i < 3; ++i) {
        if (q[i] < t->bbox[i][0]) {
            dists[i] = (q[i] - t->bbox[i][0]) * (q[i] - t->bbox[i][0]);
            dist += dists[i];
        }
        if (q[i] > t->bbox[i][1]) {
            dists[i] = (q[i] - t->bbox[i][1]) * (q[i] - t->bbox[i][1]);
            dist += dists[i];
        }
    }
    search_level(t, &rs, q, t->root, dist, dists);
    int cnt = (int)rs.count;
    for (int i = 0; i < cnt; ++i) {
        indices[i] = (int)ri[i];
        sqdist[i] = rd[i];
    }
    free(ri);
    free(rd);
    return cnt;
}

/* KDTreeTwo::SearchForNearest: returns num_results per kd_tree_two.h:119-124 (0 when size == n). */
int kdo_search(void *h, double x, double y, double z, int n, int *indices, double *sqdist, float *pts_xyz) {
    kdo_tree *t = (kdo_tree *)h;
    if (t->n == 0) return 0;
    int *ti = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    double *td = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    kdo_search_raw(h, x, y, z, n, ti, td);
    int num = 0;
    if ((long long)t->n < (long long)n) num = (int)t->n;
    else if ((long long)t->n > (long long)n) num = n;
    for (int i = 0; i < num; ++i) {
        indices[i] = ti[i];
        sqdist[i] = td[i];
        if (pts_xyz) {
            pts_xyz[3 * i + 0] = t->pts[3 * (size_t)ti[i] + 0];
            pts_xyz[3 * i + 1] = t->pts[3 * (size_t)ti[i] + 1];
            pts_xyz[3 * i + 2] = t->pts[3 * (size_t)ti[i] + 2];
        }
    }
    free(ti);
    free(td);
    return num;
}

/* Ordered brute force over the FILTERED cloud with the same fp64 operation order; ties resolve to
 * the lowest index (the HIP path's documented tie policy).  Returns min(k, size). */
int kdo_bruteforce(void *h, double x, double y, double z, int k, int *indices, double *sqdist) {
    kdo_tree *t = (kdo_tree *)h;
    if (t->n == 0 || k <= 0) return 0;
    double q[3] = {x, y, z};
    size_t *ri = (size_t *)malloc(sizeof(size_t) * (size_t)k);
    double *rd = (double *)malloc(sizeof(double) * (size_t)k);
    kdo_rs rs = {ri, rd, (size_t)k, 0};
    rd[k - 1] = DBL_MAX;
    for (uint32_t i = 0; i < t->n; ++i) {
        double d = dist2(t, q, i);
        if (d < rd[k - 1] || rs.count < rs.capacity) rs_add(&rs, d, i);
    }
    int cnt = (int)rs.count;
    for (int i = 0; i < cnt; ++i) {
        indices[i] = (int)ri[i];
        sqdist[i] = rd[i];
    }
    free(ri);
    free(rd);
    return cnt;
}

/* Keyframe sweep, FrameKDMap::KeyframeThreadWorker (AM/src/FrameKDMap.cpp:462-485): for every point of the
 * last keyframe a SearchForNearest(pt, 1) in the current frame's tree; outlier when a result exists and
 * sqrt(d2) > th_dist (:468-475); with fewer than th_count outliers nothing happens (:477-479), else the
 * keyframe's tree is rebuilt from the outliers in their original order (:480-485).
 * Returns 1 when rebuilt; *n_outliers = number of outliers. */
int kdo_keyframe_sweep(void *keyframe, void *current, double th_dist, int th_count, int *n_outliers) {
    kdo_tree *kf = (kdo_tree *)keyframe;
    int *out = (int *)malloc(sizeof(int) * (size_t)(kf->n > 0 ? kf->n : 1));
    int m = 0;
    for (uint32_t i = 0; i < kf->n; ++i) {
        int idx[1];
        double d2[1];
        float pf[3];
        int cnt = kdo_search(current, kf->pts[3 * (size_t)i], kf->pts[3 * (size_t)i + 1], kf->pts[3 * (size_t)i + 2], 1, idx,
                             d2, pf);
        if (cnt > 0 && sqrt(d2[0]) > th_dist) out[m++] = (int)i;
    }
    if (n_outliers) *n_outliers = m;
    if (m < th_count) {
        free(out);
        return 0;
    }
    for (int j = 0; j < m; ++j) { /* out[j] >= j: forward in-place copy is safe */
        kf->pts[3 * (size_t)j + 0] = kf->pts[3 * (size_t)out[j] + 0];
        kf->pts[3 * (size_t)j + 1] = kf->pts[3 * (size_t)out[j] + 1];
        kf->pts[3 * (size_t)j + 2] = kf->pts[3 * (size_t)out[j] + 2];
    }
    kf->n = (uint32_t)m;
    build_index(kf);
    free(out);
    return 1;
}
