// The reference's OWN tree, node for node, and its own traversal -- the opt-in "nanoflann tie order" mode of an amk_kd
// (amk_kd_set_tie_order).
//
// Why: the bucketed index (kd_grid.h) returns the exact k nearest, equal squared distances ordered by cloud index.
// nanoflann keeps, of equal distances, the one it VISITS first (KNNResultSet::addPoint, AM/include/nanoflann_two.hpp:
// 219-246), i.e. its answer on ties depends on the shape of its tree and on its traversal.  To return the same INDEX
// lists on tie-heavy clouds (quantised depth on a pixel grid) the tree itself has to be the same.  This header builds
// it on the device and searches it exactly as nanoflann does:
//   exact_build_scene   buildIndex -> computeBoundingBox -> divideTree -> middleSplit_ -> planeSplit
//                       (nanoflann_two.hpp:1518-1541, 1694-1720, 1055-1106, 1197-1245, 1256-1294), leaf size 10
//                       (AM/include/kd_tree_two.h:68)
//   exact_knn_thread    findNeighbors -> computeInitialDistances -> searchLevel + KNNResultSet
//                       (nanoflann_two.hpp:1563-1586, 1296-1315, 1729-1793, 179-255)
// The build is level-synchronous, one workgroup per scene, one WAVEFRONT per tree node: the data passes of a node
// (min/max per dimension, the two Hoare partitions of planeSplit) run 64 points at a time.  planeSplit's sequential
// swap loop pairs the j-th misplaced element from the left with the j-th misplaced element from the right; the same
// permutation is produced here from two order-preserving compactions (ballot + popcount) and a parallel swap, so vAcc_
// ends up identical.  divlow / divhigh are the refined child boxes' bounds on the cut dimension = max of the left /
// min of the right subtree's coordinates, taken when the node is split.
// Cost: ~depth x 8 gathered passes over the cloud; milliseconds per build where the bucketed index takes a fraction
// of one -- a mode for the reference's real frame sizes (<= 3072 points), not for the synthetic 50k-200k clouds.
#pragma once
#include "kd_grid.h"

namespace amk {

constexpr int kExactLeaf = 10;      // kd_tree_two.h:68
constexpr int kExactThreads = 256;  // 4 wavefronts per scene
constexpr int kExactMaxDepth = 48;  // traversal stack (one frame per level); deeper trees fall back to the bucketed index

struct ExactTree {  // one scene
    const float *x, *y, *z;  // index-ordered planes of the NaN-x-filtered cloud
    unsigned *vind;          // vAcc_
    int *feat;               // divfeat, -1 = leaf
    unsigned *left, *right;  // node range [left, right) in vind
    int *child;              // child1; child2 = child1 + 1
    double *low, *high;      // divlow, divhigh
    double *nbbox;           // [node][6] boxes handed down by divideTree (lo0, hi0, lo1, hi1, lo2, hi2)
    double *root_bbox;       // [6] root_bbox_
    unsigned *sa, *sb;       // scratch lists of planeSplit
    int *n_nodes;            // [1] number of nodes; -1: the tree is not available (capacity / depth exceeded)
    int max_nodes;
    __device__ __forceinline__ double val(unsigned i, int dim) const {
        const unsigned a = vind[i];
        return (double)(dim == 0 ? x[a] : (dim == 1 ? y[a] : z[a]));
    }
};

struct ExactPtrs {  // the batch
    const float *x, *y, *z;
    int cap;
    unsigned *vind, *left, *right, *sa, *sb;
    int *feat, *child, *n_nodes;
    double *low, *high, *nbbox, *root_bbox;
    int max_nodes;
    __device__ __forceinline__ ExactTree scene(int s) const {
        ExactTree t;
        const size_t pc = (size_t)s * cap, nc = (size_t)s * max_nodes;
        t.x = x + pc; t.y = y + pc; t.z = z + pc;
        t.vind = vind + pc; t.sa = sa + pc; t.sb = sb + pc;
        t.feat = feat + nc; t.left = left + nc; t.right = right + nc; t.child = child + nc;
        t.low = low + nc; t.high = high + nc; t.nbbox = nbbox + nc * 6;
        t.root_bbox = root_bbox + (size_t)s * 6;
        t.n_nodes = n_nodes + s;
        t.max_nodes = max_nodes;
        return t;
    }
};

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// computeMinMax over positions [lo, hi) of the node (nanoflann_two.hpp:1037-1052), by one wavefront
__device__ __forceinline__ void node_minmax(const ExactTree &T, unsigned lo, unsigned hi, int dim, double &mn, double &mx) {
    const int lane = threadIdx.x & 63;
    double a = DBL_MAX, b = -DBL_MAX;
    for (unsigned i = lo + lane; i < hi; i += 64) {
        const double v = T.val(i, dim);
        a = fmin(a, v);
        b = fmax(b, v);
    }
    mn = wave_min_f64(a);
    mx = wave_max_f64(b);
}

// One Hoare partition of planeSplit on the node positions [lo, hi) (absolute positions in vind): elements with
// pred = true end up in front.  STRICT selects the predicate of the first loop (val < cutval), else the second
// (val <= cutval).  Returns the number of pred elements (lim - lo).
template <bool STRICT>
__device__ __forceinline__ unsigned hoare_partition(const ExactTree &T, unsigned lo, unsigned hi, int dim, double cutval) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto pred = [&](unsigned i) {
        const double v = T.val(i, dim);
        return STRICT ? v < cutval : v <= cutval;
    };
    unsigned cnt = 0;
    for (unsigned base = lo; base < hi; base += 64) {
        const unsigned i = base + lane;
        cnt += __popcll(__ballot(i < hi && pred(i)));
    }
    const unsigned lim = lo + cnt;
    // misplaced on the left: positions in [lo, lim) with !pred, ascending  -> sa[lo + j]
    unsigned ml = 0;
    for (unsigned base = lo; base < lim; base += 64) {
        const unsigned i = base + lane;
        const bool f = i < lim && !pred(i);
        const unsigned long long m = __ballot(f);
        if (f) T.sa[lo + ml + __popcll(m & lt)] = i;
        ml += __popcll(m);
    }
    // misplaced on the right: positions in [lim, hi) with pred, DESCENDING  -> sb[lo + j]
    unsigned mr = 0;
    for (unsigned top = hi; top > lim; top = top > lim + 64 ? top - 64 : lim) {
        const bool in = top >= lim + 1 + (unsigned)lane;  // position top - 1 - lane >= lim
        const unsigned i = top - 1 - lane;
        const bool f = in && pred(i);
        const unsigned long long m = __ballot(f);
        if (f) T.sb[lo + mr + __popcll(m & lt)] = i;
        mr += __popcll(m);
    }
    __threadfence_block();  // the lists were written by other lanes
    for (unsigned j = lane; j < ml; j += 64) {  // ml == mr: the j-th from the left swaps with the j-th from the right
        const unsigned a = T.sa[lo + j], b = T.sb[lo + j];
        const unsigned ta = T.vind[a], tb = T.vind[b];
        T.vind[a] = tb;
        T.vind[b] = ta;
    }
    __threadfence_block();
    return cnt;
}

// divideTree for node `id` (one wavefront): a leaf is marked, an inner node is split and its two children are appended
__device__ __forceinline__ void exact_process_node(const ExactTree &T, int id, int *n_nodes_lds, int *overflow) {
    const int lane = threadIdx.x & 63;
    const unsigned l = T.left[id], r = T.right[id], count = r - l;
    if (count <= (unsigned)kExactLeaf) {
        if (lane == 0) T.feat[id] = -1;
        return;
    }
    double bb[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) { bb[d][0] = T.nbbox[(size_t)id * 6 + 2 * d]; bb[d][1] = T.nbbox[(size_t)id * 6 + 2 * d + 1]; }
    // middleSplit_ (:1197-1245)
    const double EPS = 0.00001;
    double max_span = bb[0][1] - bb[0][0];
    for (int d = 1; d < 3; ++d) {
        const double span = bb[d][1] - bb[d][0];
        if (span > max_span) max_span = span;
    }
    double max_spread = -1.0, min_elem = 0.0, max_elem = 0.0;
    int cutfeat = 0;
    for (int d = 0; d < 3; ++d) {
        const double span = bb[d][1] - bb[d][0];
        if (span > (1 - EPS) * max_span) {
            double mn, mx;
            node_minmax(T, l, r, d, mn, mx);
            const double spread = mx - mn;
            if (spread > max_spread) { cutfeat = d; max_spread = spread; min_elem = mn; max_elem = mx; }
        }
    }
    const double split_val = (bb[cutfeat][0] + bb[cutfeat][1]) / 2;
    const double cutval = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
    // planeSplit (:1256-1294)
    const unsigned lim1 = hoare_partition<true>(T, l, r, cutfeat, cutval);
    const unsigned lim2 = lim1 + hoare_partition<false>(T, l + lim1, r, cutfeat, cutval);
    const unsigned half = count / 2;
    const unsigned idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
    // refined boxes of the children on the cut dimension (:1086,1096-1102)
    double dlo, dhi, t0, t1;
    node_minmax(T, l, l + idx, cutfeat, t0, dlo);      // divlow  = left child's high
    node_minmax(T, l + idx, r, cutfeat, dhi, t1);      // divhigh = right child's low
    int c = 0;
    if (lane == 0) {
        c = atomicAdd(n_nodes_lds, 2);
        if (c + 2 > T.max_nodes) { *overflow = 1; c = -1; }
    }
    c = __shfl(c, 0);
    if (c < 0) return;
    if (lane == 0) {
        T.feat[id] = cutfeat; T.child[id] = c; T.low[id] = dlo; T.high[id] = dhi;
        T.left[c] = l; T.right[c] = l + idx; T.left[c + 1] = l + idx; T.right[c + 1] = r;
    }
    if (lane < 6) {
        const int d = lane >> 1, hi = lane & 1;
        double vl = bb[d][hi], vr = bb[d][hi];
        if (d == cutfeat && hi == 1) vl = cutval;  // left_bbox[cutfeat].high = cutval
        if (d == cutfeat && hi == 0) vr = cutval;  // right_bbox[cutfeat].low = cutval
        T.nbbox[(size_t)c * 6 + lane] = vl;
        T.nbbox[(size_t)(c + 1) * 6 + lane] = vr;
    }
}

// buildIndex for scene s; called by every thread of a kExactThreads block.  n = cloud.pts.size().
__device__ __forceinline__ void exact_build_scene(const ExactTree &T, int n) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = kExactThreads / 64;
    __shared__ int n_nodes_lds, overflow, head, tail;
    __shared__ double red[kExactThreads / 64][6];
    for (int i = tid; i < n; i += kExactThreads) T.vind[i] = i;  // init_vind
    // computeBoundingBox (:1694-1720)
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = tid; i < n; i += kExactThreads) {
        const double v[3] = {(double)T.x[i], (double)T.y[i], (double)T.z[i]};
#pragma unroll
        for (int d = 0; d < 3; ++d) { lo[d] = fmin(lo[d], v[d]); hi[d] = fmax(hi[d], v[d]); }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = wave_min_f64(lo[d]); hi[d] = wave_max_f64(hi[d]); }
    if (lane == 0)
        for (int d = 0; d < 3; ++d) { red[w][2 * d] = lo[d]; red[w][2 * d + 1] = hi[d]; }
    if (tid == 0) { n_nodes_lds = n > 0 ? 1 : 0; overflow = 0; head = 0; tail = n > 0 ? 1 : 0; }
    __syncthreads();
    if (tid < 6) {
        double v = red[0][tid];
        for (int j = 1; j < nw; ++j) v = (tid & 1) ? fmax(v, red[j][tid]) : fmin(v, red[j][tid]);
        T.root_bbox[tid] = v;
        if (n > 0) T.nbbox[tid] = v;
    }
    if (tid == 0 && n > 0) { T.left[0] = 0; T.right[0] = (unsigned)n; }
    __threadfence_block();
    __syncthreads();
    while (head < tail) {  // one level of divideTree per round
        const int h = head, t = tail;
        for (int id = h + w; id < t; id += nw) exact_process_node(T, id, &n_nodes_lds, &overflow);
        __threadfence_block();
        __syncthreads();
        if (tid == 0) { head = t; tail = overflow ? t : n_nodes_lds; }
        __syncthreads();
    }
    if (tid == 0) *T.n_nodes = overflow ? -1 : n_nodes_lds;
}

// The traversal's stack: one frame per tree level -- either "the other child of this node is still to be considered"
// (kind 0: node, cut dimension, cut distance, mindist at the node) or, once that child has been entered, "restore
// dists[idx]" (kind 1, same slot).  Storage is the caller's: private arrays for the thread-per-query kernels (scratch
// memory), LDS for the single-lane re-query inside the plan kernels.
struct ExactStack {
    int *node, *ik;  // ik = idx | kind << 2
    double *a, *b;   // a: cut distance (kind 0) / saved dists[idx] (kind 1); b: mindist at the node
};
struct ExactStackStorage {  // kExactMaxDepth frames
    int node[kExactMaxDepth], ik[kExactMaxDepth];
    double a[kExactMaxDepth], b[kExactMaxDepth];
    __device__ __forceinline__ ExactStack view() { return ExactStack{node, ik, a, b}; }
};

// findNeighbors for one query by ONE thread, exactly nanoflann's traversal.  rd / ri: the KNNResultSet arrays (k
// entries).  Returns the number of results (min(k, size)), or -1 when the tree is deeper than the stack (fallback).
__device__ __forceinline__ int exact_knn_thread(const ExactTree &T, double qx, double qy, double qz, int k, double *rd,
                                                int *ri, const ExactStack st) {
    const int nn = *T.n_nodes;
    if (nn <= 0) return nn < 0 ? -1 : 0;
    const double q[3] = {qx, qy, qz};
    int count = 0;
    rd[k - 1] = DBL_MAX;  // KNNResultSet::init (:196-202)
    double dists[3] = {0.0, 0.0, 0.0};
    double mind = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {  // computeInitialDistances (:1296-1315)
        const double blo = T.root_bbox[2 * d], bhi = T.root_bbox[2 * d + 1];
        if (q[d] < blo) { dists[d] = (q[d] - blo) * (q[d] - blo); mind += dists[d]; }
        if (q[d] > bhi) { dists[d] = (q[d] - bhi) * (q[d] - bhi); mind += dists[d]; }
    }
    auto set_dist = [&](int idx, double v) {  // dists[] stays in registers: no dynamic indexing
        dists[0] = idx == 0 ? v : dists[0];
        dists[1] = idx == 1 ? v : dists[1];
        dists[2] = idx == 2 ? v : dists[2];
    };
    int sp = 0, node = 0;  // explicit stack of the recursion of searchLevel (:1729-1793)
    for (;;) {
        for (int idx = T.feat[node]; idx >= 0; idx = T.feat[node]) {  // descend along the best children
            const double val = idx == 0 ? qx : (idx == 1 ? qy : qz);
            const double diff1 = val - T.low[node], diff2 = val - T.high[node];
            int best, other;
            double cut;
            if ((diff1 + diff2) < 0) { best = T.child[node]; other = best + 1; cut = diff2 * diff2; }
            else { other = T.child[node]; best = other + 1; cut = diff1 * diff1; }
            if (sp >= kExactMaxDepth) return -1;
            st.node[sp] = other; st.ik[sp] = idx; st.a[sp] = cut; st.b[sp] = mind;
            ++sp;
            node = best;
        }
        {   // leaf (:1733-1751): worst distance cached at entry
            const double worst = rd[k - 1];
            const unsigned lf = T.left[node], rt = T.right[node];
            for (unsigned i = lf; i < rt; ++i) {
                const unsigned a = T.vind[i];
                const double dist = sq_dist(qx, qy, qz, T.x[a], T.y[a], T.z[a]);
                if (dist < worst) {  // KNNResultSet::addPoint (:219-246)
                    int j;
                    for (j = count; j > 0; --j) {
                        if (rd[j - 1] > dist) {
                            if (j < k) { rd[j] = rd[j - 1]; ri[j] = ri[j - 1]; }
                        } else {
                            break;
                        }
                    }
                    if (j < k) { rd[j] = dist; ri[j] = (int)a; }
                    if (count < k) ++count;
                }
            }
        }
        bool go = false;
        while (sp > 0) {  // unwind
            --sp;
            const int ik = st.ik[sp], idx = ik & 3;
            if (ik >> 2) { set_dist(idx, st.a[sp]); continue; }  // kind 1: dists[idx] = dst (:1791)
            const double cut = st.a[sp], dst = idx == 0 ? dists[0] : (idx == 1 ? dists[1] : dists[2]);
            const double m2 = st.b[sp] + cut - dst;
            if (m2 * 1.0f <= rd[k - 1]) {  // (epsError = 1 + eps, eps = 0: :1572,1780)
                set_dist(idx, cut);
                node = st.node[sp];
                st.ik[sp] = idx | 4; st.a[sp] = dst;  // restore once the other child returns
                ++sp;
                mind = m2;
                go = true;
                break;
            }
        }
        if (!go) break;
    }
    return count;
}

}  // namespace amk

// host side: the batch pointers of a handle whose reference-shaped tree exists (amk_common.h: amk_kd)
inline amk::ExactPtrs amk_exact_ptrs(amk_kd *kd) {
    amk::ExactPtrs ep;
    ep.x = kd->x.p; ep.y = kd->y.p; ep.z = kd->z.p; ep.cap = kd->cap;
    ep.vind = kd->ex_vind.p; ep.left = kd->ex_left.p; ep.right = kd->ex_right.p; ep.sa = kd->ex_sa.p; ep.sb = kd->ex_sb.p;
    ep.feat = kd->ex_feat.p; ep.child = kd->ex_child.p; ep.n_nodes = kd->ex_nn.p;
    ep.low = kd->ex_low.p; ep.high = kd->ex_high.p; ep.nbbox = kd->ex_nbbox.p; ep.root_bbox = kd->ex_root.p;
    ep.max_nodes = kd->ex_max_nodes;
    return ep;
}
