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
// The build of a scene runs in two kernels, one workgroup each.  exact_build_top (8 wavefronts): every node of more than
// kExactBigNode points -- the top four levels of a 50k-point tree -- is split by the WHOLE workgroup, level by level.
// exact_build_rest (16 wavefronts): a ring of open nodes in LDS; a wavefront pops a node and
//   * > kExactWindow points: splits it in global memory (exact_process_node) and pushes the two children,
//   * 65 .. kExactWindow points: builds everything below it on a copy of its window in LDS (exact_window_wave),
//   * <= 64 points: builds everything below it with the points in registers (exact_subtree_wave).
// The data passes of a node (min / max per dimension, the statistics of the cut, the two Hoare partitions of planeSplit) read
// the coordinates from planes kept IN vAcc_ ORDER (permuted together with vAcc_): coalesced streams instead of gathers
// through the permutation.  planeSplit's sequential swap loop pairs the j-th misplaced element from the left with the j-th
// misplaced element from the right; the same permutation is produced here from two order-preserving compactions (ballot +
// popcount; across the workgroup through an LDS prefix for big nodes) and a parallel swap, so vAcc_ ends up identical.
// divlow / divhigh are the refined child boxes' bounds on the cut dimension = max of the left / min of the right subtree's
// coordinates, taken from the statistics of the cut when the node is split.  Round 2 (wave per node, gathers): 23 ms per
// 256 x 50k-point build, round 3: 5.1 ms, round 4: 2.5 ms (0.2 ms at the reference's 3072 points): DESIGN.md section 4.
#pragma once
#include <cstring>
#include "kd_grid.h"

namespace amk {

constexpr int kExactLeaf = 10;      // kd_tree_two.h:68
#ifndef AMK_EXACT_THREADS
#define AMK_EXACT_THREADS 1024
#endif
constexpr int kExactThreads = AMK_EXACT_THREADS; // 16 wavefronts per scene (exact_build_rest)
#ifndef AMK_EXACT_TOP_THREADS
#define AMK_EXACT_TOP_THREADS 512
#endif
constexpr int kExactTopThreads = AMK_EXACT_TOP_THREADS; // exact_build_top: 8 wavefronts with a 256-register budget (at 1024 threads / 128 registers the
                                      // block-wide split spilled ~500 B per lane to scratch)
constexpr int kExactBigNode = 4096; // more points than this: the node is split by the whole workgroup
constexpr int kExactTodo = -2;      // feat of a node divideTree has not visited yet (a leaf is -1)
constexpr int kExactSubtree = 64;   // a node of at most this many points: its whole subtree is built by one wavefront, in registers
constexpr int kExactSubStack = 8;   // pending right children of that wavefront: only children of more than kExactLeaf points whose sibling
                                    // is one too are ever pending, <= 64 / 11 of them
#ifndef AMK_EXACT_WINDOW
#define AMK_EXACT_WINDOW 416
#endif
constexpr int kExactWindow = AMK_EXACT_WINDOW;   // a node of at most this many points: its whole subtree by one wavefront on a copy in LDS
constexpr int kExactWinStack = 16;  // pending nodes of more than kExactSubtree points inside a window (<= kExactWindow / 65 + the current one)
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
    // the coordinates in vAcc_ order: pc[d * pstride + i] == coordinate d of point vind[i] (kept so by every swap of the
    // build).  ONE base pointer and a plane stride, not three pointers: `dim == 0 ? px[i] : dim == 1 ? py[i] : pz[i]` made the
    // compiler select the ADDRESS of the member and load the pointer from the struct, which then lived in scratch memory
    // (160 B per lane; scratch on a queue is what tests/test_abi.py forbids outside the opt-in thread-per-query kernels).
    float *pc;
    size_t pstride;
    int *n_nodes;            // [1] number of nodes; -1: the tree is not available (capacity / depth exceeded)
    int max_nodes;
    int lists_local;         // sa / sb are a wavefront's own short buffers, indexed from 0 (a window in LDS); else [lo + j]
    __device__ __forceinline__ float *plane(int dim) const { return pc + (size_t)dim * pstride; }
    __device__ __forceinline__ double val(unsigned i, int dim) const { return (double)pc[(size_t)dim * pstride + i]; }
};

struct ExactPtrs {  // the batch
    const float *x, *y, *z;
    int cap;
    unsigned *vind, *left, *right, *sa, *sb;
    float *pc;        // [3][S][cap]
    size_t pstride;   // S * cap
    int *feat, *child, *n_nodes;
    double *low, *high, *nbbox, *root_bbox;
    int max_nodes;
    __device__ __forceinline__ ExactTree scene(int s) const {
        ExactTree t;
        const size_t pc = (size_t)s * cap, nc = (size_t)s * max_nodes;
        t.x = x + pc; t.y = y + pc; t.z = z + pc;
        t.vind = vind + pc; t.sa = sa + pc; t.sb = sb + pc;
        t.pc = this->pc + pc; t.pstride = pstride;
        t.feat = feat + nc; t.left = left + nc; t.right = right + nc; t.child = child + nc;
        t.low = low + nc; t.high = high + nc; t.nbbox = nbbox + nc * 6;
        t.root_bbox = root_bbox + (size_t)s * 6;
        t.n_nodes = n_nodes + s;
        t.max_nodes = max_nodes;
        t.lists_local = 0;
        return t;
    }
};

// min / max over the 64 lanes, every lane gets the result.  DPP inside the rows of 16 lanes (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror: each level is valid because every lane of the lower level already holds that level's result)
// and four v_readlane for the rows -- ~10 short-latency VALU operations where six __shfl_xor rounds on a double are twelve
// ds_bpermute_b32 with an LDS-crossbar round trip each.  The bottom of the tree (exact_subtree_wave: 90 % of the nodes) is a
// chain of such reductions: round 4 measured the build's lower levels at 5 of its 6 ms.
template <int CTRL>
__device__ __forceinline__ double exact_dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_minmax_f64(double v) {
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : fmin(a, b); };
    v = op(v, exact_dpp_f64<0xB1>(v));    // quad_perm:[1,0,3,2]
    v = op(v, exact_dpp_f64<0x4E>(v));    // quad_perm:[2,3,0,1]
    v = op(v, exact_dpp_f64<0x141>(v));   // row_half_mirror
    v = op(v, exact_dpp_f64<0x140>(v));   // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ double wave_min_f64(double v) { return wave_minmax_f64<false>(v); }
__device__ __forceinline__ double wave_max_f64(double v) { return wave_minmax_f64<true>(v); }

// ---- the collectives of the wavefront that works on a node (nodes of more than kExactBigNode points: exact_process_big_node)
struct WaveCoop {
    static constexpr int kN = 64;
    __device__ __forceinline__ int tid() const { return threadIdx.x & 63; }
    __device__ __forceinline__ void sync() const { __threadfence_block(); }   // lanes of one wave: memory order only
    // exclusive prefix of `f` over the group (in tid order) and the group's total
    __device__ __forceinline__ unsigned scan(bool f, unsigned &total) const {
        const unsigned long long m = __ballot(f);
        total = __popcll(m);
        return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
    }
    __device__ __forceinline__ unsigned sum(unsigned v) const {   // DPP inside the rows, v_readlane across them
        int x = (int)v;
        x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);
        return (unsigned)(__builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
                          __builtin_amdgcn_readlane(x, 48));
    }
    __device__ __forceinline__ double min(double v) const { return wave_min_f64(v); }
    __device__ __forceinline__ double max(double v) const { return wave_max_f64(v); }
    __device__ __forceinline__ int bcast(int v) const { return __builtin_amdgcn_readfirstlane(v); }
};
// One Hoare partition of planeSplit on the node positions [lo, hi) (absolute positions in vind): elements with
// pred = true end up in front.  STRICT selects the predicate of the first loop (val < cutval), else the second
// (val <= cutval).  Returns the number of pred elements (lim - lo).
// set bits of a wave mask below this lane (v_mbcnt_lo / _hi)
__device__ __forceinline__ int mask_rank_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
template <bool STRICT, class G>
__device__ __forceinline__ unsigned hoare_partition(const G &g, const ExactTree &T, unsigned lo, unsigned hi, int dim,
                                                    double cutval, int known_cnt) {
    const unsigned tid = (unsigned)g.tid();
    const unsigned cnt = (unsigned)known_cnt;   // (the caller's statistics pass counted)
    const unsigned lim = lo + cnt;
    const unsigned lb = T.lists_local ? 0u : lo;   // where this partition's lists start in sa / sb
    // ONE forward pass, four chunks of 64 in flight: misplaced on the left (positions in [lo, lim) with !pred) -> sa[lo + j],
    // misplaced on the right (positions in [lim, hi) with pred) -> sb[lo + j], both ASCENDING
    const float *pl = T.plane(dim);
    unsigned ml = 0, mr = 0;
    for (unsigned base = lo; base < hi; base += 4 * G::kN) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = base + G::kN * u + tid;
            v[u] = i < hi ? pl[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = base + G::kN * u + tid;
            const bool in = i < hi;
            const bool pr = in && (STRICT ? (double)v[u] < cutval : (double)v[u] <= cutval);
            const bool fl = in && i < lim && !pr, fr = in && i >= lim && pr;
            const unsigned long long bl = __ballot(fl), br = __ballot(fr);
            if (fl) T.sa[lb + ml + mask_rank_below(bl)] = i;
            if (fr) T.sb[lb + mr + mask_rank_below(br)] = i;
            ml += __popcll(bl); mr += __popcll(br);
        }
    }
    g.sync();  // the lists were written by other threads of the group
    // planeSplit's swap loop pairs the j-th misplaced from the left (ascending) with the j-th from the right (DESCENDING); ml == mr
#pragma unroll 2
    for (unsigned j = tid; j < ml; j += G::kN) {
        const unsigned a = T.sa[lb + j], b = T.sb[lb + (ml - 1 - j)];
        const unsigned ta = T.vind[a], tb = T.vind[b];
        T.vind[a] = tb; T.vind[b] = ta;
        float *p0 = T.plane(0), *p1 = T.plane(1), *p2 = T.plane(2);
        const float xa = p0[a], xb = p0[b], ya = p1[a], yb = p1[b], za = p2[a], zb = p2[b];
        p0[a] = xb; p0[b] = xa; p1[a] = yb; p1[b] = ya; p2[a] = zb; p2[b] = za;
    }
    g.sync();
    return cnt;
}

// divideTree for node `id` (by the group g): a leaf is marked, an inner node is split and its two children are appended.
// Eight passes over the node (round 3: thirteen): min / max of the three dimensions in ONE pass (the reference computes only
// the dimensions whose box span qualifies, :1201-1225 -- the others are simply not used), ONE statistics pass on the cut
// dimension -- #{v < cut}, #{v == cut}, max{v < cut}, min{v > cut} -- that gives both partition sizes and divlow / divhigh
// (see exact_process_big_node), then the two Hoare partitions (two list passes and a swap each).
// Returns child1 of the split (child2 = child1 + 1), -1 for a leaf or when the node capacity is exhausted.
template <class G>
__device__ __forceinline__ int exact_process_node(const G &g, const ExactTree &T, int id, int *n_nodes_lds, int *overflow) {
    const int tid = g.tid();
    const unsigned l = T.left[id], r = T.right[id], count = r - l;
    if (count <= (unsigned)kExactLeaf) {
        if (tid == 0) T.feat[id] = -1;
        return -1;
    }
    // the node's box in six scalars: a private array indexed by the cut dimension would live in scratch (or be promoted
    // to 48 B of LDS per thread)
    const double b0l = T.nbbox[(size_t)id * 6 + 0], b0h = T.nbbox[(size_t)id * 6 + 1], b1l = T.nbbox[(size_t)id * 6 + 2],
                 b1h = T.nbbox[(size_t)id * 6 + 3], b2l = T.nbbox[(size_t)id * 6 + 4], b2h = T.nbbox[(size_t)id * 6 + 5];
#define blo(d) ((d) == 0 ? b0l : ((d) == 1 ? b1l : b2l))
#define bhi(d) ((d) == 0 ? b0h : ((d) == 1 ? b1h : b2h))
    // middleSplit_ (:1197-1245): computeMinMax only of the dimensions whose box span qualifies (usually one), in one pass
    const double EPS = 0.00001;
    double max_span = b0h - b0l;
#pragma unroll
    for (int d = 1; d < 3; ++d) {
        const double span = bhi(d) - blo(d);
        if (span > max_span) max_span = span;
    }
    const bool q0 = (b0h - b0l) > (1 - EPS) * max_span, q1 = (b1h - b1l) > (1 - EPS) * max_span, q2 = (b2h - b2l) > (1 - EPS) * max_span;
    double mn0 = DBL_MAX, mn1 = DBL_MAX, mn2 = DBL_MAX, mx0 = -DBL_MAX, mx1 = -DBL_MAX, mx2 = -DBL_MAX;
#pragma unroll 4
    for (unsigned i = l + tid; i < r; i += G::kN) {
        if (q0) { const double x = T.val(i, 0); mn0 = fmin(mn0, x); mx0 = fmax(mx0, x); }
        if (q1) { const double y = T.val(i, 1); mn1 = fmin(mn1, y); mx1 = fmax(mx1, y); }
        if (q2) { const double z = T.val(i, 2); mn2 = fmin(mn2, z); mx2 = fmax(mx2, z); }
    }
    if (q0) { mn0 = g.min(mn0); mx0 = g.max(mx0); }
    if (q1) { mn1 = g.min(mn1); mx1 = g.max(mx1); }
    if (q2) { mn2 = g.min(mn2); mx2 = g.max(mx2); }
    double max_spread = -1.0, min_elem = 0.0, max_elem = 0.0;
    int cutfeat = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double span = bhi(d) - blo(d);
        if (span > (1 - EPS) * max_span) {
            const double mn = d == 0 ? mn0 : (d == 1 ? mn1 : mn2), mx = d == 0 ? mx0 : (d == 1 ? mx1 : mx2);
            const double spread = mx - mn;
            if (spread > max_spread) { cutfeat = d; max_spread = spread; min_elem = mn; max_elem = mx; }
        }
    }
    const double split_val = (blo(cutfeat) + bhi(cutfeat)) / 2;
    const double cutval = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
    // the statistics of the cut
    unsigned na = 0, nb = 0;
    double maxa = -DBL_MAX, minc = DBL_MAX;
#pragma unroll 4
    for (unsigned i = l + tid; i < r; i += G::kN) {
        const double v = T.val(i, cutfeat);
        na += v < cutval ? 1u : 0u; nb += v == cutval ? 1u : 0u;
        maxa = v < cutval ? fmax(maxa, v) : maxa;
        minc = v > cutval ? fmin(minc, v) : minc;
    }
    na = g.sum(na); nb = g.sum(nb); maxa = g.max(maxa); minc = g.min(minc);
    // planeSplit (:1256-1294)
    const unsigned lim1 = na > 0 ? hoare_partition<true>(g, T, l, r, cutfeat, cutval, (int)na) : 0u;
    // (no element equal to the cut -- the rule on continuous coordinates, where the cut is a box midpoint: the second loop of
    // planeSplit finds everything in place)
    const unsigned lim2 = lim1 + (nb > 0 ? hoare_partition<false>(g, T, l + lim1, r, cutfeat, cutval, (int)nb) : 0u);
    const unsigned half = count / 2;
    const unsigned idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
    // refined boxes of the children on the cut dimension (:1086,1096-1102)
    const double dlo = idx > lim1 ? cutval : maxa;     // divlow  = the left child's high
    const double dhi = idx < lim2 ? cutval : minc;     // divhigh = the right child's low
    int c = 0;
    if (tid == 0) {
        c = atomicAdd(n_nodes_lds, 2);
        if (c + 2 > T.max_nodes) { *overflow = 1; c = -1; }
    }
    c = g.bcast(c);
    if (c < 0) return -1;
    if (tid == 0) {
        T.feat[id] = cutfeat; T.child[id] = c; T.low[id] = dlo; T.high[id] = dhi;
        T.left[c] = l; T.right[c] = l + idx; T.left[c + 1] = l + idx; T.right[c + 1] = r;
        T.feat[c] = kExactTodo; T.feat[c + 1] = kExactTodo;
    }
    if (tid < 6) {
        const int d = tid >> 1, hi = tid & 1;
        double vl = hi ? bhi(d) : blo(d), vr = vl;
        if (d == cutfeat && hi == 1) vl = cutval;  // left_bbox[cutfeat].high = cutval
        if (d == cutfeat && hi == 0) vr = cutval;  // right_bbox[cutfeat].low = cutval
        T.nbbox[(size_t)c * 6 + tid] = vl;
        T.nbbox[(size_t)(c + 1) * 6 + tid] = vr;
    }
    return c;
#undef blo
#undef bhi
}

// divideTree for node `id` AND everything below it, by one wavefront: the node's <= 64 points live in registers (lane i =
// position left + i of vAcc_: its index and its three coordinates), every computeMinMax is a wave reduction over the lanes
// of the current range, every planeSplit the same left / right misplaced-pair swap as hoare_partition with the pairing
// through 64 LDS bytes per list -- the bottom levels of the tree (90 % of its nodes) cost a few hundred instructions per
// node instead of a dozen dependent round trips to memory.  Same splits, same order of vAcc_ inside the leaves.
struct ExactSubLds {   // per wavefront
    unsigned char il[64], ir[64];                 // lane of the j-th misplaced element on the left / on the right
    int l[kExactSubStack], r[kExactSubStack], id[kExactSubStack];
    double bb[kExactSubStack][6];
};
// min / max of a float over the wavefront (every lane gets it): the coordinates ARE floats, min / max do not round, so the
// reductions of the register-resident subtrees run on one register per value (v_min_f32 / v_max_f32 with a DPP operand: one
// instruction per level where the double version needs two moves and the operation) and are widened afterwards.
template <bool MAX>
__device__ __forceinline__ float wave_minmax_f32(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
    auto dpp = [](float x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), C, 0xF, 0xF, true));
    };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm:[1,0,3,2]
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm:[2,3,0,1]
    v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
    v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ void exact_subtree_wave(const ExactTree &T, int root, int *n_nodes_lds, int *overflow, ExactSubLds *ws) {
    const int lane = threadIdx.x & 63;
    const unsigned L0 = T.left[root];
    const int W = (int)(T.right[root] - L0);
    const bool mine = lane < W;
    unsigned vi = mine ? T.vind[L0 + lane] : 0u;
    float px = mine ? T.plane(0)[L0 + lane] : 0.f, py = mine ? T.plane(1)[L0 + lane] : 0.f, pz = mine ? T.plane(2)[L0 + lane] : 0.f;
    int sp = 0;   // stack of pending nodes (LDS); the current node is in registers
    int l = 0, r = W, id = root;
    double b0l = T.nbbox[(size_t)root * 6 + 0], b0h = T.nbbox[(size_t)root * 6 + 1], b1l = T.nbbox[(size_t)root * 6 + 2],
           b1h = T.nbbox[(size_t)root * 6 + 3], b2l = T.nbbox[(size_t)root * 6 + 4], b2h = T.nbbox[(size_t)root * 6 + 5];
    auto coordf = [&](int d) { return d == 0 ? px : (d == 1 ? py : pz); };
    // one Hoare partition of planeSplit over lanes [a, b): returns the number of pred lanes; the registers are permuted
    auto partition = [&](int a, int b, float v, double cutval, bool strict) {
        const bool in = lane >= a && lane < b;
        const bool pred = in && (strict ? (double)v < cutval : (double)v <= cutval);
        const unsigned long long mp = __ballot(pred);
        const int cnt = __popcll(mp), lim = a + cnt;
        const bool ml = in && lane < lim && !pred, mr = in && lane >= lim && pred;
        const unsigned long long mml = __ballot(ml), mmr = __ballot(mr);
        if (mml == 0) return cnt;   // nothing misplaced (as many on the right as on the left): e.g. no element equals the cut
        const int jl = mask_rank_below(mml);                                   // ascending rank among the misplaced on the left
        const int jr = __popcll(mmr) - mask_rank_below(mmr) - 1;               // descending rank among the misplaced on the right
        if (ml) ws->il[jl] = (unsigned char)lane;
        if (mr) ws->ir[jr] = (unsigned char)lane;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the two lists: written and read by lanes of this wave
        int partner = lane;
        if (ml) partner = ws->ir[jl];
        if (mr) partner = ws->il[jr];
        vi = __shfl(vi, partner); px = __shfl(px, partner); py = __shfl(py, partner); pz = __shfl(pz, partner);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (the lists are rewritten by the next partition)
        return cnt;
    };
    for (;;) {
        const int count = r - l;   // > kExactLeaf, except possibly at the root: a leaf child is marked when its parent is split
        if (count <= kExactLeaf) {
            if (lane == 0) T.feat[id] = -1;
            break;                 // (only the root can get here: nothing is pending)
        }
#define blo(d) ((d) == 0 ? b0l : ((d) == 1 ? b1l : b2l))
#define bhi(d) ((d) == 0 ? b0h : ((d) == 1 ? b1h : b2h))
        // middleSplit_ (:1197-1245)
        const double EPS = 0.00001;
        double max_span = b0h - b0l;
#pragma unroll
        for (int d = 1; d < 3; ++d) {
            const double span = bhi(d) - blo(d);
            if (span > max_span) max_span = span;
        }
        const bool in = lane >= l && lane < r;
        double max_spread = -1.0, min_elem = 0.0, max_elem = 0.0;
        int cutfeat = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double span = bhi(d) - blo(d);
            if (span > (1 - EPS) * max_span) {   // computeMinMax over the lanes of the node
                const float v = coordf(d);
                const double mn = (double)wave_minmax_f32<false>(in ? v : __builtin_inff());
                const double mx = (double)wave_minmax_f32<true>(in ? v : -__builtin_inff());
                const double spread = mx - mn;
                if (spread > max_spread) { cutfeat = d; max_spread = spread; min_elem = mn; max_elem = mx; }
            }
        }
        const double split_val = (blo(cutfeat) + bhi(cutfeat)) / 2;
        const double cutval = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
        // divlow / divhigh from the values on both sides of the cut, as exact_process_node does: the left child is
        // [all v < cut | idx - lim1 elements equal to cut]  (an empty side's +-inf is never used: see dlo / dhi)
        const float vc = coordf(cutfeat);
        const double maxa = (double)wave_minmax_f32<true>(in && (double)vc < cutval ? vc : -__builtin_inff());
        const double minc = (double)wave_minmax_f32<false>(in && (double)vc > cutval ? vc : __builtin_inff());
        // planeSplit (:1256-1294)
        const int lim1 = partition(l, r, vc, cutval, true);
        const int lim2 = lim1 + partition(l + lim1, r, coordf(cutfeat), cutval, false);
        const int half = count / 2;
        const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
        const double dlo = idx > lim1 ? cutval : maxa;     // divlow  = left child's high
        const double dhi = idx < lim2 ? cutval : minc;     // divhigh = right child's low
        int c = 0;
        if (lane == 0) {
            c = atomicAdd(n_nodes_lds, 2);
            if (c + 2 > T.max_nodes) { *overflow = 1; c = -1; }
        }
        c = __builtin_amdgcn_readfirstlane(c);
        if (c < 0) break;
        // A child of at most kExactLeaf points is a leaf: marked here, never pushed (nanoflann_two.hpp:1061).  The stack therefore
        // only ever holds children of more than kExactLeaf points whose sibling is one too: <= 64 / 11 entries.
        const bool left_leaf = idx <= kExactLeaf, right_leaf = count - idx <= kExactLeaf;
        {   // the node's record and its children's ranges: one store, lane j writes field j
            int *ptr = T.feat + id;
            int val = cutfeat;
            if (lane == 1) { ptr = T.child + id; val = c; }
            if (lane == 2) { ptr = (int *)T.left + c; val = (int)(L0 + l); }
            if (lane == 3) { ptr = (int *)T.right + c; val = (int)(L0 + l + idx); }
            if (lane == 4) { ptr = (int *)T.left + c + 1; val = (int)(L0 + l + idx); }
            if (lane == 5) { ptr = (int *)T.right + c + 1; val = (int)(L0 + r); }
            if (lane == 6) { ptr = T.feat + c; val = left_leaf ? -1 : kExactTodo; }
            if (lane == 7) { ptr = T.feat + c + 1; val = right_leaf ? -1 : kExactTodo; }
            if (lane < 8) *ptr = val;
            if (lane < 2) (lane == 0 ? T.low : T.high)[id] = lane == 0 ? dlo : dhi;
        }
        if (!left_leaf && !right_leaf) {   // right child waits: box = this box with low[cutfeat] = cutval
            if (sp >= kExactSubStack) { if (lane == 0) *overflow = 1; break; }   // (cannot happen: see above)
            if (lane < 6) {
                const int d = lane >> 1, hi = lane & 1;
                ws->bb[sp][lane] = (d == cutfeat && hi == 0) ? cutval : (hi ? bhi(d) : blo(d));
            }
            if (lane == 0) { ws->l[sp] = l + idx; ws->r[sp] = r; ws->id[sp] = c + 1; }
            ++sp;
        }
        if (!left_leaf) {          // left child next: box = this box with high[cutfeat] = cutval
            if (cutfeat == 0) b0h = cutval; else if (cutfeat == 1) b1h = cutval; else b2h = cutval;
            r = l + idx;
            id = c;
            continue;
        }
        if (!right_leaf) {         // right child next: box = this box with low[cutfeat] = cutval
            if (cutfeat == 0) b0l = cutval; else if (cutfeat == 1) b1l = cutval; else b2l = cutval;
            l = l + idx;
            id = c + 1;
            continue;
        }
#undef blo
#undef bhi
        if (sp == 0) break;
        --sp;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        l = ws->l[sp]; r = ws->r[sp]; id = ws->id[sp];
        b0l = ws->bb[sp][0]; b0h = ws->bb[sp][1]; b1l = ws->bb[sp][2]; b1h = ws->bb[sp][3]; b2l = ws->bb[sp][4]; b2h = ws->bb[sp][5];
    }
    if (mine) {   // the permuted window back into vAcc_ and the coordinate planes
        T.vind[L0 + lane] = vi;
        T.plane(0)[L0 + lane] = px; T.plane(1)[L0 + lane] = py; T.plane(2)[L0 + lane] = pz;
    }
}


// divideTree for node `root` of 65 .. kExactWindow points AND everything below it, by one wavefront on a COPY of the node's
// window of vAcc_ and of the coordinate planes in LDS.  Measured before this existed (round 4): a wavefront-level split in
// global memory is a chain of ~10 dependent round trips (range and box of the node, min / max, statistics, the two lists,
// the swap, the children's records) = 13 us per node whatever its size, and 56 of the 63 nodes a wavefront splits hold fewer
// than 512 points.  On the copy the same code runs (exact_process_node / exact_subtree_wave on a view of the tree whose
// data pointers address the window: flat loads that land in LDS), the node records still go to global memory (stores only),
// and the window is written back once.
struct ExactWinLds {   // per wavefront
    float x[kExactWindow], y[kExactWindow], z[kExactWindow];
    unsigned vind[kExactWindow];
    unsigned sa[kExactWindow / 2], sb[kExactWindow / 2];   // as many misplaced on the left as on the right: <= half the node each
    int stack[kExactWinStack];
};
__device__ __forceinline__ void exact_window_wave(const ExactTree &T, int root, int *n_nodes_lds, int *overflow, ExactSubLds *ws,
                                                  ExactWinLds *win) {
    const int lane = threadIdx.x & 63;
    const unsigned L0 = T.left[root];
    const int W = (int)(T.right[root] - L0);
    for (int i = lane; i < W; i += 64) {
        win->x[i] = T.plane(0)[L0 + i]; win->y[i] = T.plane(1)[L0 + i]; win->z[i] = T.plane(2)[L0 + i];
        win->vind[i] = T.vind[L0 + i];
    }
    ExactTree TL = T;   // same node arrays; the data of positions [L0, L0 + W) live in the window
    TL.pc = (float *)((uintptr_t)(float *)win->x - (uintptr_t)L0 * sizeof(float));
    TL.pstride = kExactWindow;
    TL.vind = (unsigned *)((uintptr_t)(unsigned *)win->vind - (uintptr_t)L0 * sizeof(unsigned));
    TL.sa = win->sa; TL.sb = win->sb;
    TL.lists_local = 1;
    const WaveCoop gw{};
    int sp = 0;
    if (lane == 0) win->stack[0] = root;
    sp = 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    while (sp > 0) {
        --sp;
        const int id = __builtin_amdgcn_readfirstlane(win->stack[sp]);
        const unsigned cnt = T.right[id] - T.left[id];
        if (cnt <= (unsigned)kExactSubtree) {
            exact_subtree_wave(TL, id, n_nodes_lds, overflow, ws);
        } else {
            const int c = exact_process_node(gw, TL, id, n_nodes_lds, overflow);
            __threadfence_block();   // the children's ranges and boxes are read back from memory by this wavefront
            if (c >= 0) {
                if (sp + 2 > kExactWinStack) { if (lane == 0) *overflow = 1; break; }   // (cannot happen: each pending node holds > 64 points)
                if (lane == 0) { win->stack[sp] = c + 1; win->stack[sp + 1] = c; }
                sp += 2;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    for (int i = lane; i < W; i += 64) {   // the permuted window back into vAcc_ and the coordinate planes
        T.vind[L0 + i] = win->vind[i];
        T.plane(0)[L0 + i] = win->x[i]; T.plane(1)[L0 + i] = win->y[i]; T.plane(2)[L0 + i] = win->z[i];
    }
}

// ---- round 4: the two ends of the tree that cost the build its 5 ms at 50k points ------------------------------------------
// (1) Nodes of more than kExactBigNode points (the top 4 levels of a 50k-point tree) are split by the whole workgroup.  Rounds
// 2-3 ran planeSplit's compactions CHUNK by chunk (1024 elements, two barriers per chunk and direction: ~200 barrier pairs per
// level).  Here every wavefront owns one contiguous SEGMENT of the node, walks it with ballots (no barrier), keeps its
// misplaced positions in its own part of the lists, and the sixteen counts meet in LDS once: a partition is three barriers.
// The passes are fused as well: min / max of all three dimensions in one pass, and ONE statistics pass on the cut dimension
// (#{v < cut}, #{v == cut}, max{v < cut}, min{v > cut}) gives both partition sizes and divlow / divhigh -- the left child is
// [all v < cut | idx - lim1 elements equal to cut], so its maximum is cutval when idx > lim1 and max{v < cut} otherwise; the
// right child mirrors it (nanoflann_two.hpp:1086,1096-1102 take them from the children's recomputed boxes).
struct ExactBigLds {
    double red[kExactTopThreads / 64][6];
    unsigned cnt[kExactTopThreads / 64][2];
    unsigned pre[2][kExactTopThreads / 64 + 1];
};
template <bool STRICT>
__device__ __forceinline__ void block_partition(const ExactTree &T, unsigned lo, unsigned hi, int dim, double cutval, unsigned cnt,
                                                ExactBigLds *B) {
    constexpr int NW = kExactTopThreads / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned n = hi - lo, lim = lo + cnt;
    const unsigned seg = ((n + NW - 1) / NW + 63) & ~63u;   // whole chunks of 64 per wavefront
    const unsigned s_lo = min(hi, lo + w * seg), s_hi = min(hi, s_lo + seg);
    // misplaced positions of this wavefront's segment, both ASCENDING, at the start of its own part of sa / sb
    unsigned ml = 0, mr = 0;
    for (unsigned base = s_lo; base < s_hi; base += 256) {   // four chunks per round: their loads are in flight together
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = base + 64 * u + lane;
            v[u] = i < s_hi ? T.plane(dim)[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = base + 64 * u + lane;
            const bool in = i < s_hi;
            const bool pr = in && (STRICT ? (double)v[u] < cutval : (double)v[u] <= cutval);
            const bool fl = in && i < lim && !pr, fr = in && i >= lim && pr;
            const unsigned long long bl = __ballot(fl), br = __ballot(fr);
            if (fl) T.sa[s_lo + ml + mask_rank_below(bl)] = i;
            if (fr) T.sb[s_lo + mr + mask_rank_below(br)] = i;
            ml += __popcll(bl); mr += __popcll(br);
        }
    }
    if (lane == 0) { B->cnt[w][0] = ml; B->cnt[w][1] = mr; }
    __threadfence_block();
    __syncthreads();
    if (tid < 2) {
        unsigned acc = 0;
        for (int j = 0; j < NW; ++j) { B->pre[tid][j] = acc; acc += B->cnt[j][tid]; }
        B->pre[tid][NW] = acc;
    }
    __syncthreads();
    const unsigned total = B->pre[0][NW];   // == B->pre[1][NW]: as many misplaced on the left as on the right
    // planeSplit's swap loop pairs the j-th misplaced from the left (ascending) with the j-th from the right (DESCENDING)
#pragma unroll 2
    for (unsigned j = tid; j < total; j += kExactTopThreads) {
        int wl = 0, wr = 0;
        const unsigned jr = total - 1 - j;
#pragma unroll
        for (int st = 8; st > 0; st >>= 1) {   // the last wavefront whose prefix is <= j (prefixes ascend): 4 probes (NW <= 16)
            if (wl + st < NW && B->pre[0][wl + st] <= j) wl += st;
            if (wr + st < NW && B->pre[1][wr + st] <= jr) wr += st;
        }
        const unsigned a = T.sa[min(hi, lo + wl * seg) + (j - B->pre[0][wl])];
        const unsigned b = T.sb[min(hi, lo + wr * seg) + (jr - B->pre[1][wr])];
        const unsigned ta = T.vind[a], tb = T.vind[b];
        T.vind[a] = tb; T.vind[b] = ta;
        float *p0 = T.plane(0), *p1 = T.plane(1), *p2 = T.plane(2);
        const float xa = p0[a], xb = p0[b], ya = p1[a], yb = p1[b], za = p2[a], zb = p2[b];
        p0[a] = xb; p0[b] = xa; p1[a] = yb; p1[b] = ya; p2[a] = zb; p2[b] = za;
    }
    __threadfence_block();
    __syncthreads();
}

__device__ __forceinline__ void exact_process_big_node(const ExactTree &T, int id, int *n_nodes_lds, int *overflow, ExactBigLds *B) {
    constexpr int NW = kExactTopThreads / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned l = T.left[id], r = T.right[id], count = r - l;
    const double b0l = T.nbbox[(size_t)id * 6 + 0], b0h = T.nbbox[(size_t)id * 6 + 1], b1l = T.nbbox[(size_t)id * 6 + 2],
                 b1h = T.nbbox[(size_t)id * 6 + 3], b2l = T.nbbox[(size_t)id * 6 + 4], b2h = T.nbbox[(size_t)id * 6 + 5];
#define blo(d) ((d) == 0 ? b0l : ((d) == 1 ? b1l : b2l))
#define bhi(d) ((d) == 0 ? b0h : ((d) == 1 ? b1h : b2h))
    // pass 1: computeMinMax (:1037-1052) of the dimensions whose box span qualifies (:1201-1225; usually one), in one pass
    const double EPS = 0.00001;
    double max_span = b0h - b0l;
#pragma unroll
    for (int d = 1; d < 3; ++d) {
        const double span = bhi(d) - blo(d);
        if (span > max_span) max_span = span;
    }
    const bool q0 = (b0h - b0l) > (1 - EPS) * max_span, q1 = (b1h - b1l) > (1 - EPS) * max_span, q2 = (b2h - b2l) > (1 - EPS) * max_span;
    double mn0 = DBL_MAX, mn1 = DBL_MAX, mn2 = DBL_MAX, mx0 = -DBL_MAX, mx1 = -DBL_MAX, mx2 = -DBL_MAX;
#pragma unroll 8
    for (unsigned i = l + tid; i < r; i += kExactTopThreads) {
        if (q0) { const double x = T.val(i, 0); mn0 = fmin(mn0, x); mx0 = fmax(mx0, x); }
        if (q1) { const double y = T.val(i, 1); mn1 = fmin(mn1, y); mx1 = fmax(mx1, y); }
        if (q2) { const double z = T.val(i, 2); mn2 = fmin(mn2, z); mx2 = fmax(mx2, z); }
    }
    mn0 = wave_min_f64(mn0); mn1 = wave_min_f64(mn1); mn2 = wave_min_f64(mn2);
    mx0 = wave_max_f64(mx0); mx1 = wave_max_f64(mx1); mx2 = wave_max_f64(mx2);
    if (lane == 0) { B->red[w][0] = mn0; B->red[w][1] = mx0; B->red[w][2] = mn1; B->red[w][3] = mx1; B->red[w][4] = mn2; B->red[w][5] = mx2; }
    __syncthreads();
    mn0 = B->red[0][0]; mx0 = B->red[0][1]; mn1 = B->red[0][2]; mx1 = B->red[0][3]; mn2 = B->red[0][4]; mx2 = B->red[0][5];
#pragma unroll
    for (int j = 1; j < NW; ++j) {
        mn0 = fmin(mn0, B->red[j][0]); mx0 = fmax(mx0, B->red[j][1]); mn1 = fmin(mn1, B->red[j][2]); mx1 = fmax(mx1, B->red[j][3]);
        mn2 = fmin(mn2, B->red[j][4]); mx2 = fmax(mx2, B->red[j][5]);
    }
    __syncthreads();   // red is reused below
    // middleSplit_ (:1197-1245)
    double max_spread = -1.0, min_elem = 0.0, max_elem = 0.0;
    int cutfeat = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double span = bhi(d) - blo(d);
        if (span > (1 - EPS) * max_span) {
            const double mn = d == 0 ? mn0 : (d == 1 ? mn1 : mn2), mx = d == 0 ? mx0 : (d == 1 ? mx1 : mx2);
            const double spread = mx - mn;
            if (spread > max_spread) { cutfeat = d; max_spread = spread; min_elem = mn; max_elem = mx; }
        }
    }
    const double split_val = (blo(cutfeat) + bhi(cutfeat)) / 2;
    const double cutval = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
    // pass 2: the statistics of the cut
    unsigned na = 0, nb = 0;
    double maxa = -DBL_MAX, minc = DBL_MAX;
#pragma unroll 8
    for (unsigned i = l + tid; i < r; i += kExactTopThreads) {
        const double v = T.val(i, cutfeat);
        na += v < cutval ? 1u : 0u; nb += v == cutval ? 1u : 0u;
        maxa = v < cutval ? fmax(maxa, v) : maxa;
        minc = v > cutval ? fmin(minc, v) : minc;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { na += __shfl_xor(na, off); nb += __shfl_xor(nb, off); }
    maxa = wave_max_f64(maxa); minc = wave_min_f64(minc);
    if (lane == 0) { B->cnt[w][0] = na; B->cnt[w][1] = nb; B->red[w][0] = maxa; B->red[w][1] = minc; }
    __syncthreads();
    na = 0; nb = 0; maxa = -DBL_MAX; minc = DBL_MAX;
#pragma unroll
    for (int j = 0; j < NW; ++j) { na += B->cnt[j][0]; nb += B->cnt[j][1]; maxa = fmax(maxa, B->red[j][0]); minc = fmin(minc, B->red[j][1]); }
    __syncthreads();
    // planeSplit (:1256-1294): the two Hoare partitions
    const unsigned lim1 = na, lim2 = na + nb;
    if (na > 0) block_partition<true>(T, l, r, cutfeat, cutval, na, B);
    if (nb > 0) block_partition<false>(T, l + lim1, r, cutfeat, cutval, nb, B);   // (block-uniform: na, nb come from LDS)
    const unsigned half = count / 2;
    const unsigned idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
    const double dlo = idx > lim1 ? cutval : maxa;     // divlow  = the left child's high on the cut dimension
    const double dhi = idx < lim2 ? cutval : minc;     // divhigh = the right child's low
    int c = 0;
    if (tid == 0) {
        c = atomicAdd(n_nodes_lds, 2);
        if (c + 2 > T.max_nodes) { *overflow = 1; c = -1; }
        B->cnt[0][0] = (unsigned)c;
    }
    __syncthreads();
    c = (int)B->cnt[0][0];
    __syncthreads();
    if (c < 0) return;
    if (tid == 0) {
        T.feat[id] = cutfeat; T.child[id] = c; T.low[id] = dlo; T.high[id] = dhi;
        T.left[c] = l; T.right[c] = l + idx; T.left[c + 1] = l + idx; T.right[c + 1] = r;
        T.feat[c] = kExactTodo; T.feat[c + 1] = kExactTodo;
    }
    if (tid < 6) {
        const int d = tid >> 1, hi = tid & 1;
        double vl = hi ? bhi(d) : blo(d), vr = vl;
        if (d == cutfeat && hi == 1) vl = cutval;  // left_bbox[cutfeat].high = cutval
        if (d == cutfeat && hi == 0) vr = cutval;  // right_bbox[cutfeat].low = cutval
        T.nbbox[(size_t)c * 6 + tid] = vl;
        T.nbbox[(size_t)(c + 1) * 6 + tid] = vr;
    }
#undef blo
#undef bhi
}

// buildIndex for scene s in two kernels (each called by every thread of a kExactThreads block); n = cloud.pts.size().
// exact_build_top: identity vAcc_ + planes, computeBoundingBox, and every node of more than kExactBigNode points, level by level,
// by the whole workgroup; leaves the node count in *T.n_nodes (-1: capacity exceeded) and the remaining nodes marked kExactTodo.
// exact_build_rest: everything below, one wavefront per node (<= 64 points: the whole subtree in registers).  Two kernels because the two halves
// want different register files: together (round 3, and the first version of this round) the uniform state of both -- two
// views of the tree, the LDS blocks -- spilled SGPRs through VGPR lanes into scratch at the 128-VGPR budget of a 1024-thread block.
__device__ __forceinline__ void exact_build_top(const ExactTree T, int n) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nw = kExactTopThreads / 64;
    __shared__ int n_nodes_lds, overflow, head, tail, any_big;
    __shared__ double red[kExactTopThreads / 64][6];
    __shared__ ExactBigLds big;
    // init_vind + the coordinate planes in the same (identity) order, and computeBoundingBox (:1694-1720) on the way
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
#pragma unroll 4
    for (int i = tid; i < n; i += kExactTopThreads) {
        const float xf = T.x[i], yf = T.y[i], zf = T.z[i];
        T.vind[i] = i;
        T.plane(0)[i] = xf; T.plane(1)[i] = yf; T.plane(2)[i] = zf;
        const double v[3] = {(double)xf, (double)yf, (double)zf};
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
    if (tid == 0 && n > 0) { T.left[0] = 0; T.right[0] = (unsigned)n; T.feat[0] = kExactTodo; }
    __threadfence_block();
    __syncthreads();
    while (head < tail) {  // one level of divideTree per round, big nodes only (children are appended behind `tail`)
        const int h = head, t = tail;
        if (tid == 0) any_big = 0;
        __syncthreads();
        for (int id = h; id < t; ++id)
            if (T.right[id] - T.left[id] > (unsigned)kExactBigNode) {
                exact_process_big_node(T, id, &n_nodes_lds, &overflow, &big);
                if (tid == 0) any_big = 1;
            }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) { head = t; tail = (overflow || !any_big) ? t : n_nodes_lds; }
        __syncthreads();
    }
    if (tid == 0) *T.n_nodes = overflow ? -1 : n_nodes_lds;
}

constexpr int kExactMaxIdleSpins = 200000;
constexpr int kExactQueue = 2048;   // ring of open nodes (only nodes of more than kExactWindow points put their children here: the frontier of a 200 k-point tree holds < 1 k)
// qcap: capacity of the ring actually used (a power of two <= kExactQueue; smaller only in tests: amk__exact_set_queue_cap)
__device__ __forceinline__ void exact_build_rest(const ExactTree T, int n, int qcap = kExactQueue) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ int n_nodes_lds, overflow, q_head, q_tail, pending;
    __shared__ int queue[kExactQueue];
    __shared__ ExactSubLds sub[kExactThreads / 64];
    __shared__ ExactWinLds win[kExactThreads / 64];
    const WaveCoop gw{};
    if (tid == 0) {
        const int nn = *T.n_nodes;
        n_nodes_lds = nn < 0 ? 0 : nn; overflow = nn < 0 ? 1 : 0; q_head = 0; q_tail = 0; pending = 0;
    }
    for (int i = tid; i < kExactQueue; i += kExactThreads) queue[i] = -1;
    __syncthreads();
    // A window keeps one wavefront busy for ~100 us: small clouds (the reference's own 3072-point frames) take smaller windows,
    // or most of the sixteen wavefronts would have nothing to do (3072 points, 256 scenes: 0.28 ms with 416-point windows,
    // 0.19 with 96; 50 k points: 2.55 against 2.70)
    const unsigned win_thr = (unsigned)max(96, min(kExactWindow, n / 64));
    // The nodes the top kernel left.  No level loop below them (rounds 2-3 and the first half of round 4 had one: the barrier
    // per level cost 29 % of this kernel in waiting for the level's slowest wavefront, and every wavefront re-read feat[] of
    // the nodes other wavefronts had already built, 6 %): a ring of open nodes in LDS.  A wavefront pops a node, splits it (or
    // builds its whole subtree when it holds <= kExactSubtree points) and pushes the two children; `pending` counts the nodes
    // pushed and not yet finished, the kernel ends when it reaches zero.  Slot i of the ring holds -1 while it is empty or
    // reserved; a push that would lap the consumers marks the scene's tree unavailable (the bucketed index answers).
    const int nn0 = n_nodes_lds;
    for (int id = tid; id < nn0; id += kExactThreads)
        if (T.feat[id] == kExactTodo) {
            const int pos = atomicAdd(&q_tail, 1);
            if (pos < qcap) { queue[pos] = id; atomicAdd(&pending, 1); }
            else overflow = 1;
        }
    __syncthreads();
    if (overflow) {   // (more open nodes than the ring holds, or the top kernel ran out of node capacity)
        if (tid == 0) *T.n_nodes = -1;
        return;
    }
    int spins = 0;
#ifdef AMK_EXACT_TRACE
    unsigned long long t_sub = 0, t_win = 0, t_mid = 0, n_sub = 0, n_win = 0, n_mid = 0;
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_last_work = t_begin;
#endif
    for (;;) {
        int id = -1, idle = 0;
        if (++spins > kExactMaxIdleSpins) {   // bounded waiting: ~0.1 s of idle polling means something is wrong (it never
            overflow = 1;                     // happened in any test): give up, the scene keeps the bucketed index's answer.
            break;                            // (Without this bound an earlier build of this loop hung on the hardware; with it
        }                                     // the loop provably ends, and the compiler cannot treat it as endless either.)
        if (lane == 0) {
            for (;;) {
                const int hd = __atomic_load_n(&q_head, __ATOMIC_RELAXED);
                if (hd >= __atomic_load_n(&q_tail, __ATOMIC_RELAXED)) break;
                if (atomicCAS(&q_head, hd, hd + 1) == hd) {
                    const int slot = hd & (qcap - 1);
                    // (reserved, being written: the producer advanced q_tail before it stored the id.  It stores it unless the
                    // scene was given up meanwhile -- then nobody will, and this wave leaves with the others)
                    do { id = __atomic_load_n(&queue[slot], __ATOMIC_RELAXED); } while (id < 0 && !__atomic_load_n(&overflow, __ATOMIC_RELAXED));
                    if (id >= 0) __atomic_store_n(&queue[slot], -1, __ATOMIC_RELAXED);
                    break;
                }
            }
            if (id < 0) idle = __atomic_load_n(&pending, __ATOMIC_RELAXED) == 0 || __atomic_load_n(&overflow, __ATOMIC_RELAXED);
        }
        id = __builtin_amdgcn_readfirstlane(id);
        idle = __builtin_amdgcn_readfirstlane(idle);
        if (id < 0) {
            if (idle) break;
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        spins = 0;
        const unsigned cnt = T.right[id] - T.left[id];
#ifdef AMK_EXACT_TRACE
        const unsigned long long t0 = wall_clock64();
#endif
        if (cnt <= (unsigned)kExactSubtree) exact_subtree_wave(T, id, &n_nodes_lds, &overflow, &sub[w]);
        else if (cnt <= win_thr) exact_window_wave(T, id, &n_nodes_lds, &overflow, &sub[w], &win[w]);
        else {
            const int c = exact_process_node(gw, T, id, &n_nodes_lds, &overflow);
            __threadfence_block();   // the children's range / box and the permuted window are in memory before anyone can pop them
            if (lane == 0 && c >= 0) {   // split: the two children are open nodes
                // Reserve two slots -- capacity is checked BEFORE q_tail moves (advancing first and then giving up left
                // two positions inside [q_head, q_tail) that nobody would ever write, and the consumer that claimed one spun on it
                // for good; reachable beyond ~430 k points per scene).  q_head only grows, so a stale read errs on the safe side.
                int pos = __atomic_load_n(&q_tail, __ATOMIC_RELAXED);
                bool room;
                for (;;) {
                    room = pos + 2 - __atomic_load_n(&q_head, __ATOMIC_RELAXED) <= qcap;
                    if (!room) break;
                    const int seen = atomicCAS(&q_tail, pos, pos + 2);
                    if (seen == pos) break;
                    pos = seen;
                }
                if (room) {
                    atomicAdd(&pending, 2);
                    for (int e = 0; e < 2; ++e) {
                        int *slot = &queue[(pos + e) & (qcap - 1)];
                        // (a lapped slot whose consumer has claimed it but not read it yet)
                        while (__atomic_load_n(slot, __ATOMIC_RELAXED) != -1 && !__atomic_load_n(&overflow, __ATOMIC_RELAXED)) {}
                        __atomic_store_n(slot, c + e, __ATOMIC_RELAXED);
                    }
                } else {
                    __atomic_store_n(&overflow, 1, __ATOMIC_RELAXED);   // the scene keeps the bucketed index's answer
                }
            }
        }
        if (lane == 0) atomicSub(&pending, 1);
#ifdef AMK_EXACT_TRACE
        {
            const unsigned long long t1 = wall_clock64();
            if (cnt <= (unsigned)kExactSubtree) { t_sub += t1 - t0; ++n_sub; }
            else if (cnt <= win_thr) { t_win += t1 - t0; ++n_win; }
            else { t_mid += t1 - t0; ++n_mid; }
            t_last_work = t1;
        }
#endif
    }
#ifdef AMK_EXACT_TRACE
    {
        const unsigned long long t_end = wall_clock64();
        __syncthreads();
        const unsigned long long t_all = wall_clock64();
        if (blockIdx.x == 0 && lane == 0) {   // 100 MHz ticks
            unsigned *o = T.sa + w * 16;
            o[0] = (unsigned)t_sub; o[1] = (unsigned)n_sub; o[2] = (unsigned)t_win; o[3] = (unsigned)n_win; o[4] = (unsigned)t_mid; o[5] = (unsigned)n_mid;
            o[6] = (unsigned)(t_end - t_begin); o[7] = (unsigned)(t_last_work - t_begin); o[8] = (unsigned)(t_all - t_begin);
        }
    }
#endif
    __syncthreads();
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

// findNeighbors for one query by ONE WAVEFRONT (every lane holds the same query): the same traversal, node data read once
// per wave (uniform addresses), the <= 10 points of a leaf scored by lanes 0..9 from the vAcc_-ordered planes (one
// contiguous run) and then offered to the result set ONE AT A TIME in leaf order, exactly as the sequential loop does
// (:1733-1751) -- the order decides which of several equidistant points is kept.  The KNNResultSet lives in lanes
// 0..k-1 (rd, ri sorted ascending); KNNResultSet::addPoint (:219-246) = count the entries <= dist (they form a prefix),
// shift the rest up by one lane.  `st`: this wave's stack in LDS.  Returns the number of results, -1 = tree unavailable.
struct ExactWaveStack {
    int node[kExactMaxDepth], ik[kExactMaxDepth];
    double a[kExactMaxDepth], b[kExactMaxDepth];
};
__device__ __forceinline__ int exact_knn_wave(const ExactTree &T, double qx, double qy, double qz, int k, double &rd, int &ri,
                                              ExactWaveStack *st) {
    const int lane = threadIdx.x & 63;
    const int nn = *T.n_nodes;
    rd = DBL_MAX;   // KNNResultSet::init (:196-202): dists[capacity - 1] = max; the other slots are never read before written
    ri = -1;
    if (nn <= 0) return nn < 0 ? -1 : 0;
    int count = 0;
    double dists[3] = {0.0, 0.0, 0.0};
    double mind = 0.0;
    const double q[3] = {qx, qy, qz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {  // computeInitialDistances (:1296-1315)
        const double blo = T.root_bbox[2 * d], bhi = T.root_bbox[2 * d + 1];
        if (q[d] < blo) { dists[d] = (q[d] - blo) * (q[d] - blo); mind += dists[d]; }
        if (q[d] > bhi) { dists[d] = (q[d] - bhi) * (q[d] - bhi); mind += dists[d]; }
    }
    auto set_dist = [&](int idx, double v) {
        dists[0] = idx == 0 ? v : dists[0];
        dists[1] = idx == 1 ? v : dists[1];
        dists[2] = idx == 2 ? v : dists[2];
    };
    int sp = 0, node = 0;
    for (;;) {
        for (int idx = T.feat[node]; idx >= 0; idx = T.feat[node]) {
            const double val = idx == 0 ? qx : (idx == 1 ? qy : qz);
            const double diff1 = val - T.low[node], diff2 = val - T.high[node];
            int best, other;
            double cut;
            if ((diff1 + diff2) < 0) { best = T.child[node]; other = best + 1; cut = diff2 * diff2; }
            else { other = T.child[node]; best = other + 1; cut = diff1 * diff1; }
            if (sp >= kExactMaxDepth) return -1;
            if (lane == 0) { st->node[sp] = other; st->ik[sp] = idx; st->a[sp] = cut; st->b[sp] = mind; }
            ++sp;
            node = best;
        }
        {   // leaf: worst distance cached at entry
            const double worst = readlane_f64(rd, k - 1);
            const unsigned lf = T.left[node], rt = T.right[node], cnt = rt - lf;
            double dist = DBL_MAX;
            int pidx = -1;
            if ((unsigned)lane < cnt) {
                const unsigned i = lf + lane;
                pidx = (int)T.vind[i];
                dist = sq_dist(qx, qy, qz, T.plane(0)[i], T.plane(1)[i], T.plane(2)[i]);
            }
            for (unsigned e = 0; e < cnt; ++e) {
                const double de = readlane_f64(dist, (int)e);
                if (!(de < worst)) continue;                       // (NaN never enters, as in the reference)
                const int ae = __builtin_amdgcn_readlane(pidx, (int)e);
                const int pos = __popcll(__ballot(lane < count && rd <= de));   // after the existing equals
                const double up = __shfl_up(rd, 1);
                const int upi = __shfl_up(ri, 1);
                if (lane > pos && lane < k) { rd = up; ri = upi; }
                if (lane == pos && pos < k) { rd = de; ri = ae; }
                if (count < k) ++count;
            }
        }
        __threadfence_block();   // lane 0's stack writes are read by the whole wave
        bool go = false;
        while (sp > 0) {
            --sp;
            const int ik = st->ik[sp], idx = ik & 3;
            if (ik >> 2) { set_dist(idx, st->a[sp]); continue; }
            const double cut = st->a[sp], dst = idx == 0 ? dists[0] : (idx == 1 ? dists[1] : dists[2]);
            const double m2 = st->b[sp] + cut - dst;
            if (m2 * 1.0f <= readlane_f64(rd, k - 1)) {
                set_dist(idx, cut);
                node = st->node[sp];
                __threadfence_block();
                if (lane == 0) { st->ik[sp] = idx | 4; st->a[sp] = dst; }
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
    std::memset(&ep, 0, sizeof ep);   // padding bytes too: step_frames.hip compares tables of these bytewise
    ep.x = kd->x.p; ep.y = kd->y.p; ep.z = kd->z.p; ep.cap = kd->cap;
    ep.vind = kd->ex_vind.p; ep.left = kd->ex_left.p; ep.right = kd->ex_right.p; ep.sa = kd->ex_sa.p; ep.sb = kd->ex_sb.p;
    ep.pc = kd->ex_pc.p; ep.pstride = (size_t)kd->n_scenes * kd->cap;
    ep.feat = kd->ex_feat.p; ep.child = kd->ex_child.p; ep.n_nodes = kd->ex_nn.p;
    ep.low = kd->ex_low.p; ep.high = kd->ex_high.p; ep.nbbox = kd->ex_nbbox.p; ep.root_bbox = kd->ex_root.p;
    ep.max_nodes = kd->ex_max_nodes;
    return ep;
}
