// Streaming exact kNN over the scan-friendly SoA cloud of an amk_kd (gfx950).  Shared by the C-ABI
// search (kd_index.hip) and the fused control step (step.hip).
//
// One wavefront answers QPW queries of one scene: coalesced 16-byte loads of x/y/z, fp64 squared
// distance with the reference's operation order (AM/include/kd_tree_two.h:24-27,
// AM/include/nanoflann_two.hpp:590-598), a wave-uniform k-th-best threshold per query in scalar
// registers, a ballot to find the rare lanes that beat it, and a shuffle insertion into the sorted
// top-k list that lives in lanes 0..k-1 (the role of KNNResultSet::addPoint, nanoflann_two.hpp:219-246).
#pragma once
#include "amk_common.h"

namespace amk {

__device__ __forceinline__ double sq_dist(double qx, double qy, double qz, float px, float py, float pz) {
    // r = d0*d0; r += d1*d1; r += d2*d2.  Contraction into FMAs must stay off here (HIP's
    // __dmul_rn/__dadd_rn do contract): the squared distances are part of the bit-exact contract.
#pragma clang fp contract(off)
    const double d0 = qx - (double)px;
    const double d1 = qy - (double)py;
    const double d2 = qz - (double)pz;
    double r = d0 * d0;
    r = r + d1 * d1;
    r = r + d2 * d2;
    return r;
}

__device__ __forceinline__ double shfl_up1_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, 1);
    hi = __shfl_up(hi, 1);
    return __hiloint2double(hi, lo);
}

constexpr int kNoIndex = 0x7fffffff;

// Launch geometry of the scan kernels: queries per wavefront (1 or 5), query groups per scene and
// wavefronts per block (consecutive groups of one scene; <= 8).  Five queries per wave amortise the
// tile loads over 20 point.query tests per lane; N = 10/20/30 reference points give 2/4/6 full waves.
inline void scan_geometry(int n_queries, int &qpw, int &groups, int &wpb) {
    qpw = n_queries == 1 ? 1 : 5;
    groups = (n_queries + qpw - 1) / qpw;
    wpb = groups < 8 ? groups : 8;
}

// Conservative fp32 image of a squared-distance threshold: every point whose exact fp64 squared
// distance is <= tau has an fp32-evaluated squared distance <= filter_threshold(tau, e).
//   e = 2^-22 (|q|_inf + max|p|) bounds sqrt(3) x the per-coordinate error of fl32(fl32(q) - p)
//   (half an ulp for rounding q, half an ulp for the subtraction), so the fp32 difference vector is
//   within e of the true one and sqrt(D) <= sqrt(d) + e; the three fp32 roundings of the sum of
//   squares add < 2^-21 relative; the last factor also covers rounding the threshold itself.
// The square root is taken in fp32 (1 ulp) and widened by 2^-20.
__device__ __forceinline__ float filter_threshold(double tau, double e) {
    if (!(tau < 1e37)) return __builtin_inff();  // list not full yet (DBL_MAX) or out of fp32 range
    const double r = (double)sqrtf((float)tau) * (1.0 + 0x1p-20) + e + 1e-30;
    return (float)(r * r * (1.0 + 0x1p-18));
}

// Per-wavefront LDS scratch of the scan: the sorted top-k lists (lane i < k of row qq holds the i-th
// best (distance, index) of query qq), the current k-th best distance and the query coordinates.
// They live in LDS, not registers, so that the rare insertion path can address them with a run-time
// query index (one copy of that code) and the hot loop stays a few dozen instructions.
template <int QPW>
struct ScanLds {
    double ld[QPW][64];
    int li[QPW][64];
    double tau[QPW];
    double q[QPW][3];
    double eabs[QPW];
};

// dynamic LDS of a scan block: one ScanLds per wavefront + a 3*QPW staging row for its queries
template <int QPW>
inline size_t scan_lds_bytes(int wpb) { return (size_t)wpb * (sizeof(ScanLds<QPW>) + sizeof(double) * 3 * QPW); }

// Scans the whole cloud of one scene for QPW queries (coordinates at q[qq * q_stride + {0,1,2}]).
// On return ws->ld[qq][i] / ws->li[qq][i], i < k, hold the i-th best (distance, index); empty slots
// hold (DBL_MAX, kNoIndex).  Ties order by index (DESIGN.md "tie policy").
//
// Hot loop, per 256-point tile: three prefetched 16-byte loads, then for each of the 4 x QPW
// point.query pairs of a lane an fp32 squared distance (3 sub, 1 mul, 2 fma, packed two points per
// instruction) against the conservative fp32 threshold -- no branch, the ballots are OR-ed into one
// bit mask.  Only when the mask is non-zero (k ln(n/k) ~ 50 times per query over 50k points) the
// marked pairs are re-evaluated in fp64 in the reference's operation order and inserted.  The filter
// only produces false positives, so the results are exactly those of a pure fp64 scan.
template <int QPW>
__device__ __forceinline__ void scan_cloud(const float *__restrict__ xs, const float *__restrict__ ys,
                                           const float *__restrict__ zs, int size, float pmax,
                                           const double *__restrict__ q, int q_stride, int k, ScanLds<QPW> *ws) {
    const int lane = threadIdx.x & 63;
    float tauf[QPW], qxf[QPW], qyf[QPW], qzf[QPW];
#pragma unroll
    for (int qq = 0; qq < QPW; ++qq) {
        const double x = q[qq * q_stride], y = q[qq * q_stride + 1], z = q[qq * q_stride + 2];
        qxf[qq] = (float)x;
        qyf[qq] = (float)y;
        qzf[qq] = (float)z;
        tauf[qq] = __builtin_inff();
        // every lane stores the same values: a lane-0-only store would let the compiler keep the
        // other lanes' view of these cells in registers forever (no cross-lane ordering in C++)
        ws->q[qq][0] = x; ws->q[qq][1] = y; ws->q[qq][2] = z;
        ws->eabs[qq] = 0x1p-22 * (fmax(fabs(x), fmax(fabs(y), fabs(z))) + (double)pmax);
        ws->tau[qq] = DBL_MAX;       // KNNResultSet::init, nanoflann_two.hpp:196-202
        ws->ld[qq][lane] = DBL_MAX;
        ws->li[qq][lane] = kNoIndex;
    }
    // Bootstrap: while a list is not full every point passes, and inserting the first 256 points one at a
    // time costs more than scanning the other 50k.  Rank the first 64 points (index 4*lane) against
    // each other instead -- 64 wave-uniform broadcasts per query -- and drop the k best into the list.
    if (size > 0) {
        const int idx0 = 4 * lane;
        const float bx = xs[idx0], by = ys[idx0], bz = zs[idx0];
#pragma unroll 1
        for (int qq = 0; qq < QPW; ++qq) {
            const double d = sq_dist(ws->q[qq][0], ws->q[qq][1], ws->q[qq][2], bx, by, bz);
            const bool valid = d < DBL_MAX;  // false for NaN (padding, NaN coordinates) and overflow
            int rank = 0;
#pragma unroll 1
            for (int jn = 0; jn < 64; ++jn) {
                const double dj = readlane_f64(d, jn);
                rank += (dj < d || (dj == d && jn < lane)) ? 1 : 0;  // NaN dj never counts
            }
            if (valid && rank < k) {
                ws->ld[qq][rank] = d;
                ws->li[qq][rank] = idx0;
            }
        }
#pragma unroll
        for (int qq = 0; qq < QPW; ++qq) {
            const double t = ws->ld[qq][k - 1];
            ws->tau[qq] = t;
            tauf[qq] = filter_threshold(t, ws->eabs[qq]);
        }
    }
    // software pipeline, three tiles deep: one tile is ~150-500 cycles of VALU work but an L2/HBM round
    // trip is ~800-2000, and a scene's scan may be the only wave on its SIMD.  The planes carry >= 1024
    // floats of NaN padding past `size`, so the look-ahead never leaves the scene's slice.
    const float4 *xv = reinterpret_cast<const float4 *>(xs) + lane;
    const float4 *yv = reinterpret_cast<const float4 *>(ys) + lane;
    const float4 *zv = reinterpret_cast<const float4 *>(zs) + lane;
    float4 ax = xv[0], ay = yv[0], az = zv[0];
    float4 bx4 = xv[64], by4 = yv[64], bz4 = zv[64];
    float4 cx4 = xv[128], cy4 = yv[128], cz4 = zv[128];
    for (int base = 0; base < size; base += 4 * kWave) {
        const float4 x4 = ax, y4 = ay, z4 = az;
        ax = bx4; ay = by4; az = bz4;
        bx4 = cx4; by4 = cy4; bz4 = cz4;
        {
            const int t3 = (base >> 2) + 192;  // tile + 3, in float4 units
            cx4 = xv[t3]; cy4 = yv[t3]; cz4 = zv[t3];
        }
        const float px[4] = {x4.x, x4.y, x4.z, x4.w};
        const float py[4] = {y4.x, y4.y, y4.z, y4.w};
        const float pz[4] = {z4.x, z4.y, z4.z, z4.w};
        unsigned hit = 0u;  // bit e*QPW+qq: some lane's point e may beat the k-th best of query qq
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int qq = 0; qq < QPW; ++qq) {
                const float fx = qxf[qq] - px[e], fy = qyf[qq] - py[e], fz = qzf[qq] - pz[e];
                const float d32 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                // NaN padding / NaN coordinates compare false here and in the exact test below
                hit |= (__ballot(d32 <= tauf[qq]) != 0ull) ? (1u << (e * QPW + qq)) : 0u;
            }
        }
        if (base == 0) hit &= ~((1u << QPW) - 1u);  // points 4*lane were consumed by the bootstrap
        if (hit == 0u) continue;
        unsigned changed = 0u;
#pragma unroll 1
        while (hit) {
            const int b = __ffs((int)hit) - 1;
            hit &= hit - 1;
            const int e = b / QPW, qq = b % QPW;
            const float x = e == 0 ? px[0] : (e == 1 ? px[1] : (e == 2 ? px[2] : px[3]));
            const float y = e == 0 ? py[0] : (e == 1 ? py[1] : (e == 2 ? py[2] : py[3]));
            const float z = e == 0 ? pz[0] : (e == 1 ? pz[1] : (e == 2 ? pz[2] : pz[3]));
            const double d = sq_dist(ws->q[qq][0], ws->q[qq][1], ws->q[qq][2], x, y, z);
            double tau = ws->tau[qq];
            unsigned long long m = __ballot(d <= tau);
            if (!m) continue;
            double cd = ws->ld[qq][lane];
            int ci = ws->li[qq][lane];
#pragma unroll 1
            while (m) {  // a lane beats (or ties) the current k-th best
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const double dc = readlane_f64(d, src);
                const int ic = base + 4 * src + e;
                // rank of the candidate in (distance, index) order among the kept entries
                const bool lt = (lane < k) && (cd < dc || (cd == dc && ci < ic));
                const int pos = __popcll(__ballot(lt));
                if (pos < k && dc < DBL_MAX) {
                    const double up_d = shfl_up1_f64(cd);
                    const int up_i = __shfl_up(ci, 1);
                    if (lane > pos) {
                        cd = up_d;
                        ci = up_i;
                    } else if (lane == pos) {
                        cd = dc;
                        ci = ic;
                    }
                    tau = readlane_f64(cd, k - 1);
                }
            }
            ws->ld[qq][lane] = cd;
            ws->li[qq][lane] = ci;
            ws->tau[qq] = tau;  // all lanes, same value (see above)
            changed |= 1u << qq;
        }
#pragma unroll
        for (int qq = 0; qq < QPW; ++qq)
            if (changed & (1u << qq)) tauf[qq] = filter_threshold(ws->tau[qq], ws->eabs[qq]);
    }
}

}  // namespace amk
