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

// Scans the whole cloud of one scene for QPW queries.  On return lane i < k holds the i-th best
// (distance, index) of query qq in (ld[qq], li[qq]); empty slots hold (DBL_MAX, kNoIndex).
// Ties order by index (DESIGN.md "tie policy").
template <int QPW>
__device__ __forceinline__ void scan_cloud(const float *__restrict__ xs, const float *__restrict__ ys,
                                           const float *__restrict__ zs, int size, const double (&qx)[QPW],
                                           const double (&qy)[QPW], const double (&qz)[QPW], int k, double (&ld)[QPW],
                                           int (&li)[QPW]) {
    const int lane = threadIdx.x & 63;
    double tau[QPW];
#pragma unroll
    for (int qq = 0; qq < QPW; ++qq) {
        tau[qq] = DBL_MAX;  // KNNResultSet::init, nanoflann_two.hpp:196-202
        ld[qq] = DBL_MAX;
        li[qq] = kNoIndex;
    }
    // software pipeline: the loads of tile t+1 are in flight while tile t is evaluated (one wave per
    // SIMD pair otherwise sits on ~2 us of L2/HBM latency per 256-point tile)
    float4 nx4 = *reinterpret_cast<const float4 *>(xs + 4 * lane);
    float4 ny4 = *reinterpret_cast<const float4 *>(ys + 4 * lane);
    float4 nz4 = *reinterpret_cast<const float4 *>(zs + 4 * lane);
    for (int base = 0; base < size; base += 4 * kWave) {
        const float4 x4 = nx4, y4 = ny4, z4 = nz4;
        {   // the planes carry >= 256 floats of NaN padding past `size`, so this never leaves the scene's slice
            const int i1 = base + 4 * kWave + 4 * lane;
            nx4 = *reinterpret_cast<const float4 *>(xs + i1);
            ny4 = *reinterpret_cast<const float4 *>(ys + i1);
            nz4 = *reinterpret_cast<const float4 *>(zs + i1);
        }
        const float px[4] = {x4.x, x4.y, x4.z, x4.w};
        const float py[4] = {y4.x, y4.y, y4.z, y4.w};
        const float pz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int qq = 0; qq < QPW; ++qq) {
                const double d = sq_dist(qx[qq], qy[qq], qz[qq], px[e], py[e], pz[e]);
                // NaN padding / NaN coordinates compare false, as in the reference's dist < worst
                unsigned long long m = __ballot(d <= tau[qq]);
                while (m) {  // rare: a lane beats (or ties) the current k-th best
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const double dc = readlane_f64(d, src);
                    const int ic = base + 4 * src + e;
                    // rank of the candidate in (distance, index) order among the kept entries
                    const bool lt = (lane < k) && (ld[qq] < dc || (ld[qq] == dc && li[qq] < ic));
                    const int pos = __popcll(__ballot(lt));
                    if (pos < k && dc < DBL_MAX) {
                        const double up_d = shfl_up1_f64(ld[qq]);
                        const int up_i = __shfl_up(li[qq], 1);
                        if (lane > pos) {
                            ld[qq] = up_d;
                            li[qq] = up_i;
                        } else if (lane == pos) {
                            ld[qq] = dc;
                            li[qq] = ic;
                        }
                        tau[qq] = readlane_f64(ld[qq], k - 1);
                    }
                }
            }
        }
    }
}

}  // namespace amk
