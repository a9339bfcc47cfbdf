// Bucketed spatial index of an amk_kd scene and the exact kNN search over it (gfx950).
//
// Role in the reference: nanoflann's buildIndex / findNeighbors (AM/include/nanoflann_two.hpp:1518-1541,
// 1563-1586, 1729-1793) behind KDTreeTwo::InitializeNew / SearchForNearest (AM/include/kd_tree_two.h).
// Only the *results* of a search must equal the reference's (exact k nearest, ascending), so the tree
// shape is free.  On a GPU a pointer tree with 10-point leaves is the wrong shape:
// the build here is a counting sort of the points into the cells of a uniform grid (bucket-contiguous
// records, ~64 points per cell, <= 1024 cells), the search one wavefront per query that
// visits the cells in growing Chebyshev rings around the query's cell -- 64 lanes evaluate 64 bucket
// points at a time -- and stops as soon as the k-th best squared distance is below the squared distance
// to everything not yet visited.  That is the same branch-and-bound argument as nanoflann's
// searchLevel (mindist <= worstDist, :1780-1790), applied to rings instead of half-spaces.
#pragma once
#include <cfloat>
#include <type_traits>
#include "kd_device.h"

namespace amk {

#ifndef AMK_GRID_MAX_CELLS
#define AMK_GRID_MAX_CELLS 1024
#endif
constexpr int kGridMaxCells = AMK_GRID_MAX_CELLS;
#ifndef AMK_GRID_PPC
#define AMK_GRID_PPC 64
#endif
// Resolution: ~64 points per cell, <= 1024 cells (round 1: ~8 per cell, <= 8192) -- chosen when the build scattered every
// record over the scene's whole region and kept one open output line per cell (measured with 20 steps in flight,
// tools/experiments/cells_sens.sh: 50k-point clouds 385 -> 402 k steps/s, 200k-point clouds 125 -> 137 k, 5k / 3k-point
// clouds unchanged); with the one-pass tile build below it bounds the per-tile tables (4 KB each) and the LDS histogram.
// The search pays with longer candidate lists (the ball clipping still applies).
constexpr int kGridPointsPerCell = AMK_GRID_PPC;  // target occupancy of a cell
// 512, not 1024: a 1024-thread workgroup needs four wave slots on every SIMD of one CU at the same moment, and the
// dispatcher holds everything behind it until a CU qualifies -- measured: 1024-thread builds do not overlap with
// the solves (or the searches) of other streams AT ALL (time = sum), 512/256-thread builds overlap almost fully
// (solves + builds on separate streams: 32 ms vs 27 ms for the solves alone, 48 ms as a sum).  Alone on the chip
// the 1024-thread version is ~25 % faster (16 instead of 8 waves per CU hide the load latency better).
#ifndef AMK_BUILD_THREADS
#define AMK_BUILD_THREADS 512
#endif
constexpr int kGridBuildThreads = AMK_BUILD_THREADS;
#ifndef AMK_GRID_UNROLL
#define AMK_GRID_UNROLL 16  // points in flight per thread of the two-pass build from SoA planes (keyframe sweep)
#endif
constexpr int kGridUnroll = AMK_GRID_UNROLL;
constexpr int kGridParamDoubles = 8;  // bbmin[3], h, inv_h, gx, gy, gz

// ------------------------------------------------------------------------------------------------
// build: one block per scene over the compacted SoA planes
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool finite3(float x, float y, float z) {
    return fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f;  // false for NaN and inf
}

__device__ __forceinline__ int cell_of(double p, double bbmin, double inv_h, int g) {
    double c = floor((p - bbmin) * inv_h);
    c = c < 0.0 ? 0.0 : c;
    const double gm = (double)(g - 1);
    c = c > gm ? gm : c;  // NaN never gets here (finite3), +-inf query coordinates clamp
    return (int)c;
}

// Grid geometry of a scene: ~kGridPointsPerCell points per cell, <= kGridMaxCells cells, cubic cells of edge h; one thread.
// bbox need not contain every point (amk_kd_build passes the box of a sample): a point outside is clamped into a boundary
// cell, which every rule of the search tolerates (boundary faces are never used as bounds).
__device__ __forceinline__ void grid_geometry(const float *bbox6, int npoints, double *geo, double *gp) {
    double lo[3], hi[3], ext[3], emax = 0.0;
    for (int a = 0; a < 3; ++a) {
        float m0 = bbox6[a], m1 = bbox6[3 + a];
        if (!(m1 >= m0)) { m0 = 0.f; m1 = 0.f; }  // no finite point at all
        lo[a] = m0; hi[a] = m1;
        ext[a] = (double)m1 - (double)m0;
        emax = fmax(emax, ext[a]);
    }
    const double efloor = fmax(emax * 1e-6, 1e-30);
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) vol *= fmax(ext[a], efloor);
    double target = fmin(fmax((double)npoints / (double)kGridPointsPerCell, 1.0), (double)kGridMaxCells);
    double h = cbrt(vol / target);
    if (!(h > 0.0) || !(h < 1e300)) h = 1.0;
    int g[3];
    for (int it = 0; it < 64; ++it) {
        long long prod = 1;
        for (int a = 0; a < 3; ++a) {
            double c = ceil(fmax(ext[a], efloor) / h);
            g[a] = (int)fmin(fmax(c, 1.0), 1024.0);
            prod *= g[a];
        }
        if (prod <= kGridMaxCells) break;
        h *= 1.26;
    }
    if ((long long)g[0] * g[1] * g[2] > kGridMaxCells) { g[0] = g[1] = g[2] = 1; }
    geo[0] = lo[0]; geo[1] = lo[1]; geo[2] = lo[2];
    geo[3] = h; geo[4] = 1.0 / h;
    geo[5] = g[0]; geo[6] = g[1]; geo[7] = g[2];
    for (int a = 0; a < kGridParamDoubles; ++a) gp[a] = geo[a];
}

// The index of scene s from the compacted SoA planes (the keyframe sweep rebuilds from them after its in-place
// compaction; amk_kd_build itself uses grid_build_tiles_scene below): a two-pass counting sort -- histogram, exclusive
// scan, scatter -- that writes ONE tile (tile 0 holds every point, the other tiles of the handle are left empty).  Called
// by every thread of a kGridBuildThreads block.
__device__ __forceinline__ void grid_build_scene(int s, const float *__restrict__ xs, const float *__restrict__ ys,
                                                 const float *__restrict__ zs, int cap, int n, const float *__restrict__ bbox,
                                                 float4 *__restrict__ GP, int *__restrict__ cell_start, int ntiles,
                                                 double *__restrict__ gparams) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float4 *gpt4 = GP + (size_t)s * cap;  // bucket-contiguous points: (x, y, z, cloud index as int bits)
    int *cs = cell_start + (size_t)s * ntiles * (kGridMaxCells + 2);
    double *gp = gparams + (size_t)s * kGridParamDoubles;

    __shared__ int hist[kGridMaxCells + 2];
    __shared__ int wsum[kGridBuildThreads / 64];
    __shared__ double geo[kGridParamDoubles];

    // 1. bounding box of the finite points: reduced by the caller (bbox[s][6] = min xyz, max xyz)
    // 2. grid geometry: ~kGridPointsPerCell points per cell, <= kGridMaxCells cells, cubic cells of edge h
    if (tid == 0) grid_geometry(bbox + 6 * s, n, geo, gp);
    __syncthreads();
    const double b0 = geo[0], b1 = geo[1], b2 = geo[2], inv_h = geo[4];
    const int g0 = (int)geo[5], g1 = (int)geo[6], g2 = (int)geo[7];
    const int ncell = g0 * g1 * g2;  // + one trash bucket (index ncell) for non-finite points
    for (int i = tid; i <= ncell + 1; i += kGridBuildThreads) hist[i] = 0;
    __syncthreads();
    auto cell_id = [&](float x, float y, float z) {
        return finite3(x, y, z)
                   ? (cell_of(z, b2, inv_h, g2) * g1 + cell_of(y, b1, inv_h, g1)) * g0 + cell_of(x, b0, inv_h, g0)
                   : ncell;
    };
    // 3. histogram.  Both point passes load kGridUnroll points per thread before touching them: the loop is bound by
    // load latency, not bandwidth, unless several loads are in flight.
    for (int i0 = tid; i0 < n; i0 += kGridUnroll * kGridBuildThreads) {
        float x[kGridUnroll], y[kGridUnroll], z[kGridUnroll];
#pragma unroll
        for (int j = 0; j < kGridUnroll; ++j) {
            const int i = i0 + j * kGridBuildThreads;
            const int ii = i < n ? i : i0;
            x[j] = xs[ii]; y[j] = ys[ii]; z[j] = zs[ii];
        }
#pragma unroll
        for (int j = 0; j < kGridUnroll; ++j)
            if (i0 + j * kGridBuildThreads < n) atomicAdd(&hist[cell_id(x[j], y[j], z[j])], 1);
    }
    __syncthreads();
    // 4. exclusive scan of hist[0 .. ncell] -> bucket starts (global) and scatter cursors (LDS)
    {
        const int per = (ncell + 1 + kGridBuildThreads - 1) / kGridBuildThreads;
        const int i0 = tid * per;
        int loc = 0;
        for (int j = 0; j < per; ++j)
            if (i0 + j <= ncell) loc += hist[i0 + j];
        int incl = loc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int wbase = 0;
        for (int j = 0; j < w; ++j) wbase += wsum[j];
        int run = wbase + incl - loc;
        for (int j = 0; j < per; ++j)
            if (i0 + j <= ncell) {
                const int c = hist[i0 + j];
                hist[i0 + j] = run;
                cs[i0 + j] = run;
                run += c;
            }
        if (tid == 0) cs[ncell + 1] = n;
    }
    for (int tt = 1; tt < ntiles; ++tt)   // the handle's other tiles: empty runs behind the records
        for (int i = tid; i <= ncell + 1; i += kGridBuildThreads) cs[(size_t)tt * (kGridMaxCells + 2) + i] = n;
    __syncthreads();
    // 5. scatter into bucket-contiguous order (order inside a bucket is irrelevant: results are ordered by
    // (distance, original index))
    for (int i0 = tid; i0 < n; i0 += kGridUnroll * kGridBuildThreads) {
        float x[kGridUnroll], y[kGridUnroll], z[kGridUnroll];
#pragma unroll
        for (int j = 0; j < kGridUnroll; ++j) {
            const int i = i0 + j * kGridBuildThreads;
            const int ii = i < n ? i : i0;
            x[j] = xs[ii]; y[j] = ys[ii]; z[j] = zs[ii];
        }
#pragma unroll
        for (int j = 0; j < kGridUnroll; ++j) {
            const int i = i0 + j * kGridBuildThreads;
            if (i < n) {
                const int pos = atomicAdd(&hist[cell_id(x[j], y[j], z[j])], 1);
                gpt4[pos] = make_float4(x[j], y[j], z[j], __int_as_float(i));  // one 16-byte store per point
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// one-pass build: tile-major records
// ------------------------------------------------------------------------------------------------
// The two-pass counting sort above reads the cloud twice and scatters every record over the scene's whole 800 KB region
// (measured: 1.64 x the algorithmic bytes, 0.29 of the HBM peak).  Here the cloud is cut into TILES of kTilePoints
// consecutive points (4096); a tile is loaded ONCE into registers, counted and ranked in LDS, and its records are written into
// the tile's own window of the record array (bucket-contiguous inside the window, windows packed one after the other) --
// 64 KB that stay in L2 until every line is full.  The index becomes a list of tiles, each with its own table of bucket
// starts: cell c of the scene = one run of records per tile, [cs[t][c], cs[t][c + 1]) for t < ntiles; the searches
// enumerate (row, tile) pairs where they enumerated rows.  Tiles beyond the cloud's last one hold empty runs.
#ifndef AMK_TILE_P
#define AMK_TILE_P 8    // same-box A/B of the step, k steps/s steady / over 20 steps: 6 (106 VGPRs, 17 tiles at 50 k points) 564 / 460,
                        // 8 (122, 13 tiles) 566 / 452, 10 (141, 10 tiles) 558 / 442; 12 needs 159 VGPRs: no room beside a solve wave
#endif
constexpr int kTileP = AMK_TILE_P;                                     // points per thread and tile (in registers from load to store)
constexpr int kTilePoints = kGridBuildThreads * kTileP;        // 4096
#ifndef AMK_STAGE_RECORDS
#define AMK_STAGE_RECORDS 4096   // same-box A/B, 256 obstacle builds alone / step steady / flight: 1024 (16 KB) 78.8 us / 547 k / 1.02 M,
#endif                           // 2048 71.1 / 555 k / 1.05 M, 4096 (the whole tile in one run, 64 KB) 68.7 / 556 k / 1.06 M; unstaged 85-88 / 555 k / 1.04 M
constexpr int kStageRecords = AMK_STAGE_RECORDS;               // LDS per build block: 16 B x this (two blocks per CU hold 138 of 160 KB)
__host__ __device__ constexpr int grid_tiles(int max_points) { return max_points <= 0 ? 1 : (max_points + kTilePoints - 1) / kTilePoints; }

__device__ __forceinline__ void grid_build_tiles_scene(int s, const float *__restrict__ src, int stride, int cap, int nvis,
                                                       const float *__restrict__ bbox, float4 *__restrict__ GP,
                                                       int *__restrict__ cell_start, int ntiles, double *__restrict__ gparams,
                                                       int *__restrict__ size_out, float *__restrict__ pmax_out,
                                                       float *__restrict__ soa_x = nullptr, float *__restrict__ soa_y = nullptr,
                                                       float *__restrict__ soa_z = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NW = kGridBuildThreads / 64;
    float4 *gpt4 = GP + (size_t)s * cap;
    int *cs_all = cell_start + (size_t)s * ntiles * (kGridMaxCells + 2);
    double *gp = gparams + (size_t)s * kGridParamDoubles;
    __shared__ int hist[kGridMaxCells + 2];
    __shared__ int wsum[NW];
    __shared__ int wcnt[kTileP * NW + 1];   // kept points per (round j, wave), then before it; [kTileP * NW] = kept in the tile
    __shared__ float4 stage[kStageRecords]; // the records of kStageRecords consecutive positions of the tile's window, on their way out
    __shared__ float wmax[NW];
    __shared__ double geo[kGridParamDoubles];
    if (tid == 0) grid_geometry(bbox + 6 * s, nvis, geo, gp);
    __syncthreads();
    const double b0 = geo[0], b1 = geo[1], b2 = geo[2], inv_h = geo[4];
    const int g0 = (int)geo[5], g1 = (int)geo[6], g2 = (int)geo[7];
    const int ncell = g0 * g1 * g2;  // + one trash bucket (index ncell) for non-finite points
    float amax = 0.f;                // max |coordinate| over the kept points (fmaxf drops NaNs)
    int kept_before = 0;             // kept points of the tiles before this one = first record of this tile's window
    int t = 0;
    for (int i0 = 0; i0 < nvis; i0 += kTilePoints, ++t) {
        int *cs = cs_all + (size_t)t * (kGridMaxCells + 2);
        float x[kTileP], y[kTileP], z[kTileP];
        // point i0 + j * threads + tid: consecutive lanes read consecutive points, through ONE per-lane offset and a
        // wave-uniform base per round (scalar registers) instead of kTileP address pairs; the round that straddles the end of
        // the cloud clamps its index, rounds behind it load nothing (NaN x = not kept)
        // (one wave-uniform base per round + a 32-bit lane offset would save the address pairs, but the rounds then sit in
        // uniform branches and their loads no longer overlap: 172 us against 115 for the build alone)
        // (round 5: the NEXT tile's loads issued before this tile's stores, so that they fly while the records leave -- same
        // registers, 126 VGPRs -- measured SLOWER, 362-387 us against 351-363 per 1024-scene launch, same box; with non-temporal
        // loads 370; profiles/r05_ab_build_prefetch.txt.  The two blocks of a CU already overlap each other's phases.)
#pragma unroll
        for (int j = 0; j < kTileP; ++j) {
            const int i = i0 + j * kGridBuildThreads + tid;
            const float *q = src + (size_t)(i < nvis ? i : i0) * stride;
            x[j] = q[0]; y[j] = q[1]; z[j] = q[2];
        }
        for (int i = tid; i <= ncell + 1; i += kGridBuildThreads) hist[i] = 0;
        unsigned cell2[kTileP / 2];   // the points' cells, two 16-bit ids per register (<= 1025)
        unsigned keepbits = 0;
#pragma unroll
        for (int j = 0; j < kTileP; ++j) {
            const int i = i0 + j * kGridBuildThreads + tid;
            const bool keep = i < nvis && !(x[j] != x[j]);   // the NaN-x filter of KDTreeTwo::InitializeNew (kd_tree_two.h:96-101)
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wcnt[j * NW + w] = __popcll(m);
            keepbits |= keep ? 1u << j : 0u;
            if (keep) amax = fmaxf(amax, fmaxf(fabsf(x[j]), fmaxf(fabsf(y[j]), fabsf(z[j]))));
            unsigned c = finite3(x[j], y[j], z[j])
                             ? (cell_of(z[j], b2, inv_h, g2) * g1 + cell_of(y[j], b1, inv_h, g1)) * g0 + cell_of(x[j], b0, inv_h, g0)
                             : ncell;
            asm volatile("" : "+v"(c));   // one point's fp64 temporaries at a time: the tile's point registers leave no room for more
            cell2[j / 2] = (j & 1) ? (cell2[j / 2] | c << 16) : c;
        }
        __syncthreads();
        // count per cell; meanwhile wave 0 turns the per-(round, wave) counts into "kept before" (the cloud index of a kept
        // point = kept points before it in the caller's order = kept_before + before its (round, wave) + kept lanes below it)
#pragma unroll
        for (int j = 0; j < kTileP; ++j)
            if (keepbits >> j & 1) atomicAdd(&hist[cell2[j / 2] >> (16 * (j & 1)) & 0xffffu], 1);
        if (w == 0) {
            constexpr int PER = (kTileP * NW + 63) / 64;
            int loc[PER], sum = 0;
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int k = lane * PER + e;
                loc[e] = k < kTileP * NW ? wcnt[k] : 0;
                sum += loc[e];
            }
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            int run = incl - sum;
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int k = lane * PER + e;
                if (k < kTileP * NW) wcnt[k] = run;
                run += loc[e];
            }
            if (lane == 63) wcnt[kTileP * NW] = incl;
        }
        __syncthreads();
        // exclusive scan of hist[0 .. ncell]: bucket starts of this tile (global table) and scatter cursors (LDS)
        const int kept_tile = wcnt[kTileP * NW];
        {
            const int per = (ncell + 1 + kGridBuildThreads - 1) / kGridBuildThreads;
            const int c0 = tid * per;
            int loc = 0;
            for (int j = 0; j < per; ++j)
                if (c0 + j <= ncell) loc += hist[c0 + j];
            int incl = loc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if (lane == 63) wsum[w] = incl;
            __syncthreads();
            int wbase = 0;
            for (int j = 0; j < w; ++j) wbase += wsum[j];
            int run = kept_before + wbase + incl - loc;
            for (int j = 0; j < per; ++j)
                if (c0 + j <= ncell) {
                    const int c = hist[c0 + j];
                    hist[c0 + j] = run;
                    cs[c0 + j] = run;
                    run += c;
                }
            if (tid == 0) cs[ncell + 1] = kept_before + kept_tile;
        }
        __syncthreads();
        // records into the tile's window (order inside a bucket is irrelevant: results are ordered by (distance, index)).
        // Round 4: through LDS.  Rounds 2-3 stored every record straight to its bucket position -- 64 lanes, 64 different
        // buckets, 64 separate 16-byte transactions per store instruction: with the stores made (wrongly) consecutive the
        // build ran 27 % faster, i.e. the write path was transaction-bound.  Now every point takes its position once (LDS
        // atomic on its bucket's cursor), the records of kStageRecords consecutive positions are put in place in LDS, and
        // the staged run leaves as whole 1 KB stores per wavefront.
        // position inside the tile's window (< kTilePoints; kept points only) in place of the cell id: two 16-bit values per register
#pragma unroll
        for (int j = 0; j < kTileP; ++j) {
            unsigned r = 0;
            if (keepbits >> j & 1) r = (unsigned)(atomicAdd(&hist[cell2[j / 2] >> (16 * (j & 1)) & 0xffffu], 1) - kept_before);
            cell2[j / 2] = (j & 1) ? ((cell2[j / 2] & 0xffffu) | r << 16) : ((cell2[j / 2] & 0xffff0000u) | r);
        }
        for (int h0 = 0; h0 < kept_tile; h0 += kStageRecords) {
#pragma unroll
            for (int j = 0; j < kTileP; ++j) {
                const bool keep = keepbits >> j & 1;
                const unsigned long long m = __ballot(keep);
                const int r = (int)(cell2[j / 2] >> (16 * (j & 1)) & 0xffffu) - h0;
                if (keep && r >= 0 && r < kStageRecords) {
                    const int idx = kept_before + wcnt[j * NW + w] + __popcll(m & ((1ull << lane) - 1ull));
                    stage[r] = make_float4(x[j], y[j], z[j], __int_as_float(idx));
                    // the index-ordered planes (only a handle in nanoflann tie order asks for them here: its tree is built from
                    // them right after this kernel): consecutive lanes hold consecutive cloud indices, so these are whole-line
                    // stores, where making the planes from the finished records is a scatter (0.3 ms per 256 x 50k points)
                    if (soa_x) { soa_x[idx] = x[j]; soa_y[idx] = y[j]; soa_z[idx] = z[j]; }
                }
            }
            __syncthreads();
            const int nrec = min(kStageRecords, kept_tile - h0);
            for (int i = tid; i < nrec; i += kGridBuildThreads) gpt4[kept_before + h0 + i] = stage[i];
            __syncthreads();   // the staging buffer is refilled by the next run
        }
        kept_before += kept_tile;
        __syncthreads();   // hist / wcnt are rewritten by the next tile
    }
    if (soa_x) {   // NaN padding behind the points, as kd_records_to_soa_kernel leaves it
        const float qnan = __builtin_nanf("");
        for (int i = kept_before + tid; i < cap; i += kGridBuildThreads) { soa_x[i] = qnan; soa_y[i] = qnan; soa_z[i] = qnan; }
    }
    // tiles the cloud does not reach: empty runs
    for (int tt = t; tt < ntiles; ++tt) {
        int *cs = cs_all + (size_t)tt * (kGridMaxCells + 2);
        for (int i = tid; i <= ncell + 1; i += kGridBuildThreads) cs[i] = kept_before;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) wmax[w] = amax;
    __syncthreads();
    if (tid == 0) {
        float mx = 0.f;
        for (int j = 0; j < NW; ++j) mx = fmaxf(mx, wmax[j]);
        *size_out = kept_before;
        *pmax_out = mx;
    }
}

static __global__ __launch_bounds__(kGridBuildThreads) void kd_grid_build_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ Z, int cap,
    const int *__restrict__ sizes, const float *__restrict__ bbox, float4 *__restrict__ GP,
    int *__restrict__ cell_start, int ntiles, double *__restrict__ gparams) {
    const int s = blockIdx.x;
    grid_build_scene(s, X + (size_t)s * cap, Y + (size_t)s * cap, Z + (size_t)s * cap, cap, sizes[s], bbox, GP, cell_start, ntiles,
                     gparams);
}

// ------------------------------------------------------------------------------------------------
// search: one wavefront per query
// ------------------------------------------------------------------------------------------------
struct GridScene {
    const float4 *pt;           // bucket-contiguous points (x, y, z, index in the NaN-x-filtered cloud)
    const int *cs;              // bucket starts of every tile, [nt][kGridMaxCells + 2] (entries [0, ncell + 1] used)
    const double *gp;           // geometry
    int nt;                     // tiles: cell c = the runs [cs[t][c], cs[t][c + 1]) for t < nt
};

struct GridPtrs {  // the batch: what a kernel needs to find scene s
    const float4 *pt;
    const int *cs;
    const double *gp;
    int cap;
    int nt;
    __device__ __forceinline__ GridScene scene(int s) const {
        GridScene g;
        g.pt = pt + (size_t)s * cap;
        g.cs = cs + (size_t)s * nt * (kGridMaxCells + 2);
        g.gp = gp + (size_t)s * kGridParamDoubles;
        g.nt = nt;
        return g;
    }
};

struct GridWaveLds {  // per-wavefront scratch: ranges of the rows of the current shell
    int4 item[64];   // (first candidate number, length of run A, start of run A, start of run B) of the (row, tile) item of lane o
    int mark[65];    // slot j of the current 64-candidate window: (window number << 6 | o) of the item whose candidates begin there
                     // (slot 64: where the lanes with nothing to announce write)
};

// DPP data movement inside the wavefront (no LDS crossbar, no s_waitcnt): whole-wave shift by one lane and the
// gfx9 inclusive-scan ladder (row_shr 1/2/4/8, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2, 3)
__device__ __forceinline__ int wave_shr1_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ double wave_shr1_f64(double v) {
    return __hiloint2double(wave_shr1_i32(__double2hiint(v)), wave_shr1_i32(__double2loint(v)));
}
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// inclusive running maximum over the lanes, values >= 0 (the same ladder with v_max)
__device__ __forceinline__ int wave_incl_scan_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));
    return v;
}

// minimum over the 64 lanes of non-negative ints (wave-uniform result): one v_min_i32 with a DPP operand per step
__device__ __forceinline__ int grid_wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));    // quad_perm:[1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));    // quad_perm:[2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));   // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// minimum over the 64 lanes (every lane gets it): DPP inside the rows of 16, v_readlane across them
__device__ __forceinline__ double grid_wave_min_f64(double v) {
    auto dpp = [](double x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), C, 0xF, 0xF, true),
                                __builtin_amdgcn_update_dpp(0, __double2loint(x), C, 0xF, 0xF, true));
    };
    v = fmin(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm:[1,0,3,2]
    v = fmin(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm:[2,3,0,1]
    v = fmin(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
    v = fmin(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// Exact k nearest neighbours of (qx,qy,qz).  On return lane i < k holds the i-th best (squared
// distance, index) in (ld, li) and the position of its record in gs.pt in lpos (the neighbour's coordinates are
// gs.pt[lpos]: the index build keeps no index-ordered copy of the points); empty slots hold (DBL_MAX, kNoIndex, 0).
// Ties order by index.
//
// The cells are visited in Chebyshev rings around the query's cell (shell (rp, r] with r = rp + 1; two rings per
// round were measured slower: the BASELINE clouds are dense enough that ring r + 1 is rarely needed).  Once k
// candidates are known, a ring only visits the cells that the ball of the current k-th distance reaches: a row
// (iy, iz) is skipped when its slab is farther than that, and its run of cells is clipped to the ball's x extent.
#ifdef AMK_KNN_COUNT   // diagnostics build: queries, item batches, candidate batches, insertions, candidates (tools/experiments/knn_counts.py)
__device__ unsigned long long g_knn_cnt[8];
#endif
__device__ __forceinline__ void grid_knn(const GridScene &gs, double qx, double qy, double qz, int k, double &ld,
                                         int &li, int &lpos, GridWaveLds *ws) {
    const int lane = threadIdx.x & 63;
#ifdef AMK_KNN_COUNT
    unsigned c_items = 0, c_batches = 0, c_ins = 0, c_cand = 0, c_rings = 0;
#endif
    const double b[3] = {gs.gp[0], gs.gp[1], gs.gp[2]};
    const double h = gs.gp[3], inv_h = gs.gp[4];
    const int g[3] = {(int)gs.gp[5], (int)gs.gp[6], (int)gs.gp[7]};
    const double q[3] = {qx, qy, qz};
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = (q[a] == q[a]) ? cell_of(q[a], b[a], inv_h, g[a]) : 0;
    ld = DBL_MAX;
    li = kNoIndex;
    lpos = 0;
    double tau = DBL_MAX;
    int rmax = 0;
    int win = 0;           // number of the candidate window being fetched (tags ws->mark)
    ws->mark[lane] = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) rmax = max(rmax, max(c[a], g[a] - 1 - c[a]));
    // rounding slack of the cell boundaries (cell_of is evaluated in fp64 on fp32 coordinates)
    const double slack = 1e-9 * h + 1e-12 * (fabs(qx) + fabs(qy) + fabs(qz) + fabs(b[0]) + fabs(b[1]) + fabs(b[2]));
    int rp = -1;  // everything within Chebyshev radius rp of the query's cell has been seen
    for (int r = 0;; ) {
        // the (iy, iz) rows of the shell that lie inside the grid: [ylo, yhi] x [zlo, zhi] (a degenerate grid -- 1024 x 1 x 1
        // cells of a cloud on a line -- has few rows however large r gets; at most kGridMaxCells of them)
        const int ylo = max(c[1] - r, 0), ny = min(c[1] + r, g[1] - 1) - ylo + 1;
        const int zlo = max(c[2] - r, 0), nz = min(c[2] + r, g[2] - 1) - zlo + 1;
        const int nitems = ny * nz * gs.nt;
        // item -> (row, tile) and row -> (iy, iz) through float reciprocals: exact while nitems < 4e6 ((x + 0.5) / n is at
        // least 0.5 / n away from an integer; nitems <= kGridMaxCells x tiles); the integer-division sequences cost 40
        // instructions and 10 registers
        const float inv_nt = 1.0f / (float)gs.nt, inv_ny = 1.0f / (float)ny;
        for (int row0 = 0; row0 < nitems; row0 += 64) {
            // lane -> one (iy, iz) row of the shell (rp, r] in one tile: the whole run of cells x in [cx - r, cx + r] if
            // the row lies outside the box already seen, else the two end runs left and right of that box
            const int item = row0 + lane;
            int sA = 0, lA = 0, sB = 0, lB = 0;
            if (item < nitems) {
                const int j = (int)(((float)item + 0.5f) * inv_nt);
                const int *cst = gs.cs + (size_t)(item - j * gs.nt) * (kGridMaxCells + 2);
                const int jz = (int)(((float)j + 0.5f) * inv_ny);
                const int iy = ylo + (j - jz * ny), iz = zlo + jz;
                const int dy = iy - c[1], dz = iz - c[2];
                {
                    const int rowbase = (iz * g[1] + iy) * g[0];
                    int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g[0] - 1);
                    if (tau < DBL_MAX) {  // clip to the ball of radius sqrt(tau) (conservatively: + slack)
                        const double ey = dy < 0 ? q[1] - (b[1] + (double)(iy + 1) * h) : (dy > 0 ? (b[1] + (double)iy * h) - q[1] : 0.0);
                        const double ez = dz < 0 ? q[2] - (b[2] + (double)(iz + 1) * h) : (dz > 0 ? (b[2] + (double)iz * h) - q[2] : 0.0);
                        const double fy = fmax(0.0, ey - slack), fz = fmax(0.0, ez - slack);
                        const double rem = tau * (1.0 + 1e-12) - (fy * fy + fz * fz);
                        if (rem < 0.0) {
                            x1 = x0 - 1;  // the whole row is out of reach
                        } else {
                            const double rx = sqrt(rem) * (1.0 + 1e-12) + slack;
                            x0 = max(x0, cell_of(q[0] - rx, b[0], inv_h, g[0]));
                            x1 = min(x1, cell_of(q[0] + rx, b[0], inv_h, g[0]));
                        }
                    }
                    if (dy < -rp || dy > rp || dz < -rp || dz > rp) {
                        if (x0 <= x1) {
                            sA = cst[rowbase + x0];
                            lA = cst[rowbase + x1 + 1] - sA;
                        }
                    } else {
                        const int a1 = min(c[0] - rp - 1, x1), b0 = max(c[0] + rp + 1, x0);
                        if (x0 <= a1) {
                            sA = cst[rowbase + x0];
                            lA = cst[rowbase + a1 + 1] - sA;
                        }
                        if (b0 <= x1) {
                            sB = cst[rowbase + b0];
                            lB = cst[rowbase + x1 + 1] - sB;
                        }
                    }
                }
            }
#ifdef AMK_KNN_COUNT
            ++c_items;
#endif
            if (__ballot(lA + lB > 0) == 0) continue;  // nothing but empty buckets in these rows
            // flatten the <= 128 ranges into one index space so that every lane gets a point
            const int incl = wave_incl_scan_i32(lA + lB);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int mypre = incl - (lA + lB), mylen = lA + lB;
            ws->item[lane] = make_int4(mypre, lA, sA, sB);
            // candidate t of the window [w0, w0 + 64) belongs to the last non-empty item that begins at or before t.  Rounds 1-5
            // found it by a six-step binary search over LDS (six dependent round trips per batch); now the items that BEGIN inside
            // the window announce their slot, a running maximum over the lanes carries them forward, and the item that holds w0
            // comes from a ballot.  Entries of earlier windows are told apart by the window number.
            auto fetch = [&](int w0, double &d, int &ic, int &ip) {
                ++win;
                const int rel = mypre - w0;
                // every lane stores (the silent ones into the spare slot) and the wave is fenced before it reads: with the store
                // under a branch the compiler placed the LOAD below inside that branch as well, and only the announcing lanes got
                // their slot (lanes communicating through LDS without a fence are a data race by the letter of the memory model)
                ws->mark[(mylen > 0 && rel >= 0 && rel < 64) ? rel : 64] = (win << 6) | lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned long long before = __ballot(mylen > 0 && mypre <= w0);
                const int carry = 63 - __clzll((long long)before);   // (w0 < total: some non-empty item holds w0)
                const int mk = ws->mark[lane];
                const int o = max(wave_incl_scan_max_i32((mk >> 6) == win ? (mk & 63) + 1 : 0) - 1, carry);
                const int t = w0 + lane;
                d = __builtin_nan("");
                ic = kNoIndex;
                ip = 0;
                if (t < total) {
                    const int4 it = ws->item[o];
                    const int u = t - it.x;
                    const int pos = u < it.y ? it.z + u : it.w + (u - it.y);
                    const float4 p4 = gs.pt[pos];
                    d = sq_dist(qx, qy, qz, p4.x, p4.y, p4.z);
                    ic = __float_as_int(p4.w);
                    ip = pos;
                }
            };
            double d, dn;
            int ic, icn, ip, ipn;
            fetch(0, d, ic, ip);
#ifdef AMK_KNN_COUNT
            c_cand += total;
#endif
            for (int t0 = 0; t0 < total; t0 += 64) {
#ifdef AMK_KNN_COUNT
                ++c_batches;
#endif
                if (t0 + 64 < total) fetch(t0 + 64, dn, icn, ipn);  // next batch in flight while this one is merged
                // candidates that beat (or tie) the current k-th best enter BEST FIRST: the k nearest of a batch tighten tau as
                // fast as it can be tightened, so a batch costs about as many insertions as it has entries that end up in the
                // list (<= k) plus exact ties -- rounds 1-3 offered every lane that passed the ballot at the batch's start in
                // lane order: 64 serial insertions for the first batch of every query, whose tau is still infinite.
                bool live = d <= tau;   // (NaN: no candidate)
                for (;;) {
                    const unsigned long long m = __ballot(live);
                    if (!m) break;
                    int src = __ffsll((long long)m) - 1;
                    if (m & (m - 1)) {   // several: (one of) the nearest of them by the HIGH WORD of the distance -- monotone for
                        // doubles >= 0, one v_min_i32 per DPP step where the fp64 minimum took three instructions; candidates that
                        // agree in those 32 bits enter in lane order (any order gives the same list: the insertion ranks by
                        // (distance, index))
                        const int hk = live ? __double2hiint(d) : 0x7FFFFFFF;
                        const int hmin = grid_wave_min_i32(hk);
                        src = __ffsll((long long)__ballot(hk == hmin)) - 1;
                    }
                    const double dc = readlane_f64(d, src);
                    const int icc = __builtin_amdgcn_readlane(ic, src);
                    const int ipc = __builtin_amdgcn_readlane(ip, src);
                    // rank of the candidate in (distance, index) order among the kept entries
                    // (bitwise, not short-circuit: three compares and two scalar ANDs instead of nested exec-mask regions)
                    const bool lt = (lane < k) & ((ld < dc) | ((ld == dc) & (li < icc)));
                    const int pos = __popcll(__ballot(lt));
#ifdef AMK_KNN_COUNT
                    ++c_ins;
#endif
                    if (pos < k && dc < DBL_MAX) {
                        const double up_d = wave_shr1_f64(ld);
                        const int up_i = wave_shr1_i32(li), up_p = wave_shr1_i32(lpos);
                        const bool above = lane > pos, here = lane == pos;   // selects, not branches
                        ld = here ? dc : (above ? up_d : ld);
                        li = here ? icc : (above ? up_i : li);
                        lpos = here ? ipc : (above ? up_p : lpos);
                        tau = readlane_f64(ld, k - 1);
                    }
                    live = live && lane != src && d <= tau;
                }
                d = dn;
                ic = icn;
                ip = ipn;
            }
        }
        // everything within Chebyshev radius r of the query's cell has been seen.  A cell outside that box
        // differs by more than r along some axis, so it lies beyond one of the box's faces; stop when the
        // k-th best is closer than the nearest face that still has cells behind it.
        if (tau < DBL_MAX) {
            double dmin = DBL_MAX;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (c[a] - r > 0) dmin = fmin(dmin, fmax(0.0, q[a] - (b[a] + (double)(c[a] - r) * h)));
                if (c[a] + r < g[a] - 1) dmin = fmin(dmin, fmax(0.0, (b[a] + (double)(c[a] + r + 1) * h) - q[a]));
            }
            if (dmin == DBL_MAX) break;  // the box covers the whole grid
            dmin = fmax(0.0, dmin - slack);
            if (tau < dmin * dmin) break;
        }
        if (r >= rmax) break;
        rp = r;
        r = r + 1;
#ifdef AMK_KNN_COUNT
        ++c_rings;
#endif
    }
#ifdef AMK_KNN_COUNT
    if (lane == 0) {
        atomicAdd(&g_knn_cnt[0], 1ull); atomicAdd(&g_knn_cnt[1], (unsigned long long)c_items); atomicAdd(&g_knn_cnt[2], (unsigned long long)c_batches);
        atomicAdd(&g_knn_cnt[3], (unsigned long long)c_ins); atomicAdd(&g_knn_cnt[4], (unsigned long long)c_cand); atomicAdd(&g_knn_cnt[5], (unsigned long long)c_rings);
    }
#endif
}

// 1-NN squared distance of q, one THREAD per query (the n-queries-on-n-points keyframe sweep,
// AM/src/FrameKDMap.cpp:466-476): same ring walk and stopping rule as grid_knn with k = 1, scalar per lane.
// Returns DBL_MAX when the index holds no point with a finite distance to q.
__device__ __forceinline__ double grid_nn1_thread(const GridScene &gs, double qx, double qy, double qz) {
    const double b[3] = {gs.gp[0], gs.gp[1], gs.gp[2]};
    const double h = gs.gp[3], inv_h = gs.gp[4];
    const int g[3] = {(int)gs.gp[5], (int)gs.gp[6], (int)gs.gp[7]};
    const double q[3] = {qx, qy, qz};
    if (!(qx == qx && qy == qy && qz == qz)) return DBL_MAX;
    int c[3], rmax = 0;
    for (int a = 0; a < 3; ++a) {
        c[a] = cell_of(q[a], b[a], inv_h, g[a]);
        rmax = max(rmax, max(c[a], g[a] - 1 - c[a]));
    }
    const double slack = 1e-9 * h + 1e-12 * (fabs(qx) + fabs(qy) + fabs(qz) + fabs(b[0]) + fabs(b[1]) + fabs(b[2]));
    double best = DBL_MAX;
    for (int r = 0; r <= rmax; ++r) {
        for (int dz = max(-r, -c[2]); dz <= min(r, g[2] - 1 - c[2]); ++dz) {   // (only the rows inside the grid)
            const int iz = c[2] + dz;
            for (int dy = max(-r, -c[1]); dy <= min(r, g[1] - 1 - c[1]); ++dy) {
                const int iy = c[1] + dy;
                const int rowbase = (iz * g[1] + iy) * g[0];
                const bool face = (dy == -r || dy == r || dz == -r || dz == r);
                const int x0 = c[0] - r, x1 = c[0] + r;
                for (int pt = 0; pt < 2 * gs.nt; ++pt) {   // (part, tile)
                    const int part = pt & 1;
                    const int *cst = gs.cs + (size_t)(pt >> 1) * (kGridMaxCells + 2);
                    int s0, s1;
                    if (face) {
                        if (part == 1) continue;
                        const int a0 = max(x0, 0), a1 = min(x1, g[0] - 1);
                        if (a0 > a1) break;
                        s0 = cst[rowbase + a0];
                        s1 = cst[rowbase + a1 + 1];
                    } else {
                        const int ix = part == 0 ? x0 : x1;
                        if (ix < 0 || ix >= g[0] || (part == 1 && r == 0)) continue;
                        s0 = cst[rowbase + ix];
                        s1 = cst[rowbase + ix + 1];
                    }
                    for (int pos = s0; pos < s1; ++pos) {
                        const float4 p4 = gs.pt[pos];
                        const double d = sq_dist(qx, qy, qz, p4.x, p4.y, p4.z);
                        best = d < best ? d : best;
                    }
                }
            }
        }
        if (best < DBL_MAX) {
            double dmin = DBL_MAX;
            for (int a = 0; a < 3; ++a) {
                if (c[a] - r > 0) dmin = fmin(dmin, fmax(0.0, q[a] - (b[a] + (double)(c[a] - r) * h)));
                if (c[a] + r < g[a] - 1) dmin = fmin(dmin, fmax(0.0, (b[a] + (double)(c[a] + r + 1) * h) - q[a]));
            }
            if (dmin == DBL_MAX) break;
            dmin = fmax(0.0, dmin - slack);
            if (best < dmin * dmin) break;
        }
    }
    return best;
}


// The keyframe sweep does not need the nearest DISTANCE, only whether it exceeds th (FrameKDMap.cpp:470-475: outlier iff
// sqrt(squared_distances[0]) > mParamKeyframeDistanceTh) -- a FIXED-RADIUS question: a point within th of q lies in a cell that
// the cube [q - th, q + th]^3 touches (cell_of is monotone per axis, clamping included), so only those cells are read -- one to
// eight of them for th below the cell edge, against the 27 of the two rings the exact walk needs before it may stop (measured on
// the 50 k-point flight frames: 20-24 ms per 512-scene sweep with the ring walk once the vehicles fly among the cylinders,
// where most keyframe points have their nearest current point in a NEIGHBOUR cell or none within th).  Leaves at the first point
// within th (sqrt is monotone: the minimum is within th too).  Returns 1 = outlier, 0 = not (also when the index holds no
// point with finite coordinates: the reference then has no result to test).  Same flags as `sqrt(grid_nn1_thread(...)) > th`
// (tests/test_keyframe_gpu.py, test_kfmap_gpu.py).
#ifndef AMK_SWEEP_TILES
#define AMK_SWEEP_TILES 2
#endif
#ifndef AMK_SWEEP_RECS
#define AMK_SWEEP_RECS 6
#endif
constexpr int kSweepTiles = AMK_SWEEP_TILES;   // tiles whose bucket-table entries a sweep thread reads in one step
constexpr int kSweepRecs = AMK_SWEEP_RECS;     // records of a run it reads in one step
__device__ __forceinline__ int grid_outlier_thread(const GridScene &gs, double qx, double qy, double qz, double th) {
    const double b[3] = {gs.gp[0], gs.gp[1], gs.gp[2]};
    const double h = gs.gp[3], inv_h = gs.gp[4];
    const int g[3] = {(int)gs.gp[5], (int)gs.gp[6], (int)gs.gp[7]};
    const double q[3] = {qx, qy, qz};
    if (!(qx == qx && qy == qy && qz == qz)) return 0;
    // (the cube is widened by a rounding allowance: a point whose computed distance is <= th may be th + 1 ulp away along one axis)
    const double r = th + 1e-9 * h + 1e-12 * (fabs(qx) + fabs(qy) + fabs(qz) + fabs(b[0]) + fabs(b[1]) + fabs(b[2]) + th);
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = cell_of(q[a] - r, b[a], inv_h, g[a]);
        hi[a] = cell_of(q[a] + r, b[a], inv_h, g[a]);
    }
    // sqrt(d) <= th is decided without the square root except in a band of relative width 2e-15 around th^2 (the correctly rounded
    // sqrt and the rounded square differ from the real ones by < 2.3e-16 relative: outside the band both tests agree)
    const double t2 = th * th, t2lo = t2 * (1.0 - 1e-15), t2hi = t2 * (1.0 + 1e-15);
    // the query as floats: exact when it IS a float (the sweep's queries are records); otherwise the screen is off
    const float qxf = (float)qx, qyf = (float)qy, qzf = (float)qz;
    const bool q_is_float = (double)qxf == qx && (double)qyf == qy && (double)qzf == qz;
    const float t2f = q_is_float ? (float)(t2 * (1.0 + 1e-5)) * (1.0f + 1e-6f) : __builtin_inff();
    // rows (iy, iz) of the cube, the query's OWN row first: that is where a point within th most likely lies, and the first one ends the walk
    const int ny = hi[1] - lo[1] + 1, nrows = ny * (hi[2] - lo[2] + 1);
    const int own = (cell_of(q[2], b[2], inv_h, g[2]) - lo[2]) * ny + (cell_of(q[1], b[1], inv_h, g[1]) - lo[1]);
    for (int r = 0; r < nrows; ++r) {
        {
            const int rr = r == 0 ? own : (r <= own ? r - 1 : r);
            const int iz = lo[2] + rr / ny, iy = lo[1] + rr % ny;
            const int rowbase = (iz * g[1] + iy) * g[0];
            // the run of cells [lo x, hi x] of this row, tile by tile.  The walk is a chain of dependent loads (bucket table, then
            // the run's records) and nothing else: the table entries of kSweepTiles tiles are fetched together, then kSweepRecs
            // records of a run at a time (a run holds ~5; indices past its end repeat its last record, which decides nothing
            // new).  Wider steps cost registers, and the kernel lives on the number of threads in flight: see the table in
            // profiles/r05_sweep_mlp.txt.
            for (int t0 = 0; t0 < gs.nt; t0 += kSweepTiles) {
                int s0[kSweepTiles], s1[kSweepTiles];
#pragma unroll
                for (int u = 0; u < kSweepTiles; ++u) {
                    const int *cst = gs.cs + (size_t)min(t0 + u, gs.nt - 1) * (kGridMaxCells + 2);
                    s0[u] = cst[rowbase + lo[0]];
                    s1[u] = t0 + u < gs.nt ? cst[rowbase + hi[0] + 1] : s0[u];
                }
#pragma unroll
                for (int u = 0; u < kSweepTiles; ++u) {
                    for (int pos = s0[u]; pos < s1[u]; pos += kSweepRecs) {
                        const int last = s1[u] - 1;
                        float4 p[kSweepRecs];
#pragma unroll
                        for (int e = 0; e < kSweepRecs; ++e) p[e] = gs.pt[min(pos + e, last)];
                        // fp32 screen (both points are floats; the fp32 squared distance is within 3e-7 relative of the real one): a
                        // step none of whose records comes within t2 (1 + 1e-5) cannot pass the exact test -- the candidates' fp64
                        // distances (11 VALU instructions each) were what bounded the sweep (profiles/r05_sweep_target.txt)
                        float m32 = FLT_MAX;
#pragma unroll
                        for (int e = 0; e < kSweepRecs; ++e) {
                            const float dx = qxf - p[e].x, dy = qyf - p[e].y, dz = qzf - p[e].z;
                            m32 = fminf(m32, dx * dx + dy * dy + dz * dz);
                        }
                        if (!(m32 <= t2f)) continue;
                        double d = sq_dist(qx, qy, qz, p[0].x, p[0].y, p[0].z);
#pragma unroll
                        for (int e = 1; e < kSweepRecs; ++e) d = fmin(d, sq_dist(qx, qy, qz, p[e].x, p[e].y, p[e].z));
                        // (the test is monotone in d: the minimum decides for all.  NaN distances: records never hold them --
                        // NaN-x points are dropped by the build, other non-finite ones sit in the trash bucket)
                        if (d <= t2lo || (d <= t2hi && sqrt(d) <= th)) return 0;
                    }
                }
            }
        }
    }
    int finite = 0;   // an outlier needs a nearest point at all: does the index hold a point with finite coordinates?
    for (int t = 0; t < gs.nt; ++t) {
        const int *cst = gs.cs + (size_t)t * (kGridMaxCells + 2);
        finite += cst[g[0] * g[1] * g[2]] - cst[0];   // (bucket ncell is the trash bucket of the non-finite points)
    }
    return finite > 0 ? 1 : 0;
}

}  // namespace amk
