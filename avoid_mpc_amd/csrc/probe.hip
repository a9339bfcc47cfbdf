// Measurement helpers of libavoid_mpc_amd.so (internal: not part of the C ABI in include/avoid_mpc_amd.h).
//
// amk__hbm_copy_probe: a plain float4 device copy, the "achievable" HBM figure SURVEY.md section 8(d) / BASELINE.md section 3
// ask to report BESIDE the vendor peak (bench.py: hbm_peak_measured_gbs; every HBM fraction of the JSON line is given against
// both).  One 16-byte load + one 16-byte store per thread and round, kCopyUnroll independent rounds in flight, a grid sized to
// the chip (every CU holds its full complement of waves, XCD-interleaved by the hardware's own round robin over blockIdx).
#include "amk_common.h"

namespace {
constexpr int kCopyThreads = 256;
typedef float v4f __attribute__((ext_vector_type(4)));   // (the non-temporal builtins take native vectors, not HIP's float4 struct)
// UNROLL independent 16-byte loads in flight per thread, then the stores; NT: non-temporal (streaming) loads and stores
template <int UNROLL, bool NT>
__global__ __launch_bounds__(kCopyThreads) void hbm_copy_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * kCopyThreads;
    size_t i = (size_t)blockIdx.x * kCopyThreads + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// ONE 16-byte element per thread and a grid as large as the data (no persistent loop): the shape that reaches the guide's
// 6.29 TB/s on this chip (6.33 / 6.28 / 6.36 TB/s at 1 / 2 / 4 GiB with non-temporal stores, 6.13 - 6.24 plain; the best
// persistent shape of 120 tried reached 5.80 -- profiles/r06_hbm_probe_shapes.txt, tools/experiments/hip/hbm_shapes.hip).
// The workgroup dispatcher hands fresh blocks to whichever CU retires one, so every channel always has a requester; a
// persistent block's lanes march in lock step through one stride pattern.
template <bool NT>
__global__ __launch_bounds__(kCopyThreads) void hbm_copy_flat_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * kCopyThreads + threadIdx.x;
    if (i >= n16) return;
    const v4f v = src[i];
    if (NT) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
}

template <bool NT>
int time_copy_flat(const void *d_src, void *d_dst, size_t n16, int reps, hipStream_t stream, double *ms_out) {
    const size_t nblk = (n16 + kCopyThreads - 1) / kCopyThreads;
    if (nblk > 0x7fffffffull) return AMK_ERR_UNSUPPORTED;
    hipEvent_t e0, e1;
    AMK_HIP(hipEventCreate(&e0));
    AMK_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL((hbm_copy_flat_kernel<NT>), dim3((unsigned)nblk), dim3(kCopyThreads), 0, stream, (const v4f *)d_src, (v4f *)d_dst, n16);  // warm-up
    AMK_HIP(hipEventRecord(e0, stream));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((hbm_copy_flat_kernel<NT>), dim3((unsigned)nblk), dim3(kCopyThreads), 0, stream, (const v4f *)d_src, (v4f *)d_dst, n16);
    AMK_HIP(hipEventRecord(e1, stream));
    AMK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    AMK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    AMK_HIP(hipGetLastError());
    *ms_out = (double)ms / reps;
    return AMK_OK;
}

template <int UNROLL, bool NT>
int time_copy(const void *d_src, void *d_dst, size_t n16, int blocks, int reps, hipStream_t stream, double *ms_out) {
    hipEvent_t e0, e1;
    AMK_HIP(hipEventCreate(&e0));
    AMK_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL((hbm_copy_kernel<UNROLL, NT>), dim3(blocks), dim3(kCopyThreads), 0, stream, (const v4f *)d_src, (v4f *)d_dst, n16);  // warm-up
    AMK_HIP(hipEventRecord(e0, stream));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((hbm_copy_kernel<UNROLL, NT>), dim3(blocks), dim3(kCopyThreads), 0, stream, (const v4f *)d_src, (v4f *)d_dst, n16);
    AMK_HIP(hipEventRecord(e1, stream));
    AMK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    AMK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    AMK_HIP(hipGetLastError());
    *ms_out = (double)ms / reps;
    return AMK_OK;
}
}  // namespace

// copies n_bytes (a multiple of 16) from d_src to d_dst `reps` times on `stream` in several shapes -- 2 / 4 / 8 loads in flight per
// thread, plain and non-temporal, 8 / 16 / 32 blocks of 256 threads per CU, and one element per thread on a grid as large as the data
// (plain / non-temporal stores) -- and returns the BEST shape's average milliseconds per
// copy (HIP events on that stream): the figure is "what a copy can reach on this box", not one kernel's.  The caller owns the
// buffers (>= 1 GiB each for a figure that is not the Infinity Cache's).  variant_out (or NULL): unroll * 1000 + nt * 100 + blocks per CU; 1 / 101: the flat shapes.
extern "C" int amk__hbm_copy_probe(const void *d_src, void *d_dst, size_t n_bytes, int reps, void *stream_, double *ms_out, int *variant_out) {
    if (!d_src || !d_dst || n_bytes < 16 || (n_bytes & 15) || reps < 1 || !ms_out) return AMK_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int dev = 0, cus = 256;
    AMK_HIP(hipGetDevice(&dev));
    AMK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t n16 = n_bytes / 16;
    double best = 1e30;
    int best_v = 0;
    for (int bpc : {8, 16, 32}) {
        const int blocks = cus * bpc;
        double ms = 0.0;
        int st;
#define AMK_TRY(U, NT)                                                                   \
        st = time_copy<U, NT>(d_src, d_dst, n16, blocks, reps, stream, &ms);           \
        if (st != AMK_OK) return st;                                                     \
        if (ms < best) { best = ms; best_v = U * 1000 + (NT ? 100 : 0) + bpc; }
        AMK_TRY(2, false) AMK_TRY(4, false) AMK_TRY(8, false) AMK_TRY(2, true) AMK_TRY(4, true) AMK_TRY(8, true)
#undef AMK_TRY
    }
    for (int nt = 0; nt < 2; ++nt) {   // the flat shapes: variant 1 (plain) / 101 (non-temporal stores)
        double ms = 0.0;
        const int st = nt ? time_copy_flat<true>(d_src, d_dst, n16, reps, stream, &ms) : time_copy_flat<false>(d_src, d_dst, n16, reps, stream, &ms);
        if (st != AMK_OK) return st;
        if (ms < best) { best = ms; best_v = 1 + 100 * nt; }
    }
    *ms_out = best;
    if (variant_out) *variant_out = best_v;
    return AMK_OK;
}
