// Measurement helpers of libavoid_mpc_amd.so (internal: not part of the C ABI in include/avoid_mpc_amd.h).
//
// amk__hbm_copy_probe: a plain float4 device copy, the "achievable" HBM figure SURVEY.md section 8(d) / BASELINE.md section 3
// ask to report BESIDE the vendor peak (bench.py: hbm_peak_measured_gbs; every HBM fraction of the JSON line is given against
// both).  One 16-byte load + one 16-byte store per thread and round, kCopyUnroll independent rounds in flight, a grid sized to
// the chip (every CU holds its full complement of waves, XCD-interleaved by the hardware's own round robin over blockIdx).
#include "amk_common.h"

namespace {
constexpr int kCopyThreads = 256, kCopyUnroll = 8;
typedef float v4f __attribute__((ext_vector_type(4)));   // (the non-temporal builtins take native vectors, not HIP's float4 struct)
__global__ __launch_bounds__(kCopyThreads) void hbm_copy_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * kCopyThreads;
    size_t i = (size_t)blockIdx.x * kCopyThreads + threadIdx.x;
    for (; i + (kCopyUnroll - 1) * stride < n16; i += kCopyUnroll * stride) {
        v4f v[kCopyUnroll];
#pragma unroll
        for (int u = 0; u < kCopyUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < kCopyUnroll; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

// copies n_bytes (a multiple of 16) from d_src to d_dst `reps` times on `stream`; *ms_out = average milliseconds per copy (HIP
// events on that stream).  The caller owns the buffers (>= 1 GiB each for a figure that is not the Infinity Cache's).
extern "C" int amk__hbm_copy_probe(const void *d_src, void *d_dst, size_t n_bytes, int reps, void *stream_, double *ms_out) {
    if (!d_src || !d_dst || n_bytes < 16 || (n_bytes & 15) || reps < 1 || !ms_out) return AMK_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int dev = 0, cus = 256;
    AMK_HIP(hipGetDevice(&dev));
    AMK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t n16 = n_bytes / 16;
    const int blocks = cus * 8;   // 8 x 4 waves per CU
    hipEvent_t e0, e1;
    AMK_HIP(hipEventCreate(&e0));
    AMK_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(hbm_copy_kernel, dim3(blocks), dim3(kCopyThreads), 0, stream, (const v4f *)d_src, (v4f *)d_dst, n16);  // warm-up
    AMK_HIP(hipEventRecord(e0, stream));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(hbm_copy_kernel, dim3(blocks), dim3(kCopyThreads), 0, stream, (const v4f *)d_src, (v4f *)d_dst, n16);
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
