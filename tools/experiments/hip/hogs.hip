// Interference probes: which CU resource is the solve short of when 8 solve waves share a CU?  Each "hog" loads ONE
// resource (LDS bandwidth, fp64 VALU issue, HBM bandwidth) from another stream while the solves run
// (tools/experiments/interfere.py).  One wave per block, tiny LDS / VGPR footprint, `waves_per_cu` blocks per CU.
#include <hip/hip_runtime.h>
extern "C" {
__global__ __launch_bounds__(64) void lds_hog(int iters, double *sink) {
    __shared__ double buf[128];
    const int l = threadIdx.x;
    buf[l] = l; buf[l + 64] = l;
    double a = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a += buf[(l + u * 5 + i) & 127];   // conflict-free 8-byte reads
        buf[(l + i) & 127] = a;
    }
    if (a == 1.2345) sink[0] = a;
}
__global__ __launch_bounds__(64) void valu_hog(int iters, double *sink) {
    double a[8];
    for (int u = 0; u < 8; ++u) a[u] = threadIdx.x + u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = fma(a[u], 1.0000001, 0.5);  // 8 independent fp64 chains
    }
    double s = 0;
    for (int u = 0; u < 8; ++u) s += a[u];
    if (s == 1.2345) sink[0] = s;
}
__global__ __launch_bounds__(256) void mem_hog(const float4 *src, float4 *dst, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void launch_lds_hog(int blocks, int iters, double *sink, void *stream) { hipLaunchKernelGGL(lds_hog, dim3(blocks), dim3(64), 0, (hipStream_t)stream, iters, sink); }
void launch_valu_hog(int blocks, int iters, double *sink, void *stream) { hipLaunchKernelGGL(valu_hog, dim3(blocks), dim3(64), 0, (hipStream_t)stream, iters, sink); }
void launch_mem_hog(int blocks, const void *src, void *dst, size_t n, int reps, void *stream) { hipLaunchKernelGGL(mem_hog, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n, reps); }
}
