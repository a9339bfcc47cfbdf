#include <hip/hip_runtime.h>
#include <cstdio>
// feasibility: per-scene block writes its 50k float4 records (a) as 16-B stores to pseudo-random slots,
// (b) as SEG-record segments (SEG*16 B contiguous) at pseudo-random segment slots; reads are coalesced.
template <int SEG>
__global__ __launch_bounds__(512) void seg_kernel(const float *__restrict__ raw, float4 *__restrict__ out, int n) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *src = raw + (size_t)s * n * 3;
    float4 *dst = out + (size_t)s * n;
    const int nseg = n / SEG;
    for (int p0 = tid; p0 < nseg * SEG; p0 += 2048) {
        float v[4][3]; int pos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int p = p0 + u * 512; p = p < nseg * SEG ? p : p0;
            v[u][0] = src[3 * p]; v[u][1] = src[3 * p + 1]; v[u][2] = src[3 * p + 2];
            const int seg = p / SEG, off = p % SEG;
            pos[u] = (int)(((long long)seg * 40503ll + 977ll) % nseg) * SEG + off;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { int p = p0 + u * 512; if (p < nseg * SEG) dst[pos[u]] = make_float4(v[u][0], v[u][1], v[u][2], 0.f); }
    }
}
template <int SEG> void run(const float *raw, float4 *out, int S, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(seg_kernel<SEG>, dim3(S), dim3(512), 0, 0, raw, out, n);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(seg_kernel<SEG>, dim3(S), dim3(512), 0, 0, raw, out, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("segment of %4d records (%5d B): %.1f us per launch\n", SEG, SEG * 16, ms * 100.f);
}
int main() {
    const int S = 256, n = 50000;
    float *raw; float4 *out;
    (void)hipMalloc(&raw, sizeof(float) * 3 * n * S); (void)hipMalloc(&out, sizeof(float4) * n * S);
    (void)hipMemset(raw, 0, sizeof(float) * 3 * n * S);
    run<1>(raw, out, S, n); run<4>(raw, out, S, n); run<8>(raw, out, S, n); run<16>(raw, out, S, n); run<32>(raw, out, S, n); run<64>(raw, out, S, n); run<1024>(raw, out, S, n);
    return 0;
}
