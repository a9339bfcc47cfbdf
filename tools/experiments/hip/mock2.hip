#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// In-flight cost model of the index build: 16 streams, each runs the memory traffic of one 256-scene build.
//  current  : 2 streaming reads of raw (compaction + histogram stand-ins) + SoA write + read + random 16-B scatter
//  two-level: 2 streaming reads + partition into SEG-record segments (coarse bins) + per-bin pass (read temp, write final)
constexpr int T = 512;
__global__ __launch_bounds__(T) void read_pass(const float *__restrict__ raw, float *__restrict__ sink, int n, int wr) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *src = raw + (size_t)s * n * 3;
    float acc = 0.f;
    for (int i0 = tid; i0 < n; i0 += 4 * T) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) { int i = i0 + u * T; i = i < n ? i : i0; v[u][0] = src[3 * i]; v[u][1] = src[3 * i + 1]; v[u][2] = src[3 * i + 2]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc += v[u][0] + v[u][1] + v[u][2];
            int i = i0 + u * T;
            if (wr && i < n) { float *d = sink + (size_t)s * n * 3; d[i] = v[u][0]; d[n + i] = v[u][1]; d[2 * n + i] = v[u][2]; }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int SEG>
__global__ __launch_bounds__(T) void seg_scatter(const float *__restrict__ raw, float4 *__restrict__ out, int n) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *src = raw + (size_t)s * n * 3;
    float4 *dst = out + (size_t)s * n;
    const int nseg = n / SEG;
    for (int p0 = tid; p0 < nseg * SEG; p0 += 4 * T) {
        float v[4][3]; int pos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int p = p0 + u * T; p = p < nseg * SEG ? p : p0;
            v[u][0] = src[3 * p]; v[u][1] = src[3 * p + 1]; v[u][2] = src[3 * p + 2];
            pos[u] = (int)(((long long)(p / SEG) * 40503ll + 977ll) % nseg) * SEG + p % SEG;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { int p = p0 + u * T; if (p < nseg * SEG) dst[pos[u]] = make_float4(v[u][0], v[u][1], v[u][2], 0.f); }
    }
}
__global__ __launch_bounds__(T) void copy4(const float4 *__restrict__ a, float4 *__restrict__ b, int n) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const float4 *src = a + (size_t)s * n; float4 *dst = b + (size_t)s * n;
    for (int i0 = tid; i0 < n; i0 += 4 * T) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { int i = i0 + u * T; v[u] = src[i < n ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { int i = i0 + u * T; if (i < n) dst[i] = v[u]; }
    }
}
int main() {
    const int S = 256, n = 50000, NS = 16, REP = 6;
    std::vector<float *> raw(NS), soa(NS); std::vector<float4 *> tmp(NS), fin(NS); std::vector<hipStream_t> st(NS);
    for (int i = 0; i < NS; ++i) {
        (void)hipMalloc(&raw[i], sizeof(float) * 3 * n * S); (void)hipMalloc(&soa[i], sizeof(float) * 3 * n * S);
        (void)hipMalloc(&tmp[i], sizeof(float4) * n * S); (void)hipMalloc(&fin[i], sizeof(float4) * n * S);
        (void)hipMemset(raw[i], 0, sizeof(float) * 3 * n * S); (void)hipStreamCreate(&st[i]);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = -1; rep < REP; ++rep) {
            if (rep == 0) { (void)hipDeviceSynchronize(); (void)hipEventRecord(e0, 0); (void)hipStreamSynchronize(0); }
            for (int i = 0; i < NS; ++i) {
                if (mode == 0) {          // current: compaction (R raw, W SoA), histogram (R SoA), scatter (R SoA, random 16-B W)
                    hipLaunchKernelGGL(read_pass, dim3(S), dim3(T), 0, st[i], raw[i], soa[i], n, 1);
                    hipLaunchKernelGGL(read_pass, dim3(S), dim3(T), 0, st[i], soa[i], soa[i], n, 0);
                    hipLaunchKernelGGL(seg_scatter<1>, dim3(S), dim3(T), 0, st[i], soa[i], fin[i], n);
                } else if (mode == 1) {   // two-level on top of the SoA compaction
                    hipLaunchKernelGGL(read_pass, dim3(S), dim3(T), 0, st[i], raw[i], soa[i], n, 1);
                    hipLaunchKernelGGL(read_pass, dim3(S), dim3(T), 0, st[i], soa[i], soa[i], n, 0);
                    hipLaunchKernelGGL(seg_scatter<16>, dim3(S), dim3(T), 0, st[i], soa[i], tmp[i], n);
                    hipLaunchKernelGGL(copy4, dim3(S), dim3(T), 0, st[i], tmp[i], fin[i], n);
                } else {                  // two-level, compaction fused (no SoA): bbox pass, histogram pass, partition, per-bin pass
                    hipLaunchKernelGGL(read_pass, dim3(S), dim3(T), 0, st[i], raw[i], soa[i], n, 0);
                    hipLaunchKernelGGL(read_pass, dim3(S), dim3(T), 0, st[i], raw[i], soa[i], n, 0);
                    hipLaunchKernelGGL(seg_scatter<16>, dim3(S), dim3(T), 0, st[i], raw[i], tmp[i], n);
                    hipLaunchKernelGGL(copy4, dim3(S), dim3(T), 0, st[i], tmp[i], fin[i], n);
                }
            }
        }
        (void)hipDeviceSynchronize(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const char *names[3] = {"current traffic pattern", "two-level after SoA compaction", "two-level, compaction fused"};
        printf("%-34s: %.1f us per 256-scene build with %d builds in flight\n", names[mode], ms * 1000.f / (REP * NS), NS);
    }
    return 0;
}
