#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// feasibility: per-scene block gathers raw 12-B points in a pseudo-random order and writes float4 coalesced
__global__ __launch_bounds__(1024) void gather_kernel(const float *__restrict__ raw, float4 *__restrict__ out, int n, int mode) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const float *src = raw + (size_t)s * n * 3;
    float4 *dst = out + (size_t)s * n;
    if (mode == 0) {  // streaming passes first (bbox + hist pass stand-ins): read everything twice
        float acc = 0.f;
        for (int rep = 0; rep < 2; ++rep)
            for (int i0 = tid; i0 < n; i0 += 4096) {
                float v[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) { int i = i0 + u * 1024; i = i < n ? i : i0; v[u][0] = src[3 * i]; v[u][1] = src[3 * i + 1]; v[u][2] = src[3 * i + 2]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += v[u][0] + v[u][1] + v[u][2];
            }
        if (acc == 12345.678f) dst[0] = make_float4(acc, 0, 0, 0);
    }
    for (int p0 = tid; p0 < n; p0 += 4096) {
        float v[4][3]; int idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int p = p0 + u * 1024; p = p < n ? p : p0;
            const int i = (int)(((long long)p * 40503ll + 12345ll) % n);  // pseudo-random permutation stand-in
            idx[u] = i;
            v[u][0] = src[3 * i]; v[u][1] = src[3 * i + 1]; v[u][2] = src[3 * i + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { int p = p0 + u * 1024; if (p < n) dst[p] = make_float4(v[u][0], v[u][1], v[u][2], __int_as_float(idx[u])); }
    }
}
int main() {
    const int S = 256, n = 50000;
    float *raw; float4 *out;
    hipMalloc(&raw, sizeof(float) * 3 * n * S); hipMalloc(&out, sizeof(float4) * n * S);
    hipMemset(raw, 0, sizeof(float) * 3 * n * S);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gather_kernel, dim3(S), dim3(1024), 0, 0, raw, out, n, mode);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(gather_kernel, dim3(S), dim3(1024), 0, 0, raw, out, n, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.1f us per launch\n", mode, mode == 0 ? "2 streaming read passes + gather/write" : "gather/write only", ms * 100.f);
    }
    return 0;
}
