#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *r0, double *r1, double *r2, double *sq0, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r); r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r); r2[i] = r;
    sq0[i] = __builtin_amdgcn_sqrt(v);
}
int main() {
    const int n = 1 << 20;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n], *hs = new double[n];
    for (int i = 0; i < n; ++i) hx[i] = exp((rand() / (double)RAND_MAX - 0.5) * 40.0) * (1.0 + rand() / (double)RAND_MAX);
    double *dx, *d0, *d1, *d2, *ds;
    hipMalloc(&dx, 8 * n); hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n); hipMalloc(&ds, 8 * n);
    hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, ds, n);
    hipMemcpy(h0, d0, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 8 * n, hipMemcpyDeviceToHost);
    hipMemcpy(hs, ds, 8 * n, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, es = 0;
    for (int i = 0; i < n; ++i) {
        double t = 1.0 / hx[i];
        e0 = fmax(e0, fabs(h0[i] - t) / t); e1 = fmax(e1, fabs(h1[i] - t) / t); e2 = fmax(e2, fabs(h2[i] - t) / t);
        double s = sqrt(hx[i]); es = fmax(es, fabs(hs[i] - s) / s);
    }
    printf("max rel err: v_rcp_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e ; v_sqrt_f64 %.3e (eps = 1.1e-16)\n", e0, e1, e2, es);
    return 0;
}
