// LDS allocation granularity on gfx950: resident 64-thread blocks per CU against the dynamic-LDS request.
// hipcc --offload-arch=gfx950 -O2 lds_granule.hip -o lds_granule && ./lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) { extern __shared__ int sm[]; sm[threadIdx.x] = threadIdx.x; __syncthreads(); if (out) out[0] = sm[0]; }
int main() {
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int x : {5000, 5120, 5200, 5400, 5700, 6000, 6400, 6500, 19200, 19824, 20480, 20500, 23000}) {
        int nb = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, x);
        printf("lds %6d B -> %d blocks per CU  (163840 / %d = %.2f)\n", x, nb, x, 163840.0 / x);
    }
    return 0;
}
