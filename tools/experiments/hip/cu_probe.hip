// Which (XCC, SE, CU) a block lands on, for the CU-mask experiments (tools/experiments/cu_mask.py).
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC cu_probe.hip -o /tmp/libcuprobe.so
#include <hip/hip_runtime.h>
// s_getreg_b32 simm16 = (size - 1) << 11 | offset << 6 | id;  HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20 (gfx940+)
__global__ void cu_probe_kernel(unsigned *out, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}   // keep the block resident for a while so that the grid spreads over every allowed CU
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
extern "C" int cu_probe(unsigned *d_out, int blocks, int threads, int spin, void *stream) {
    hipLaunchKernelGGL(cu_probe_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, d_out, spin);
    return (int)hipGetLastError();
}
