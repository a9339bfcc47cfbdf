"""Solve-only throughput (16 streams x 256 scenes, as ms_parts.py) while a resource hog runs on another stream: tells which
CU resource the solves are short of.  Run on the GPU box: builds tools/experiments/hip/hogs.hip with hipcc first."""
import os, sys, time, subprocess, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, '.')
import numpy as np, torch
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import MpcBatch
so = "/tmp/libhogs.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "tools/experiments/hip/hogs.hip", "-o", so])
hogs = C.CDLL(so)
S, NS = 256, 16
prm = synth.MpcParams(T=0.66, K=8)
logs = _scene_inputs(20000, [200, 201, 202, 203], prm)
ref = torch.from_numpy(np.stack([logs[i % 4][0] for i in range(S)])).cuda()
streams = [torch.cuda.Stream() for _ in range(NS)]
mpcs = [MpcBatch(prm.T, prm.dt, prm.K, S) for _ in range(NS)]
for m in mpcs: m.configure(prm)
outs = [(torch.empty((S, 4), dtype=torch.float64, device='cuda'), torch.empty((S, 4), dtype=torch.int32, device='cuda')) for _ in range(NS)]
lib = capi.load()
def solve(i):
    m, st = mpcs[i], streams[i]
    m.reset_warm_start(st)
    capi.check(lib.amk_mpc_solve(m.h, capi.dptr(ref), capi.dptr(outs[i][0]), None, capi.dptr(outs[i][1]), 0, capi.stream_ptr(st)), 's')
hog_stream = torch.cuda.Stream()
sink = torch.zeros(8, dtype=torch.float64, device='cuda')
big = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); big2 = torch.empty_like(big)
def run(hog, reps=8):
    for i in range(NS): solve(i)
    torch.cuda.synchronize()
    if hog: hog()
    t0 = time.perf_counter()
    for r in range(reps):
        for i in range(NS): solve(i)
    for s in streams: s.synchronize()
    t = (time.perf_counter() - t0) / (reps * NS)
    torch.cuda.synchronize()
    return S / (t * 1e6)
sp = C.c_void_p(hog_stream.cuda_stream)
print("no hog:                 %.2f solves/us" % run(None))
for w in (1, 2, 4):
    print("lds hog, %d waves/CU:    %.2f solves/us" % (w, run(lambda: hogs.launch_lds_hog(256 * w, 400000, C.c_void_p(sink.data_ptr()), sp))))
for w in (1, 2, 4):
    print("valu hog, %d waves/CU:   %.2f solves/us" % (w, run(lambda: hogs.launch_valu_hog(256 * w, 200000, C.c_void_p(sink.data_ptr()), sp))))
for blocks in (512, 2048):   # the copy must outlast the measurement window (~15 ms): 256 MB x 400 at <= 5 TB/s = 40+ ms
    print("mem hog (copy 256 MB x400, %d blocks of 256): %.2f solves/us" % (blocks, run(lambda: hogs.launch_mem_hog(blocks, C.c_void_p(big.data_ptr()), C.c_void_p(big2.data_ptr()), C.c_size_t(big.numel() // 16), 400, sp))))
    torch.cuda.synchronize()
