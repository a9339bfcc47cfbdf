"""Saturated solve-only rate (16 streams x 256 scenes, zero warm start) at a horizon / K given by the environment:
AMK_T (0.33 -> N = 10, 0.66 -> N = 20, 1.0 -> N = 30), AMK_K, AMK_PREC.  Used by third_wave_fp64_n10.sh."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, '.')
import numpy as np, torch
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import MpcBatch
S = 256; NS = 16
prm = synth.MpcParams(T=float(os.environ.get("AMK_T", "0.66")), K=int(os.environ.get("AMK_K", "8")))
logs = _scene_inputs(20000, [200, 201, 202, 203], prm)
refs = np.stack([logs[i % 4][0] for i in range(S)])
ref = torch.from_numpy(refs).cuda()
streams = [torch.cuda.Stream() for _ in range(NS)]
mpcs = [MpcBatch(prm.T, prm.dt, prm.K, S) for _ in range(NS)]
for m in mpcs: m.configure(prm); m.set_precision(int(os.environ.get("AMK_PREC", "64")))
outs = [(torch.empty((S, 4), dtype=torch.float64, device='cuda'), torch.empty((S, 4), dtype=torch.int32, device='cuda')) for _ in range(NS)]
lib = capi.load()
def solve(i):
    m = mpcs[i]; st = streams[i]
    m.reset_warm_start(st)
    capi.check(lib.amk_mpc_solve(m.h, capi.dptr(ref), capi.dptr(outs[i][0]), None, capi.dptr(outs[i][1]), 0, capi.stream_ptr(st)), 's')
def run(fn, reps):
    for i in range(NS): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for i in range(NS): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * NS)
t = run(solve, int(os.environ.get('AMK_REPS', '8')))
info = outs[0][1].cpu().numpy()
print('N %d K %d: solve-only %.1f us per 256-scene launch -> %.2f solves/us; iterations mean %.1f, status>0: %d; resident solve blocks per CU %d'
      % (prm.N, prm.K, t * 1e6, S / (t * 1e6), info[:, 1].mean() if info.shape[1] > 1 else -1, int((info[:, 0] > 0).sum()), lib.amk__solve_occupancy(mpcs[0].h)))
