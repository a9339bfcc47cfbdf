"""Cost of the opt-in nanoflann tie-order mode (amk_kd_set_tie_order(AMK_TIES_NANOFLANN)) at the reference's frame size and at
BASELINE's: the extra build (the reference's own tree, kd_exact.h) and the 21-query search of a control-step pass, 256 scenes."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from avoid_mpc_amd import synth
from avoid_mpc_amd.host import KdBatch
S = 256
for n in (3072, 50000):
    base = torch.from_numpy(synth.make_cloud(n, 7)[0]).cuda()
    cl = torch.stack([base[torch.randperm(n, device="cuda")] for _ in range(S)]).contiguous()
    qs = torch.rand((S, 21, 3), dtype=torch.float64, device="cuda") * torch.tensor([20.0, 6.0, 3.0], dtype=torch.float64, device="cuda")
    out = {}
    for mode in (0, 1):
        kd = KdBatch(S, n); kd.set_tie_order(mode)
        kd.build(cl); r = kd.search(qs, 8); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): kd.build(cl)
        torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(10): r = kd.search(qs, 8, out=r)
        torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 10
        out[mode] = (tb, ts); kd.close()
    print(f"n = {n}: build {1e3 * out[0][0]:.3f} ms bucketed, {1e3 * out[1][0]:.3f} ms with the exact tree (+{1e3 * (out[1][0] - out[0][0]):.3f}); "
          f"21-query search {1e3 * out[0][1]:.3f} / {1e3 * out[1][1]:.3f} ms (+{1e3 * (out[1][1] - out[0][1]):.3f})", flush=True)
