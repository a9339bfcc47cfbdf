"""Round-2 style orchestration (torch streams + separate handles, no amk_pipeline) with / without a live RCCL communicator.
usage: python tools/experiments/rccl_presence_r2style.py [a|b]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, ".")
import numpy as np, torch
from avoid_mpc_amd import fsm, synth
from avoid_mpc_amd.host import KdBatch, MpcBatch, Shard, step_batch, kd_build_pair
case = sys.argv[1] if len(sys.argv) > 1 else "a"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sh = Shard(0, 1, Shard.unique_id()) if case == "b" else None
prm = synth.MpcParams(T=0.66, K=8); S, n, ne, N, nslots = 256, 50000, 5000, prm.N, 20
slots = []
for i in range(nslots):
    seed = 100000 + i * S
    cl, ed = synth.make_clouds_torch(n, S, seed, dev)
    sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
    for s in range(S):
        pos, vel, acc, yaw = synth.make_odom(seed + s, prm)
        sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
    m = MpcBatch(prm.T, prm.dt, prm.K, S); m.configure(prm)
    slots.append(dict(cl=cl, ed=ed, sq=torch.from_numpy(sq).to(dev), px=torch.from_numpy(posx).to(dev), ref0=torch.from_numpy(ref0).to(dev),
                      ref=torch.from_numpy(ref0).to(dev), st=torch.cuda.Stream(device=dev), ko=KdBatch(S, n), ke=KdBatch(S, ne), m=m,
                      out=dict(u=torch.empty((S, 4), dtype=torch.float64, device=dev), x0array=torch.empty((S, N, 14), dtype=torch.float64, device=dev),
                               flags=torch.empty((S, 4), dtype=torch.int32, device=dev))))
def run(steps):
    t0 = time.perf_counter()
    for j in range(steps):
        sl = slots[j % nslots]
        with torch.cuda.stream(sl["st"]):
            sl["ref"].copy_(sl["ref0"], non_blocking=True); sl["m"].reset_warm_start(sl["st"])
            kd_build_pair(sl["ko"], sl["cl"], sl["ke"], sl["ed"], stream=sl["st"])
            step_batch(sl["ko"], sl["ke"], sl["m"], prm, sl["sq"], sl["px"], sl["ref"], stream=sl["st"], out=sl["out"])
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    return time.perf_counter() - t0, te
run(40)
t, te = run(1024)
print(f"r2-style case {case}: {S * 1024 / t:.0f} steps/s; host enqueue {1e6 * te / 1024:.1f} us per step (14 launches)", flush=True)
