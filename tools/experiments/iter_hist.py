"""Distribution of interior-point iterations per solve launch in the bench workload (256 scenes, C2 size): the launch
lasts as long as its slowest scene, so the tail of this distribution is what a stream waits for."""
import sys
sys.path.insert(0, '.')
import copy
import numpy as np, torch
from avoid_mpc_amd import synth, fsm
from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch
S, n = 256, 50000
prm = synth.MpcParams(T=0.66, K=8)
dev = torch.device('cuda')
clouds, edges = synth.make_clouds_torch(n, S, 100000, dev)
sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, prm.N, 10)); posx = np.zeros(S)
for s in range(S):
    pos, vel, acc, yaw = synth.make_odom(100000 + s, prm)
    sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
kd_o, kd_e = KdBatch(S, n), KdBatch(S, n // 10)
kd_o.build(clouds); kd_e.build(edges)
tot = []
for passes in (1, 2, 3):
    p = copy.copy(prm); p.max_iter = passes
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    ref = torch.from_numpy(ref0).to(dev)
    out = step_batch(kd_o, kd_e, mpc, p, torch.from_numpy(sq[:, :passes].copy()).to(dev), torch.from_numpy(posx).to(dev), ref)
    torch.cuda.synchronize()
    tot.append(out["flags"].cpu().numpy().copy())
prev_it, prev_sol = np.zeros(S, int), np.zeros(S, int)
for i, f in enumerate(tot):
    ran = f[:, 1] - prev_sol
    its = (f[:, 3] - prev_it)[ran > 0]
    print(f"pass {i + 1}: scenes solving {int(ran.sum())}/{S}  iterations mean {its.mean():.1f} median {np.median(its):.0f} "
          f"p90 {np.percentile(its, 90):.0f} max {its.max()}  hist(<=10,<=15,<=20,<=25,<=30,<=39,40): "
          f"{[int(((its > a) & (its <= b)).sum()) for a, b in ((0,10),(10,15),(15,20),(20,25),(25,30),(30,39),(39,40))]}")
    prev_it, prev_sol = f[:, 3].copy(), f[:, 1].copy()
