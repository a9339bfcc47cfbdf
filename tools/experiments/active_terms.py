"""How many of the (N-1) K collision terms of a scene are live (softplus > 0 in double: clearance below 1.147 m) at the
solution of the bench workload: the derivative pass runs ceil(K / 3) rounds of 57 lanes whether 5 or 150 terms are live."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from avoid_mpc_amd import synth, fsm
from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch
S, n = 256, 50000
prm = synth.MpcParams(T=0.66, K=8); dev = torch.device('cuda'); N = prm.N
clouds, edges = synth.make_clouds_torch(n, S, 100000, dev)
sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
for s in range(S):
    pos, vel, acc, yaw = synth.make_odom(100000 + s, prm)
    sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
kd_o, kd_e = KdBatch(S, n), KdBatch(S, n // 10); kd_o.build(clouds); kd_e.build(edges)
mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
ref = torch.from_numpy(ref0).to(dev)
out = step_batch(kd_o, kd_e, mpc, prm, torch.from_numpy(sq).to(dev), torch.from_numpy(posx).to(dev), ref)
torch.cuda.synchronize()
X = out["x0array"][:, :, :3]                       # X_0 .. X_{N-1}; the terms sit on X_1 .. X_{N-1}
q = X[:, 1:, :].contiguous()
r = kd_o.search(q, prm.K)
rho = torch.sqrt(r["sqdist"])                      # [S, N-1, K]
live = (rho < prm.radius + 36.7 / 32.0).sum(dim=(1, 2)).cpu().numpy()
print("live terms per scene at the solution: mean %.1f median %d p90 %d max %d of %d; scenes with <= 64 live: %.0f %%, none live: %.0f %%"
      % (live.mean(), np.median(live), np.percentile(live, 90), live.max(), (N - 1) * prm.K, 100 * np.mean(live <= 64), 100 * np.mean(live == 0)))
per_round = [(rho[:, :, j0:j0 + 3] < prm.radius + 36.7 / 32.0).any(dim=(1, 2)).float().mean().item() for j0 in (0, 3, 6)]
print("fraction of scenes with a live term in round 0 / 1 / 2 (obstacle slots 0-2 / 3-5 / 6-7):", ["%.2f" % v for v in per_round])
