"""Diagnostics (build with AMK_HIPCC_FLAGS=-DAMK_KNN_COUNT): what one K-NN query of the bench's control step does -- rings, item batches
(rows x tiles, 64 per batch), candidate batches (64 records each), candidates fetched, insertions into the top-k list."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from avoid_mpc_amd import synth, fsm, capi
from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch
lib = capi.load()
S, n = 256, 50000
prm = synth.MpcParams(T=0.66, K=8); dev = torch.device('cuda'); N = prm.N
clouds, edges = synth.make_clouds_torch(n, S, 100000, dev)
sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
for s in range(S):
    pos, vel, acc, yaw = synth.make_odom(100000 + s, prm)
    sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
kd_o, kd_e = KdBatch(S, n), KdBatch(S, n // 10); kd_o.build(clouds); kd_e.build(edges)
mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
h = (C.c_ulonglong * 8)()
lib.amk__knn_counters.argtypes = [C.c_void_p, C.c_int]
lib.amk__knn_counters(h, 1)
ref = torch.from_numpy(ref0).to(dev)
step_batch(kd_o, kd_e, mpc, prm, torch.from_numpy(sq).to(dev), torch.from_numpy(posx).to(dev), ref)
lib.amk__knn_counters(h, 0)
q = h[0]
print("queries %d (incl. the edge 1-NN and the snap re-queries): per query rings %.2f, item batches %.2f, candidate batches %.2f, candidates %.1f, insertions %.2f"
      % (q, 1 + h[5] / q, h[1] / q, h[2] / q, h[4] / q, h[3] / q))
