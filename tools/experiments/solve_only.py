import sys
sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth
from avoid_mpc_amd.host import MpcBatch
S=int(sys.argv[1]); reps=int(sys.argv[2]) if len(sys.argv)>2 else 4
prm=synth.MpcParams()
logs=_scene_inputs(20000,[200,201,202,203],prm)
refs=np.stack([logs[i%4][0] for i in range(S)])
ref=torch.from_numpy(refs).cuda()
gpu=MpcBatch(prm.T,prm.dt,prm.K,S); gpu.configure(prm)
for rep in range(reps):
    gpu.reset_warm_start(); u,x0,info=gpu.Solve(ref, want_traj=False)
torch.cuda.synchronize()
print('iters', info.cpu().numpy()[:4,1])
