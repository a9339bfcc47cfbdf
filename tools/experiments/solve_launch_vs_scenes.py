import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import MpcBatch
prm=synth.MpcParams()
logs=_scene_inputs(20000,[200],prm)
for S in (1,64,256,1024,1536,3072):
    ref=torch.from_numpy(np.repeat(logs[0][0][None],S,0)).cuda()
    gpu=MpcBatch(prm.T,prm.dt,prm.K,S); gpu.configure(prm)
    ts=[]
    for rep in range(6):
        gpu.reset_warm_start(); torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); u,x0,info=gpu.Solve(ref, want_traj=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)*1e3)
    print('S',S,'solve launch us', ' '.join('%.0f'%t for t in ts), 'iters', info.cpu().numpy()[0])
