import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import MpcBatch
prm=synth.MpcParams()
logs=_scene_inputs(20000,[200],prm)
S=int(sys.argv[1]) if len(sys.argv)>1 else 1
ref=np.repeat(logs[0][0][None],S,0)
gpu=MpcBatch(prm.T,prm.dt,prm.K,S); gpu.configure(prm)
tr=torch.zeros(16*64,dtype=torch.float64,device='cuda')
capi.load().amk__debug_trace(C.c_void_p(tr.data_ptr()))
for rep in range(2):
    gpu.reset_warm_start()
    u,x0,info=gpu.Solve(torch.from_numpy(ref).cuda()); torch.cuda.synchronize()
print('gpu info',info.cpu().numpy()[0])
t=tr.cpu().numpy().reshape(-1,16)
print('it  J err mu delta a | clk(100MHz ticks): eval backward forward stepcalc linesearch | nreg nls')
for i in range(10): print(i, ' '.join('%.4g'%v for v in t[i,:5]), '|', ' '.join('%.0f'%v for v in t[i,8:13]), '| eval: collide %d reduce %d stage %d'%(t[i,13], t[i,14], t[i,15]))
print('sum ticks', t[:10,8:13].sum(axis=0), 'total us (100MHz):', t[:10,8:13].sum()/100.)
