import sys
sys.path.insert(0,'.')
import numpy as np, torch
from avoid_mpc_amd import synth
from avoid_mpc_amd.host import KdBatch
S,n=256,50000
cl=torch.empty((S,n,3),dtype=torch.float32,device='cuda')
base=torch.from_numpy(synth.make_cloud(n,7)[0]).cuda()
for s in range(S): cl[s]=base[torch.randperm(n,device='cuda')]
kd=KdBatch(S,n)
from avoid_mpc_amd import capi
lib=capi.load()
import ctypes as C
for rep in range(3): kd.build(cl)
torch.cuda.synchronize()
lib.amk__timing_enable(2)
for rep in range(10): kd.build(cl)
torch.cuda.synchronize()
ms=(C.c_double*8)(); cnt=(C.c_int*8)(); lib.amk__timing_collect(ms,cnt); lib.amk__timing_enable(0)
print('build us %.1f'%(ms[7]/cnt[7]*1e3))
