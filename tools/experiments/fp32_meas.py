import sys
sys.path.insert(0,'.')
import numpy as np, torch
from tests import _oracle
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth
from avoid_mpc_amd.host import MpcBatch
c=synth.CONFIGS['C5']
seeds=list(range(200,232))
for lam_scale,max_iter in ((1.0,10),(1.0,60),(0.0,10),(0.0,60)):
    prm=synth.MpcParams(T=c['T'],K=c['K'])
    logs=_scene_inputs(20000,seeds,prm)     # inputs of the nominal problem
    w=np.array(prm.weights,float).copy(); w[24]*=lam_scale; prm.weights=w
    S=len(seeds); ref=np.stack([l[0] for l in logs])
    res={}
    for bits in (64,32):
        g=MpcBatch(prm.T,prm.dt,prm.K,S); g.configure(prm); g.set_solver_options(1e-4,max_iter); g.set_precision(bits)
        u,x0,info=g.Solve(torch.from_numpy(ref).cuda()); torch.cuda.synchronize()
        res[bits]=(u.cpu().numpy(),g.get_warm_start().cpu().numpy(),info.cpu().numpy())
    lib=_oracle.load_oracle()
    N=prm.N;K=prm.K
    tail=np.concatenate([prm.gain,prm.tau,prm.weights,[prm.radius]])
    J={b:np.array([lib.mpco_nlp_f(np.ascontiguousarray(res[b][1][s]),np.ascontiguousarray(np.concatenate([ref[s],tail])),N,K) for s in range(S)]) for b in (64,32)}
    du=np.abs(res[64][0]-res[32][0]).max(axis=1); dw=np.abs(res[64][1]-res[32][1]).max(axis=1)
    rel=(J[32]-J[64])/np.abs(J[64])
    print(f'lambda x{lam_scale} max_iter {max_iter}: |du| median {np.median(du):.2e} p90 {np.quantile(du,0.9):.2e} max {du.max():.2e}; |dw| median {np.median(dw):.2e} max {dw.max():.2e}; (J32-J64)/J64 median {np.median(rel):.2e} p90 {np.quantile(rel,0.9):.2e} max {rel.max():.2e} min {rel.min():.2e}; iters64 {res[64][2][:,1].mean():.1f} iters32 {res[32][2][:,1].mean():.1f} status32 {np.bincount(res[32][2][:,0],minlength=3)}')
