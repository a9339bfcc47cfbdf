import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES","24")
sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import MpcBatch, KdBatch
lib=capi.load()
S=256; n=50000; NS=int(sys.argv[1]) if len(sys.argv)>1 else 8
prm=synth.MpcParams(T=0.66,K=8)
logs=_scene_inputs(20000,[200,201,202,203],prm)
ref=torch.from_numpy(np.stack([logs[i%4][0] for i in range(S)])).cuda()
cl=torch.empty((S,n,3),dtype=torch.float32,device='cuda')
base=torch.from_numpy(synth.make_cloud(n,7)[0]).cuda()
for s in range(S): cl[s]=base[torch.randperm(n,device='cuda')]
sa=[torch.cuda.Stream() for _ in range(NS)]; sb=[torch.cuda.Stream() for _ in range(NS)]
mpcs=[MpcBatch(prm.T,prm.dt,prm.K,S) for _ in range(NS)]
for m in mpcs: m.configure(prm)
kds=[KdBatch(S,n) for _ in range(NS)]
outs=[(torch.empty((S,4),dtype=torch.float64,device='cuda'),torch.empty((S,4),dtype=torch.int32,device='cuda')) for _ in range(NS)]
q=torch.from_numpy(np.random.default_rng(0).uniform([0,-8,0],[30,8,4],(S,21,3))).cuda()
kouts=[None]*NS
def solve(i):
    m=mpcs[i]; st=sa[i]
    m.reset_warm_start(st)
    capi.check(lib.amk_mpc_solve(m.h, capi.dptr(ref), capi.dptr(outs[i][0]), None, capi.dptr(outs[i][1]), 0, capi.stream_ptr(st)),'s')
def build(i): kds[i].build(cl, stream=sb[i])
def knn(i): kouts[i]=kds[i].search(q,8,stream=sb[i],out=kouts[i])
def run(fa, ra, fb, rb):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for r in range(max(ra,rb)):
        for i in range(NS):
            if fa and r<ra: fa(i)
            if fb and r<rb: fb(i)
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)*1e3
for i in range(NS): solve(i); build(i); knn(i)
torch.cuda.synchronize()
ts=run(solve,30,None,0); tb=run(None,0,build,10); tk=run(None,0,knn,30)
print(f'alone: 30x{NS} solves {ts:.1f} ms ({ts*1e3/30/NS:.0f} us each); 10x{NS} builds {tb:.1f} ms ({tb*1e3/10/NS:.0f} us each); 30x{NS} knn {tk:.1f} ms ({tk*1e3/30/NS:.0f} us each)')
t=run(solve,30,build,10); print(f'solves + builds on separate streams: {t:.1f} ms (sum {ts+tb:.1f}, max {max(ts,tb):.1f})')
# when does each class finish inside the combined run?  (events on the last kernel of every stream)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e0.record()
for r in range(30):
    for i in range(NS):
        solve(i)
        if r<10: build(i)
ea=[torch.cuda.Event(enable_timing=True) for _ in range(NS)]; eb=[torch.cuda.Event(enable_timing=True) for _ in range(NS)]
for i in range(NS): ea[i].record(sa[i]); eb[i].record(sb[i])
torch.cuda.synchronize()
print(f'  inside the combined run: builds done after {max(e0.elapsed_time(x) for x in eb):.1f} ms (alone {tb:.1f}), solves after {max(e0.elapsed_time(x) for x in ea):.1f} ms (alone {ts:.1f})')
t=run(solve,30,knn,30); print(f'solves + knn on separate streams: {t:.1f} ms (sum {ts+tk:.1f}, max {max(ts,tk):.1f})')
sa2=sa; 
def knn_a(i): kouts[i]=kds[i].search(q,8,stream=sa[i],out=kouts[i])
tk2=run(knn_a,30,None,0)
t=run(knn_a,30,build,10); print(f'knn (streams A) + builds (streams B): {t:.1f} ms (sum {tk2+tb:.1f}, max {max(tk2,tb):.1f})')
