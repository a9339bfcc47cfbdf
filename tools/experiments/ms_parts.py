import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES","24")
sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth
from avoid_mpc_amd.host import MpcBatch, KdBatch
S=256; NS=16; n=50000
prm=synth.MpcParams(T=0.66,K=8)
logs=_scene_inputs(20000,[200,201,202,203],prm)
refs=np.stack([logs[i%4][0] for i in range(S)])
ref=torch.from_numpy(refs).cuda()
cl=torch.empty((S,n,3),dtype=torch.float32,device='cuda')
base=torch.from_numpy(synth.make_cloud(n,7)[0]).cuda()
for s in range(S): cl[s]=base[torch.randperm(n,device='cuda')]
streams=[torch.cuda.Stream() for _ in range(NS)]
mpcs=[MpcBatch(prm.T,prm.dt,prm.K,S) for _ in range(NS)]
for m in mpcs: m.configure(prm); m.set_precision(int(os.environ.get("AMK_PREC","64")))
kds=[KdBatch(S,n) for _ in range(NS)]
outs=[(torch.empty((S,4),dtype=torch.float64,device='cuda'),torch.empty((S,4),dtype=torch.int32,device='cuda')) for _ in range(NS)]
import ctypes as C
from avoid_mpc_amd import capi
lib=capi.load()
def solve(i):
    m=mpcs[i]; st=streams[i]
    m.reset_warm_start(st)
    capi.check(lib.amk_mpc_solve(m.h, capi.dptr(ref), capi.dptr(outs[i][0]), None, capi.dptr(outs[i][1]), 0, capi.stream_ptr(st)),'s')
def build(i):
    kds[i].build(cl, stream=streams[i])
def run(fn, reps):
    for i in range(NS): fn(i)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for r in range(reps):
        for i in range(NS): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/(reps*NS)
# (round 6, last session: 256 repetitions, not 8 -- a 14 ms measurement ends with the ~2 ms tail of its last launches' slowest
# scenes and reads 2.40 solves/us where the sustained rate is 2.78: tools/experiments/clock_ramp.sh)
REPS=int(os.environ.get('AMK_REPS','256'))
t=run(solve,REPS); print('solve-only: %.1f us per 256-scene launch -> %.2f solves/us'%(t*1e6, S/(t*1e6)))
t=run(build,REPS); print('build-only: %.1f us per 256-scene build'%(t*1e6))
qs=torch.from_numpy(np.ascontiguousarray(refs[:,10:10+prm.N*10].reshape(S,prm.N,10)[:,:,:3])).cuda()
qs=torch.cat([qs,qs[:,:1]],1).contiguous()      # the 21 queries of a pass: N reference points + the edge query
souts=[None]*NS
def search(i):
    souts[i]=kds[i].search(qs,8,stream=streams[i],out=souts[i])
for i in range(NS): build(i)
t=run(search,REPS); print('search-only (21 queries x 256 scenes, K=8): %.1f us per launch'%(t*1e6))
def ss(i): search(i); solve(i)
t=run(ss,REPS); print('search+solve: %.1f us per pair'%(t*1e6))
def both(i): build(i); solve(i); solve(i); solve(i)
t=run(both,max(6,REPS//4)); print('build+3 solves: %.1f us per step-equivalent'%(t*1e6))
