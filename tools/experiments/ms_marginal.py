import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES","24")
sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from tests.test_mpc_gpu import _scene_inputs
from avoid_mpc_amd import synth, capi, fsm
from avoid_mpc_amd.host import MpcBatch, KdBatch, step_batch
lib=capi.load()
S=256; n=50000; NS=16
prm=synth.MpcParams(T=0.66,K=8)
logs=_scene_inputs(20000,[200,201,202,203],prm)
ref=torch.from_numpy(np.stack([logs[i%4][0] for i in range(S)])).cuda()
cl=torch.empty((S,n,3),dtype=torch.float32,device='cuda'); ed=torch.empty((S,n//10,3),dtype=torch.float32,device='cuda')
sq=np.zeros((S,prm.max_iter,10)); ref0=np.zeros((S,prm.N,10)); posx=np.zeros(S)
for s in range(S):
    sc=synth.make_scene(n,100000+s,prm)
    cl[s]=torch.from_numpy(sc['cloud']).cuda(); ed[s]=torch.from_numpy(sc['edge']).cuda()
    sq[s]=fsm.state_quads(sc['pos'],sc['vel'],sc['acc'],sc['yaw'],prm.decay,prm.max_iter); ref0[s]=sc['ref_path']; posx[s]=sc['pos'][0]
sq_d=torch.from_numpy(sq).cuda(); ref0_d=torch.from_numpy(ref0).cuda(); posx_d=torch.from_numpy(posx).cuda()
st=[torch.cuda.Stream() for _ in range(NS)]
mpcs=[MpcBatch(prm.T,prm.dt,prm.K,S) for _ in range(NS)]
for m in mpcs: m.configure(prm)
kdo=[KdBatch(S,n) for _ in range(NS)]; kde=[KdBatch(S,n//10) for _ in range(NS)]
outs=[(torch.empty((S,4),dtype=torch.float64,device='cuda'),torch.empty((S,4),dtype=torch.int32,device='cuda')) for _ in range(NS)]
q=ref0_d[:,:,:3].contiguous(); q21=torch.cat([q,q[:,:1]],1).contiguous()
kouts=[None]*NS; refs=[ref0_d.clone() for _ in range(NS)]; souts=[None]*NS
def solve(i):
    m=mpcs[i]; m.reset_warm_start(st[i])
    capi.check(lib.amk_mpc_solve(m.h, capi.dptr(ref), capi.dptr(outs[i][0]), None, capi.dptr(outs[i][1]), 0, capi.stream_ptr(st[i])),'s')
def build(i): kdo[i].build(cl,stream=st[i])
def build_e(i): kde[i].build(ed,stream=st[i])
def knn(i): kouts[i]=kdo[i].search(q21,8,stream=st[i],out=kouts[i])
def full(i):
    with torch.cuda.stream(st[i]):
        refs[i].copy_(ref0_d,non_blocking=True); mpcs[i].reset_warm_start(st[i])
        kdo[i].build(cl,stream=st[i]); kde[i].build(ed,stream=st[i])
        souts[i]=step_batch(kdo[i],kde[i],mpcs[i],prm,sq_d,posx_d,refs[i],stream=st[i],out=souts[i])
def run(fns, reps=12):
    for i in range(NS):
        for f in fns: f(i)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for r in range(reps):
        for i in range(NS):
            for f in fns: f(i)
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/(reps*NS)*1e6
for name,fns in (('3 solves',[solve]*3),('build_o',[build]),('build_o+build_e',[build,build_e]),('3 knn',[knn]*3),
                 ('builds + 3 solves',[build,build_e,solve,solve,solve]),('builds + 3x(knn,solve)',[build,build_e,knn,solve,knn,solve,knn,solve]),
                 ('full step',[full])):
    print(f'{name:28s}: {run(fns):7.1f} us per step-equivalent')
