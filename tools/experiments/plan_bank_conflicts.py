import sys, ctypes as C
sys.path.insert(0,'.')
import numpy as np
from avoid_mpc_amd import capi, synth
from avoid_mpc_amd.host import MpcBatch
prm=synth.MpcParams(T=0.66,K=8)
m=MpcBatch(prm.T,prm.dt,prm.K,1); m.configure(prm)
lib=capi.load()
buf=(C.c_int*8192)()
lib.amk__plan_dump.restype=C.c_int
n=lib.amk__plan_dump(m.h, buf, 8192)
meta=np.array(buf[8:n])   # 8-int header: LDS offsets of P, p, lam, M, Hm, G, q and the total
items=meta[:128*13].reshape(128,13)
roles=meta[128*13:].reshape(64,16)
def cost_b64(addrs):
    # ds_read_b64: 2 groups of 32 lanes; bank = (addr_words) mod 64 where a double at index a occupies banks 2a, 2a+1 (mod 64)
    tot=0
    for g in range(2):
        a=addrs[32*g:32*g+32]
        banks={}
        for x in set(a.tolist()):
            b=(2*x)%64
            banks.setdefault(b,set()).add(x)
        tot+=max(len(v) for v in banks.values())
    return tot  # cycles-ish: ideal 2 (1 per group)
tot=0; ideal=0
for h in range(2):
    for t in range(9):
        addrs=items[h*64:(h+1)*64,t]
        c=cost_b64(addrs); tot+=c; ideal+=2
        print('item',h,'term',t,'cost',c, 'distinct', len(set(addrs.tolist())))
    for name,col in (('aux',11),('out',9)):
        c=cost_b64(items[h*64:(h+1)*64,col]); tot+=c; ideal+=2; print('item',h,name,'cost',c)
print('round A total',tot,'ideal',ideal)
tot2=0
for name,base,stride in (('gi',0,1),('gj',2,3)):
    for a in range(4):
        addrs=roles[:,base]+a*roles[:,stride]
        c=cost_b64(addrs); tot2+=c; print(name,a,'cost',c)
for k in range(3):
    c=cost_b64(roles[:,4+k]); tot2+=c; print('base',k,'cost',c)
for name,col in (('out1',10),('out2',11),('lam_src',13),('lam_dst',14)):
    c=cost_b64(roles[:,col]); tot2+=c; print(name,'cost',c)
print('round BC per-lane-indexed total',tot2)
