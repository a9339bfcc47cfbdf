import sys
sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import KdBatch
S,n=256,50000
lib=capi.load()
base=synth.make_cloud(n,7)[0]
def run(order):
    pts=base.copy()
    if order=='random': pass
    elif order=='depth-image (azimuth/elevation scanlines)':
        az=np.arctan2(pts[:,1],pts[:,0]+1e-3); el=np.arctan2(pts[:,2]-1.5,np.hypot(pts[:,0],pts[:,1])+1e-3)
        rows=np.floor((el-el.min())/(el.max()-el.min()+1e-9)*200).astype(int)
        pts=pts[np.lexsort((az,rows))]
    elif order=='sorted by cell-ish (x,y,z lexicographic on 0.7 m grid)':
        g=np.floor(pts/0.7).astype(int); pts=pts[np.lexsort((g[:,0],g[:,1],g[:,2]))]
    cl=torch.from_numpy(np.repeat(pts[None],S,0).copy()).cuda()
    kd=KdBatch(S,n)
    for _ in range(3): kd.build(cl)
    torch.cuda.synchronize()
    lib.amk__timing_enable(2)
    for _ in range(10): kd.build(cl)
    torch.cuda.synchronize()
    ms=(C.c_double*8)(); cnt=(C.c_int*8)(); lib.amk__timing_collect(ms,cnt); lib.amk__timing_enable(0)
    print(f'{order}: build {ms[7]/cnt[7]*1e3:.1f} us per 256-scene launch')
for o in ('random','depth-image (azimuth/elevation scanlines)','sorted by cell-ish (x,y,z lexicographic on 0.7 m grid)'): run(o)
