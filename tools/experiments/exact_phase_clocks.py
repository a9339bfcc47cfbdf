"""Per-wavefront phase clocks of the lower-level exact-tree build (kd_exact_build_kernel), scene 0, 256 x 50k points: time in
register-resident subtrees popped directly, in LDS windows, in wavefront-level splits in global memory, and the kernel length.
Needs a diagnostics build: AMK_HIPCC_FLAGS=-DAMK_EXACT_TRACE=1 python -m avoid_mpc_amd.build --force."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import KdBatch
S, n = 256, 50000
base = torch.from_numpy(synth.make_cloud(n, 7)[0]).cuda()
cl = torch.stack([base[torch.randperm(n, device="cuda")] for _ in range(S)]).contiguous()
kd = KdBatch(S, n); kd.set_tie_order(1)
for _ in range(3): kd.build(cl)
torch.cuda.synchronize()
lib = capi.load()
buf = np.zeros(256, np.uint32)
lib.amk__kd_exact_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.amk__kd_exact_trace(kd.h, buf.ctypes.data_as(C.c_void_p), 256) == 0
b = buf.reshape(16, 16)[:, :9].astype(np.float64)
print("per wave (us): sub  n   win  n   mid  n   loop_end  last_work  kernel")
for w in range(16):
    print(w, " ".join(f"{v/100 if i in (0,2,4,6,7,8) else v:8.1f}" for i, v in enumerate(b[w])))
print("mean us: sub %.1f win %.1f mid %.1f busy %.1f of %.1f" % (b[:,0].mean()/100, b[:,2].mean()/100, b[:,4].mean()/100, (b[:,0]+b[:,2]+b[:,4]).mean()/100, b[:,8].mean()/100))
