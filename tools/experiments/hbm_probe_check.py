import sys, ctypes as C
sys.path.insert(0, '.')
import torch
from avoid_mpc_amd import capi
lib = capi.load()
lib.amk__hbm_copy_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
for gib in (1.0, 4.0):
    nb = int(gib * (1 << 30))
    a = torch.empty(nb, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a); a.fill_(1); torch.cuda.synchronize()
    ms = C.c_double(); v = C.c_int()
    print(lib.amk__hbm_copy_probe(a.data_ptr(), b.data_ptr(), nb, 10, None, C.byref(ms), C.byref(v)), gib, "GiB:", 2 * nb / (ms.value * 1e-3) / 1e9, "GB/s, variant", v.value)
    import time; t0=time.perf_counter(); b.copy_(a); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): b.copy_(a)
    torch.cuda.synchronize(); print("  torch copy_:", 2*nb*10/(time.perf_counter()-t0)/1e9, "GB/s")
    del a, b
