"""Does a live RCCL communicator slow the pipeline down?  Steady-state steps/s of the bench workload (amk_pipeline_*, 20 slots)
with (a) no communicator, (b) the library's own (amk_shard_create, world 1), (c) torch.distributed's NCCL process group, (d) both.
usage: python tools/experiments/rccl_presence.py [a|b|c|d]   (one case per process)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, ".")
import numpy as np, torch
from avoid_mpc_amd import fsm, synth
from avoid_mpc_amd.host import Pipeline, Shard
case = sys.argv[1] if len(sys.argv) > 1 else "a"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if case in "cd":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sh = Shard(0, 1, Shard.unique_id()) if case in "bde" else None
if case == "e":   # created and destroyed again before anything runs
    sh.close(); sh = None
if case == "h":   # no RCCL: one HIGH-PRIORITY stream that has run something (RCCL creates its streams with a priority)
    hp = torch.cuda.Stream(device=dev, priority=-1)
    with torch.cuda.stream(hp):
        torch.zeros(1024, device=dev).add_(1)
    torch.cuda.synchronize()
if case == "i":   # no RCCL: mapped pinned host memory + a host function on a stream (what a proxy FIFO needs)
    pin = torch.zeros(1 << 20, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()
prm = synth.MpcParams(T=0.66, K=8); S, n, ne, N, nslots = 256, 50000, 5000, prm.N, 20
pl = Pipeline(nslots, S, n, ne, prm, queue_depth=int(os.environ.get("QD", "0")))
frames = []
for i in range(nslots):
    seed = 100000 + i * S
    cl, ed = synth.make_clouds_torch(n, S, seed, dev)
    sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
    for s in range(S):
        pos, vel, acc, yaw = synth.make_odom(seed + s, prm)
        sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
    frames.append((cl, ed, torch.from_numpy(sq).to(dev), torch.from_numpy(posx).to(dev), torch.from_numpy(ref0).to(dev)))
def run(steps):
    t0 = time.perf_counter()
    for j in range(steps):
        f = frames[j % nslots]
        pl.submit(f[0], f[1], f[2], f[3], f[4])
    pl.drain(); torch.cuda.synchronize()
    return time.perf_counter() - t0
run(40)
t = run(1024)
print(f"case {case}: {S * 1024 / t:.0f} steps/s (GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']})", flush=True)
def kfd_queues():
    d = f"/sys/class/kfd/kfd/proc/{os.getpid()}/queues"
    out = []
    try:
        for q in sorted(os.listdir(d), key=int):
            rd = lambda n: open(os.path.join(d, q, n)).read().strip() if os.path.exists(os.path.join(d, q, n)) else "?"
            out.append((q, rd("type"), rd("size"), rd("gpuid")))
    except Exception as e:
        return "n/a: %r" % (e,)
    return out
qs = kfd_queues()
print(f"case {case}: KFD queues {len(qs) if isinstance(qs, list) else qs}: {qs}", flush=True)
import ctypes as C
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
fl = C.c_uint(); hip.hipGetDeviceFlags(C.byref(fl)); print(f"case {case}: hipGetDeviceFlags = {fl.value:#x}", flush=True)
import threading
print(f"case {case}: threads in process = {len(os.listdir('/proc/self/task'))}", flush=True)
v = C.c_size_t()
for name, lim in (("stack", 0), ("printf", 1), ("malloc_heap", 2)):
    hip.hipDeviceGetLimit(C.byref(v), lim); print(f"case {case}: hipLimit {name} = {v.value}", flush=True)
if case == "g":   # the communicator appears after the pipeline's streams have their hardware queues
    sh = Shard(0, 1, Shard.unique_id())
    t = run(1024)
    print(f"case g, after amk_shard_create: {S * 1024 / t:.0f} steps/s", flush=True)
    sh.close(); sh = None
    t = run(1024)
    print(f"case g, after amk_shard_destroy: {S * 1024 / t:.0f} steps/s", flush=True)
