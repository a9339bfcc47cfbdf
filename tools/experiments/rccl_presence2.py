"""Round 4: what a live RCCL communicator does to the pipeline, with the shipped configuration (10 slots x gang 4).
usage: python tools/experiments/rccl_presence2.py <case> [nslots] [gang] [queue_depth]
  a  no communicator                      b  amk_shard_create (world 1) before the pipeline
  e  created AND destroyed before         k  communicator + its first collective run (channels set up), then the pipeline
Prints steps/s (steady, 1024 steps), the 20-step burst, KFD queue count, thread count, and an event round-trip microbenchmark
(tiny kernel -> hipEventRecord -> hipEventSynchronize, 2000 times) -- host wake-up latency with and without the communicator."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, ".")
import numpy as np, torch
from avoid_mpc_amd import fsm, synth
from avoid_mpc_amd.host import Pipeline, Shard
case = sys.argv[1] if len(sys.argv) > 1 else "a"
nslots = int(sys.argv[2]) if len(sys.argv) > 2 else 10
gang = int(sys.argv[3]) if len(sys.argv) > 3 else 4
qd = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
torch.zeros(1, device=dev)
def nq():
    d = f"/sys/class/kfd/kfd/proc/{os.getpid()}/queues"
    try:
        return len(os.listdir(d))
    except Exception:
        return -1
def evt_roundtrip(tag):
    st = torch.cuda.Stream(device=dev); x = torch.zeros(64, device=dev)
    ev = torch.cuda.Event()
    ts = []
    with torch.cuda.stream(st):
        for i in range(2200):
            t0 = time.perf_counter()
            x.add_(1.0); ev.record(st); ev.synchronize()
            ts.append(time.perf_counter() - t0)
    ts = np.array(ts[200:]) * 1e6
    print(f"case {case} {tag}: event round trip us median {np.median(ts):.1f} p90 {np.quantile(ts, 0.9):.1f} p99 {np.quantile(ts, 0.99):.1f}", flush=True)
print(f"case {case}: KFD queues at start {nq()}, threads {len(os.listdir('/proc/self/task'))}", flush=True)
evt_roundtrip("before")
sh = None
if case in "bek":
    sh = Shard(0, 1, Shard.unique_id())
    if case == "k":
        w = torch.zeros((1, 8), dtype=torch.float64, device=dev); sh.gather(w[0].clone(), w); torch.cuda.synchronize()
    if case == "e":
        sh.close(); sh = None
    print(f"case {case}: KFD queues after communicator {nq()}, threads {len(os.listdir('/proc/self/task'))}", flush=True)
    evt_roundtrip("after communicator")
prm = synth.MpcParams(T=0.66, K=8); S, n, ne, N = 256, 50000, 5000, prm.N
pl = Pipeline(nslots, S, n, ne, prm, queue_depth=qd, gang=gang)
frames = []
for i in range(nslots * gang):
    seed = 100000 + i * S
    cl, ed = synth.make_clouds_torch(n, S, seed, dev)
    sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
    for s in range(S):
        pos, vel, acc, yaw = synth.make_odom(seed + s, prm)
        sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
    frames.append((cl, ed, torch.from_numpy(sq).to(dev), torch.from_numpy(posx).to(dev), torch.from_numpy(ref0).to(dev)))
torch.cuda.synchronize()
def run(steps):
    t0 = time.perf_counter()
    for j in range(steps):
        f = frames[j % len(frames)]
        pl.submit(f[0], f[1], f[2], f[3], f[4], order_after_current_stream=False)
    pl.drain(); torch.cuda.synchronize()
    return time.perf_counter() - t0
run(2 * len(frames))
t = run(1024); tb = min(run(20) for _ in range(3))
print(f"case {case} {nslots}x{gang} depth {qd}: steady {S * 1024 / t:.0f} steps/s, 20-step burst {S * 20 / tb:.0f}; KFD queues {nq()}, threads {len(os.listdir('/proc/self/task'))}", flush=True)
