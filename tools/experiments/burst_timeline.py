"""Who runs when in the driver's 20-step burst (bench.py --steps 20 --warmup 5): every kernel launch of the burst with its
start and end, from HIP events on the launch streams (amk__timing_timeline) -- rocprofv3's kernel trace makes each submit
cost 0.8 ms of host time and so staggers the frames by itself; events cost ~2 us per launch.
Usage: python tools/experiments/burst_timeline.py [streams] [gang] [steps]"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, ".")
import numpy as np
import torch
from avoid_mpc_amd import capi, fsm, synth
from avoid_mpc_amd.host import Pipeline

nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gang = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
S, n = 256, 50000
ne = n // 10
prm = synth.MpcParams(T=0.66, K=8)
N = prm.N
dev = torch.device("cuda", 0)
lib = capi.load()
nframes = nslots * gang
frames = []
for i in range(nframes):
    seed = 100000 + i * S
    cl, ed = synth.make_clouds_torch(n, S, seed, dev)
    sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
    for s in range(S):
        pos, vel, acc, yaw = synth.make_odom(seed + s, prm)
        sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter)
        ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
    frames.append((cl, ed, torch.from_numpy(sq).to(dev), torch.from_numpy(posx).to(dev), torch.from_numpy(ref0).to(dev)))
pl = Pipeline(nslots, S, n, ne, prm, queue_depth=int(os.environ.get("QD", "1")), gang=gang)
u = torch.zeros((max(steps, nframes), S, 4), dtype=torch.float64, device=dev)
k = [0]


def step(row):
    f = frames[k[0] % nframes]; k[0] += 1
    pl.submit(f[0], f[1], f[2], f[3], f[4], u_out=u[row])


for j in range(nframes):
    step(j)
pl.drain(); torch.cuda.synchronize()
for j in range(5):
    step(j)
pl.drain(); torch.cuda.synchronize()
lib.amk__timing_enable(2)
t0 = time.perf_counter()
for j in range(steps):
    step(j)
t_sub = time.perf_counter() - t0
pl.drain(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
MAXR = 1 << 16
kc = (C.c_int * MAXR)(); a = (C.c_double * MAXR)(); b = (C.c_double * MAXR)()
nrec = lib.amk__timing_timeline(MAXR, kc, a, b)
lib.amk__timing_enable(0)
print(f"streams {nslots} gang {gang} steps {steps}: {S * steps / dt:.0f} scene-steps/s with events on, wall {dt * 1e3:.2f} ms, "
      f"submit {t_sub * 1e3:.2f} ms, {nrec} launches")
NAMES = {3: "knn", 4: "plan", 5: "solve", 6: "begin", 7: "build"}
# launches are recorded in submission order = launch-major: the chain of every launch (gang of frames), one row each
per = nrec // max(1, (steps + gang - 1) // gang)
print("chains (one row per launch, submission order): kernel start-end ms")
for f in range(0, nrec, per):
    print(f"{f // per:3d}: " + "  ".join(f"{NAMES.get(kc[i], str(kc[i]))[0]}{a[i]:.2f}-{b[i]:.2f}" for i in range(f, min(f + per, nrec))))
recs = sorted((a[i], b[i], NAMES.get(kc[i], str(kc[i]))) for i in range(nrec))
end = max(r[1] for r in recs)
B = 0.5
print("time-weighted number of running launches per class, per %.1f ms:" % B)
for q in range(int(end / B) + 1):
    lo, hi = q * B, (q + 1) * B
    acc = {}
    for s_, e_, nm in recs:
        ov = min(hi, e_) - max(lo, s_)
        if ov > 0:
            acc[nm] = acc.get(nm, 0.0) + ov / B
    print(f"{lo:5.1f} ms: " + "  ".join(f"{nm}:{acc.get(nm, 0):5.2f}" for nm in ("build", "knn", "plan", "solve")))
for nm in ("build", "solve", "knn", "plan"):
    rr = [(round(s_, 2), round(e_ - s_, 2)) for s_, e_, x in recs if x == nm]
    print(nm, "(start ms, duration ms):", rr)
