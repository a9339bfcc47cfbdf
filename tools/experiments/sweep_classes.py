"""Where the keyframe sweep's time goes: the same 512-scene sweep of 50 k-point flight frames (frame t against frame t + 1 of
FlightWorldsTorch: what the closed loop sweeps every period) with (a) the real pair, (b) keyframe = current frame (every query has a
neighbour at distance 0: all inliers), (c) the keyframe moved 5 m sideways (no query has a neighbour within th: all outliers).
Run under rocprofv3 and read the mark kernel's durations in launch order (12 per class):
  rocprofv3 --kernel-trace -d out -o kt -- python tools/experiments/sweep_classes.py ; python tools/experiments/sweep_classes.py --read out
usage: python tools/experiments/sweep_classes.py [scenes] [points] [period]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import glob, sqlite3
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True)[0])
    d = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like '%kd_sweep_mark%' order by start")]
    for i, name in enumerate(("real pair", "all inliers", "all outliers")):
        print(name, "mark kernel us:", [round(x, 1) for x in d[12 * i:12 * i + 12]])
    sys.exit(0)
REPS = 12
import torch
from avoid_mpc_amd import flight, synth
from avoid_mpc_amd.host import KdBatch

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
t = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda", 0)
prm = synth.MpcParams(T=0.66, K=8)
w = flight.FlightWorldsTorch(S, n, prm, 9000, dev, length=80.0)
kf_cloud, _ = w.frame(t)
cur_cloud, _ = w.frame(t + 1)
cur = KdBatch(S, n); cur.build(cur_cloud)
for name, cloud in (("real pair (frame t vs t + 1)", kf_cloud), ("all inliers (keyframe = current)", cur_cloud),
                    ("all outliers (keyframe moved 5 m in y)", kf_cloud + torch.tensor([0.0, 5.0, 0.0], device=dev))):
    ms = []
    for rep in range(REPS):
        kf = KdBatch(S, n); kf.build(cloud.contiguous())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outl, reb = kf.keyframe_sweep(cur, 0.1, 10)
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
        del kf
    print(f"{name}: {min(ms[1:]):.3f} ms per {S}-scene sweep; outliers per scene mean {outl.float().mean().item():.0f} of {n}", flush=True)
