"""Randomised cross-check of the depth kernels (FrameKDMap::ProcessDepth / BuildEdgeCloud on the device, csrc/depth.hip) against
oracle/depth_oracle.c: random image sizes (odd, tiny, wide), resize scales, pixel types, depth ranges with holes / out-of-range /
non-finite pixels, intrinsics, poses; both clouds must have the oracle's count, order and float bits.
usage: python tools/experiments/fuzz_depth.py [seed] [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tests import _oracle
from avoid_mpc_amd.host import depth_params, depth_to_cloud

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
it = bad = frames = 0
while time.time() < t_end:
    it += 1
    rows, cols = int(rng.integers(1, 200)), int(rng.integers(1, 260))
    if rng.random() < 0.2:
        rows, cols = 480, 640
    scale = float(rng.choice([1.0, 2.0, 2.5, 3.0, 4.0, 7.3, 10.0]))
    if int(rows / scale) < 1 or int(cols / scale) < 1:
        continue
    dtype = np.uint16 if rng.random() < 0.5 else np.float32
    S = int(rng.integers(1, 5))
    far = float(rng.choice([5.0, 30.0, 90.0]))
    imgs = []
    for s in range(S):
        d = rng.uniform(0.05, far, (rows, cols))
        if rng.random() < 0.5:   # piecewise-constant walls: real edges for Canny
            d = np.kron(rng.uniform(0.3, far, (rows // 16 + 1, cols // 16 + 1)), np.ones((16, 16)))[:rows, :cols]
        d[rng.random((rows, cols)) < 0.1] = 0.0
        d[rng.random((rows, cols)) < 0.05] = 150.0
        if dtype == np.uint16:
            imgs.append(np.round(np.minimum(d, 65.0) * 1000).astype(np.uint16))
        else:
            d = d.astype(np.float32)
            m = rng.random((rows, cols))
            d[m < 0.01] = np.nan; d[(m >= 0.01) & (m < 0.02)] = np.inf; d[(m >= 0.02) & (m < 0.03)] = -1.0
            imgs.append(d)
    imgs = np.stack(imgs)
    th = rng.uniform(-np.pi, np.pi, S)
    Twb = np.zeros((S, 4, 4))
    for s in range(S):
        Twb[s] = [[np.cos(th[s]), -np.sin(th[s]), 0, rng.uniform(-50, 50)], [np.sin(th[s]), np.cos(th[s]), 0, rng.uniform(-50, 50)],
                  [0, 0, 1, rng.uniform(0.5, 3)], [0, 0, 0, 1]]
    prm = dict(pixel2meter=1e-3 if dtype == np.uint16 else 1.0, depth_min=float(rng.choice([0.1, 0.5])), depth_max=float(rng.choice([10.0, 100.0])),
               resize_scale=scale, fx=float(rng.uniform(100, 400)), fy=float(rng.uniform(100, 400)), cx=cols / 2.0 + rng.uniform(-3, 3),
               cy=rows / 2.0 + rng.uniform(-3, 3), Tbc=np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.0], [0, -1, 0, 0.05], [0, 0, 0, 1.0]]))
    dev = torch.from_numpy(imgs.view(np.int16) if dtype == np.uint16 else imgs).cuda()
    tw = torch.from_numpy(Twb).cuda()
    cloud, counts = depth_to_cloud(dev, depth_params(**prm), tw, point_stride=int(rng.choice([3, 4])))
    with_edge = int(rows / scale) * int(cols / scale) <= 16000   # AMK_EDGE_MAX_PIXELS (one workgroup per image, LDS-resident)
    if with_edge:
        edge, ecounts = depth_to_cloud(dev, depth_params(**prm), tw, point_stride=3, edge=True)
    torch.cuda.synchronize()
    cloud, counts = cloud.cpu().numpy(), counts.cpu().numpy()
    if with_edge:
        edge, ecounts = edge.cpu().numpy(), ecounts.cpu().numpy()
    for s in range(S):
        frames += 1
        ref, _ = _oracle.depth_oracle(imgs[s], prm, Twb[s])
        ok = counts[s] == len(ref) and np.array_equal(cloud[s, :counts[s], :3].view(np.uint32), ref.view(np.uint32))
        oke = True
        if with_edge:
            eref = _oracle.depth_edge_oracle(imgs[s], prm, Twb[s])[0]
            oke = ecounts[s] == len(eref) and np.array_equal(edge[s, :ecounts[s], :3].view(np.uint32), eref.view(np.uint32))
        if not (ok and oke):
            bad += 1
            print("MISMATCH", it, rows, cols, scale, dtype.__name__, s, "cloud", ok, counts[s], len(ref), "edge", oke, flush=True)
print("fuzz iterations", it, "frames checked", frames, "mismatches", bad)
