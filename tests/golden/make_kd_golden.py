"""Generates tests/golden/kd_golden.npz from the REFERENCE's nanoflann header compiled in place
(oracle/_ref/libnanoflann_ref_strict.so; see oracle/ref_nanoflann.cpp and oracle/Makefile).

Run in the build container (needs /root/reference):   python tests/golden/make_kd_golden.py
The .npz holds data only: input clouds (small ones in full, large ones as a generator seed for
avoid_mpc_amd.synth.make_cloud), query points, and the reference's answers
(KDTreeTwo::SearchForNearest semantics: indices, squared distances, result counts).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _oracle  # noqa: E402
from avoid_mpc_amd import synth  # noqa: E402


def answers(cloud, queries, k):
    t = _oracle.kd_ref(cloud, strict=True)
    assert t is not None, "oracle/_ref missing: run `make -C oracle ref` with /root/reference present"
    idx = np.full((len(queries), k), -1, np.int32)
    d2 = np.full((len(queries), k), np.finfo(np.float64).max)
    cnt = np.zeros(len(queries), np.int32)
    for i, q in enumerate(queries):
        a, b, _ = t.search(q, k)
        cnt[i] = len(a)
        idx[i, :len(a)] = a
        d2[i, :len(a)] = b
    return idx, d2, cnt


def main():
    out = {}
    rng = np.random.default_rng(20250928)
    # 1. small clouds stored in full
    small = {
        "uniform2k": rng.uniform(-10, 10, (2000, 3)).astype(np.float32),
        "corridor3k": synth.make_cloud(3000, 42)[0],
        "tiny5": rng.uniform(-1, 1, (5, 3)).astype(np.float32),
    }
    nanc = rng.uniform(-2, 2, (500, 3)).astype(np.float32)
    nanc[rng.choice(500, 40, replace=False), 0] = np.nan
    small["nan_x500"] = nanc
    for name, cloud in small.items():
        valid = cloud[~np.isnan(cloud[:, 0])]
        lo, hi = valid.min(0) - 0.5, valid.max(0) + 0.5
        qs = rng.uniform(lo, hi, (64, 3))
        out[f"{name}.cloud"] = cloud
        out[f"{name}.queries"] = qs
        for k in (1, 3, 8):
            idx, d2, cnt = answers(cloud, qs, k)
            out[f"{name}.k{k}.indices"], out[f"{name}.k{k}.sqdist"], out[f"{name}.k{k}.counts"] = idx, d2, cnt
    # size == k quirk and size < k
    for k in (5, 7):
        idx, d2, cnt = answers(small["tiny5"], out["tiny5.queries"], k)
        out[f"tiny5.k{k}.indices"], out[f"tiny5.k{k}.sqdist"], out[f"tiny5.k{k}.counts"] = idx, d2, cnt
    # 2. BASELINE.json sizes: cloud = synth.make_cloud(n, seed)[0] (obstacle) and [1] (edge)
    for tag, n, seed in (("c1_5k", 5000, 1001), ("c2_50k", 50000, 1002), ("c5_200k", 200000, 1005)):
        cloud, edge = synth.make_cloud(n, seed)
        qs = np.stack([rng.uniform(0, 12, 44), rng.uniform(-2, 2, 44), rng.uniform(0.5, 2.5, 44)], 1)
        out[f"{tag}.seed"] = np.array([n, seed])
        out[f"{tag}.queries"] = qs
        out[f"{tag}.cloud_sum"] = np.array([cloud.astype(np.float64).sum(), edge.astype(np.float64).sum()])
        idx, d2, cnt = answers(cloud, qs, 8)
        out[f"{tag}.k8.indices"], out[f"{tag}.k8.sqdist"], out[f"{tag}.k8.counts"] = idx, d2, cnt
        idx, d2, cnt = answers(edge, qs, 1)
        out[f"{tag}.edge.k1.indices"], out[f"{tag}.edge.k1.sqdist"], out[f"{tag}.edge.k1.counts"] = idx, d2, cnt
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kd_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
