import numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from tests import test_kd_gpu as T
G = T.G
for name in ["uniform2k", "corridor3k", "tiny5"]:
    cloud, qs = G[f"{name}.cloud"], G[f"{name}.queries"]
    for k in (1, 3, 8):
        r = T._gpu_search(torch, [cloud], qs[None], k)
        gi, gd = G[f"{name}.k{k}.indices"], G[f"{name}.k{k}.sqdist"]
        bad = [i for i in range(len(qs)) if not np.array_equal(r["indices"][0, i], gi[i]) or not np.array_equal(r["sqdist"][0, i], gd[i])]
        print(name, k, "queries", len(qs), "bad", len(bad))
        for i in bad[:3]:
            print("  q", i, qs[i], "\n   got", r["indices"][0, i], r["sqdist"][0, i], "\n   exp", gi[i], gd[i])
