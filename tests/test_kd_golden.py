"""CPU: the C restatement reproduces the committed golden vectors (made by the reference's own
nanoflann header, tests/golden/make_kd_golden.py).  Runs without /root/reference."""
import os

import numpy as np
import pytest

from tests import _oracle
from avoid_mpc_amd import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "kd_golden.npz"))
FMAX = np.finfo(np.float64).max


def _check(tree, queries, k, prefix):
    idx, d2, cnt = G[f"{prefix}.k{k}.indices"], G[f"{prefix}.k{k}.sqdist"], G[f"{prefix}.k{k}.counts"]
    for i, q in enumerate(queries):
        a, b, _ = tree.search(q, k)
        assert len(a) == cnt[i]
        assert np.array_equal(a, idx[i, :cnt[i]])
        assert np.array_equal(b.view(np.int64), d2[i, :cnt[i]].view(np.int64))


@pytest.mark.parametrize("name", ["uniform2k", "corridor3k", "tiny5", "nan_x500"])
def test_small_clouds(name, oracle):
    t = _oracle.kd_oracle(G[f"{name}.cloud"])
    ks = (1, 3, 8) + ((5, 7) if name == "tiny5" else ())
    for k in ks:
        _check(t, G[f"{name}.queries"], k, name)
    if name == "tiny5":
        assert (G["tiny5.k5.counts"] == 0).all()      # size == n quirk, kd_tree_two.h:119-124
        assert (G["tiny5.k7.counts"] == 5).all()


@pytest.mark.parametrize("tag", ["c1_5k", "c2_50k", "c5_200k"])
def test_baseline_sizes(tag, oracle):
    n, seed = (int(v) for v in G[f"{tag}.seed"])
    cloud, edge = synth.make_cloud(n, seed)
    assert np.array_equal(np.array([cloud.astype(np.float64).sum(), edge.astype(np.float64).sum()]),
                          G[f"{tag}.cloud_sum"]), "synthetic generator drifted from the fixture"
    _check(_oracle.kd_oracle(cloud), G[f"{tag}.queries"], 8, tag)
    _check(_oracle.kd_oracle(edge), G[f"{tag}.queries"], 1, f"{tag}.edge")
