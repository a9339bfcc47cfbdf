"""GPU: the keyframe maintenance sweep (FrameKDMap::KeyframeThreadWorker,
AM/src/FrameKDMap.cpp:462-485) -- n 1-NN queries of the last keyframe's points against the current
frame's index, outlier filter, rebuild -- against the oracle's restatement."""
import numpy as np
import pytest

from tests import _oracle
from avoid_mpc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tie_order", [0, 1])
def test_sweep_matches_oracle(tie_order):
    """tie_order = 1: clouds on a 0.05 m lattice (equal squared distances are common) and both handles in
    AMK_TIES_NANOFLANN mode -- the keyframe rebuilt by the sweep must then also return nanoflann's INDEX lists (its
    reference-shaped tree is rebuilt from the compacted planes)."""
    import torch
    from avoid_mpc_amd.host import KdBatch
    rng = np.random.default_rng(5)
    th_dist, th_count = 0.1, 10                       # keyframe_th_dist / keyframe_th_count, mpc_parameters.yaml:69-70
    scenes = []
    for s in range(6):
        cur = synth.make_cloud(20000, 700 + s)[0]
        kf = synth.make_cloud(20000, 700 + s)[0].copy()
        # the keyframe is the previous frame: mostly the same surface points (shifted a little) plus a
        # region the current frame no longer sees
        kf += rng.normal(0, 0.02 if s % 2 == 0 else 0.2, kf.shape).astype(np.float32)
        kf[:3000, 0] -= 6.0
        if s == 4:
            kf = cur.copy()                           # identical: no outliers -> no rebuild
        if s == 5:
            cur = cur[:1]                             # current tree with one point: SearchForNearest(.,1) gives nothing
        if tie_order:
            kf = np.round(kf * 20) / 20; cur = np.round(cur * 20) / 20
        scenes.append((kf.astype(np.float32), cur.astype(np.float32)))
    S = len(scenes)
    nk = max(len(k) for k, _ in scenes); nc = max(len(c) for _, c in scenes)
    kb = np.zeros((S, nk, 3), np.float32); cb = np.zeros((S, nc, 3), np.float32)
    kn = np.zeros(S, np.int32); cn = np.zeros(S, np.int32)
    for s, (k, c) in enumerate(scenes):
        kb[s, :len(k)] = k; kn[s] = len(k); cb[s, :len(c)] = c; cn[s] = len(c)
    kd_k, kd_c = KdBatch(S, nk), KdBatch(S, nc)
    kd_k.set_tie_order(tie_order); kd_c.set_tie_order(tie_order)
    kd_k.build(torch.from_numpy(kb).cuda(), torch.from_numpy(kn).cuda())
    kd_c.build(torch.from_numpy(cb).cuda(), torch.from_numpy(cn).cuda())
    outl, reb = kd_k.keyframe_sweep(kd_c, th_dist, th_count)
    torch.cuda.synchronize()
    outl, reb, sizes = outl.cpu().numpy(), reb.cpu().numpy(), kd_k.sizes()
    qs = np.stack([rng.uniform(0, 20, (S, 16)), rng.uniform(-6, 6, (S, 16)), rng.uniform(0, 4, (S, 16))], -1)
    if tie_order:
        qs = np.round(qs * 40) / 40
    res = kd_k.search(torch.from_numpy(qs).cuda(), 8)
    torch.cuda.synchronize()
    for s, (k, c) in enumerate(scenes):
        tk, tc = _oracle.kd_oracle(k), _oracle.kd_oracle(c)
        r, n_out = tk.keyframe_sweep(tc, th_dist, th_count)
        assert outl[s] == n_out and reb[s] == r, (s, outl[s], n_out, reb[s], r)
        assert sizes[s] == tk.size()
        for q in range(16):                            # the rebuilt (or untouched) keyframe answers like the oracle's
            ia, da, pa = tk.search(qs[s, q], 8)
            cnt = res["counts"][s, q].item()
            assert cnt == len(ia)
            assert np.array_equal(res["indices"][s, q, :cnt].cpu().numpy(), ia)
            assert np.array_equal(res["sqdist"][s, q, :cnt].cpu().numpy(), da)
    assert reb[4] == 0 and outl[4] == 0 and reb[5] == 0 and reb[0] == 1


def test_sweep_and_rebuild_at_ragged_sizes():
    """The compaction kernel takes four consecutive elements per thread and trip (round 6): keyframes of 1 ... 4099 points -- below one
    vector, not a multiple of four, around the 2048-element trip -- with every third point an outlier, rebuilt
    from their outliers in order (th_count 1): outlier counts, rebuilt flags, sizes and the answers of the rebuilt keyframe == oracle."""
    import torch
    from avoid_mpc_amd.host import KdBatch
    sizes = [1, 2, 3, 4, 5, 7, 13, 255, 256, 257, 1023, 1025, 2047, 2048, 2049, 4099]
    S = len(sizes)
    rng = np.random.default_rng(9)
    cur = synth.make_cloud(6000, 911)[0]
    nk = max(sizes)
    kb = np.zeros((S, nk, 3), np.float32); kn = np.array(sizes, np.int32)
    ks = []
    for s, n in enumerate(sizes):
        k = cur[rng.integers(0, len(cur), n)].copy() + rng.normal(0, 0.005, (n, 3)).astype(np.float32)   # inliers: next to a current point
        k[::3, 0] -= 9.0                                                                                   # every third point: far away
        kb[s, :n] = k; ks.append(k)
    kd_k, kd_c = KdBatch(S, nk), KdBatch(S, len(cur))
    kd_k.build(torch.from_numpy(kb).cuda(), torch.from_numpy(kn).cuda())
    kd_c.build(torch.from_numpy(np.repeat(cur[None], S, 0).copy()).cuda())
    outl, reb = kd_k.keyframe_sweep(kd_c, 0.1, 1)
    torch.cuda.synchronize()
    outl, reb, got = outl.cpu().numpy(), reb.cpu().numpy(), kd_k.sizes()
    qs = np.stack([rng.uniform(-9, 20, (S, 6)), rng.uniform(-6, 6, (S, 6)), rng.uniform(0, 4, (S, 6))], -1)
    res = kd_k.search(torch.from_numpy(qs).cuda(), 3)
    torch.cuda.synchronize()
    tc = _oracle.kd_oracle(cur)
    for s, k in enumerate(ks):
        tk = _oracle.kd_oracle(k)
        r, n_out = tk.keyframe_sweep(tc, 0.1, 1)
        assert outl[s] == n_out and reb[s] == r and got[s] == tk.size(), (sizes[s], outl[s], n_out, reb[s], r, got[s], tk.size())
        for q in range(6):
            ia, da, _ = tk.search(qs[s, q], 3)
            assert np.array_equal(res["sqdist"][s, q].cpu().numpy()[:len(da)].view(np.int64), da.view(np.int64)), (sizes[s], q)
