"""GPU: the control step over a MULTI-FRAME map (amk_step_batch_frames: current frame + keyframes, PtIsInFrame fast path,
per-frame k' = min(k, size) merge, minimum distance over the frames; AM/src/FrameKDMap.cpp:215-231,254-427) against the
oracle's restatement (oracle/step_oracle.c stepo_run_frames).  The frames are slices of one synthetic cloud as a camera
moving along +x would have seen them, so part of every reference path lies outside the current frustum and obstacles
behind the camera are remembered only by the keyframes."""
import numpy as np
import pytest

from tests import _oracle
from avoid_mpc_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _frames(sc, n_frames):
    """frame 0 (current) sees x in [6, 30), keyframes saw [0, 8), [3, 12) ...: overlapping slices."""
    c, e = sc["cloud"], sc["edge"]
    spans = [(6.0, 30.0), (0.0, 8.0), (3.0, 12.0), (-1.0, 1.0)][:n_frames]
    return ([c[(c[:, 0] >= a) & (c[:, 0] < b)] for a, b in spans], [e[(e[:, 0] >= a) & (e[:, 0] < b)] for a, b in spans])


@pytest.mark.parametrize("n_frames,with_camera,quantised", [(1, False, False), (3, True, False), (4, True, False),
                                                            (3, True, True)])
def test_multi_frame_step_matches_oracle(n_frames, with_camera, quantised):
    """quantised: clouds on a 0.25 m lattice and reference paths on a 0.125 m lattice (equal squared distances are the
    rule), every frame's handles in AMK_TIES_NANOFLANN mode -- the per-frame results then follow the reference's traversal
    and the step still matches the oracle, whose per-frame trees are nanoflann-shaped."""
    import torch
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch, step_batch_frames
    prm = synth.MpcParams(T=0.66, K=8)
    S = 12
    scenes = [synth.make_scene(20000, 900 + i, prm) for i in range(S)]
    if quantised:
        for sc in scenes:
            sc["cloud"] = (np.round(sc["cloud"] * 4) / 4).astype(np.float32)
            sc["edge"] = (np.round(sc["edge"] * 4) / 4).astype(np.float32)
            sc["ref_path"] = sc["ref_path"].copy(); sc["ref_path"][:, :3] = np.round(sc["ref_path"][:, :3] * 8) / 8
    N = prm.N
    # camera 2 m behind the start, looking along +x (camera z = world x, camera x = -world y, camera y = -world z)
    Twc = np.array([[0, 0, 1, -2.0], [-1, 0, 0, 0.0], [0, -1, 0, 1.5], [0, 0, 0, 1.0]])
    cam = (32.0, 32.0, 32.0, 24.0, 6.0, 64, 48)      # depth_max 6 m: the far end of the reference path is out of range
    kd_o, kd_e, fr = [], [], []
    for f in range(n_frames):
        cl = [_frames(sc, n_frames)[0][f] for sc in scenes]; ed = [_frames(sc, n_frames)[1][f] for sc in scenes]
        fr.append((cl, ed))
        for lst, out in ((cl, kd_o), (ed, kd_e)):
            nmax = max(max(len(x) for x in lst), 1)
            buf = np.zeros((S, nmax, 3), np.float32); cnt = np.zeros(S, np.int32)
            for s, x in enumerate(lst):
                buf[s, :len(x)] = x; cnt[s] = len(x)
            kd = KdBatch(S, nmax); kd.set_tie_order(1 if quantised else 0); kd.build(torch.from_numpy(buf).cuda(), torch.from_numpy(cnt).cuda()); out.append(kd)
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    sq = np.stack([_oracle.scene_state_quads(sc, prm) for sc in scenes])
    ref = torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()
    pos_x = torch.from_numpy(np.array([sc["pos"][0] for sc in scenes])).cuda()
    Tw = torch.from_numpy(np.repeat(Twc[None], S, 0).copy()).cuda() if with_camera else None
    fc = capi.FrameCamera(*cam) if with_camera else None
    out = step_batch_frames(kd_o, kd_e, mpc, prm, torch.from_numpy(sq).cuda(), pos_x, ref, Twc=Tw, cam=fc)
    torch.cuda.synchronize()
    u, x0, flags, rp = out["u"].cpu().numpy(), out["x0array"].cpu().numpy(), out["flags"].cpu().numpy(), ref.cpu().numpy()
    worst = 0.0
    for s, sc in enumerate(scenes):
        ko = [_oracle.kd_oracle(fr[f][0][s]) for f in range(n_frames)]
        ke = [_oracle.kd_oracle(fr[f][1][s]) for f in range(n_frames)]
        m = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m.configure(prm)
        r_ref = sc["ref_path"].copy()
        r = _oracle.step_oracle_frames(ko, ke, m, prm, sq[s], sc["pos"][0], r_ref, Twc if with_camera else None,
                                       cam if with_camera else None)
        assert np.array_equal(flags[s], r["flags"]), (s, flags[s], r["flags"])
        worst = max(worst, np.abs(u[s] - r["u"]).max(), np.abs(x0[s] - r["x0array"]).max(), np.abs(rp[s] - r_ref).max())
    print(f"frames {n_frames}: worst |gpu - oracle| = {worst:.3e}; solves {flags[:, 1].tolist()}")
    assert worst <= TOL
    if n_frames == 1:   # one frame, no camera: identical to amk_step_batch
        mpc1 = MpcBatch(prm.T, prm.dt, prm.K, S); mpc1.configure(prm)
        ref1 = torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()
        o1 = step_batch(kd_o[0], kd_e[0], mpc1, prm, torch.from_numpy(sq).cuda(), pos_x, ref1)
        torch.cuda.synchronize()
        assert torch.equal(o1["u"], out["u"]) and torch.equal(o1["flags"], out["flags"]) and torch.equal(ref1, ref)


def test_exact_distance_ties_between_frames():
    """two frames that hold DIFFERENT points at exactly the same squared distance from a reference point.
    Lattice clouds (0.25 m) cut into disjoint frames at x = 6: frame 0 holds x >= 6, frame 1 x < 6; reference points sit on
    the 0.125 m lattice and one of them exactly on the cut, so (5.75, y, z) of frame 1 and (6.25, y, z) of frame 0 tie.  The
    reference merges with an unstable std::sort on the distance alone (FrameKDMap.cpp:371); this library and the oracle
    keep the EARLIER frame first.  The test first proves that such ties occur among the K nearest of the merged lists, then
    compares the whole step (which points entered P decides the solve)."""
    import torch
    from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch_frames
    prm = synth.MpcParams(T=0.66, K=8)
    S = 8
    scenes = [synth.make_scene(20000, 1900 + i, prm) for i in range(S)]
    spans = [(6.0, 1e9), (-1e9, 6.0)]
    ties = 0
    for sc in scenes:
        sc["cloud"] = np.unique((np.round(sc["cloud"] * 4) / 4).astype(np.float32), axis=0)
        sc["edge"] = np.unique((np.round(sc["edge"] * 4) / 4).astype(np.float32), axis=0)
        rp = sc["ref_path"].copy(); rp[:, :3] = np.round(rp[:, :3] * 8) / 8
        k0 = int(np.argmin(np.abs(rp[:, 0] - 6.0))); rp[k0, 0] = 6.0
        sc["ref_path"] = rp
    fr = [([sc["cloud"][(sc["cloud"][:, 0] >= a) & (sc["cloud"][:, 0] < b)] for sc in scenes],
           [sc["edge"][(sc["edge"][:, 0] >= a) & (sc["edge"][:, 0] < b)] for sc in scenes]) for a, b in spans]
    for s, sc in enumerate(scenes):   # how many reference points see an inter-frame tie inside their merged K nearest
        t0, t1 = _oracle.kd_oracle(fr[0][0][s]), _oracle.kd_oracle(fr[1][0][s])
        for p in sc["ref_path"][:, :3]:
            (i0, d0, p0), (i1, d1, p1) = t0.search(p, prm.K), t1.search(p, prm.K)
            kth = np.sort(np.concatenate([d0, d1]))[prm.K - 1]
            common = np.intersect1d(d0[d0 <= kth], d1[d1 <= kth])
            ties += len(common) > 0
    assert ties >= 8, ties
    kd_o, kd_e = [], []
    for f in range(2):
        for lst, out in ((fr[f][0], kd_o), (fr[f][1], kd_e)):
            nmax = max(max(len(x) for x in lst), 1)
            buf = np.zeros((S, nmax, 3), np.float32); cnt = np.zeros(S, np.int32)
            for s, x in enumerate(lst):
                buf[s, :len(x)] = x; cnt[s] = len(x)
            kd = KdBatch(S, nmax); kd.set_tie_order(1); kd.build(torch.from_numpy(buf).cuda(), torch.from_numpy(cnt).cuda()); out.append(kd)
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    sq = np.stack([_oracle.scene_state_quads(sc, prm) for sc in scenes])
    ref = torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()
    pos_x = torch.from_numpy(np.array([sc["pos"][0] for sc in scenes])).cuda()
    out = step_batch_frames(kd_o, kd_e, mpc, prm, torch.from_numpy(sq).cuda(), pos_x, ref)   # no camera: every frame is searched
    torch.cuda.synchronize()
    u, x0, flags, rp = out["u"].cpu().numpy(), out["x0array"].cpu().numpy(), out["flags"].cpu().numpy(), ref.cpu().numpy()
    worst = 0.0
    for s, sc in enumerate(scenes):
        ko = [_oracle.kd_oracle(fr[f][0][s]) for f in range(2)]
        ke = [_oracle.kd_oracle(fr[f][1][s]) for f in range(2)]
        m = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m.configure(prm)
        r_ref = sc["ref_path"].copy()
        r = _oracle.step_oracle_frames(ko, ke, m, prm, sq[s], sc["pos"][0], r_ref, None, None)
        assert np.array_equal(flags[s], r["flags"]), (s, flags[s], r["flags"])
        worst = max(worst, np.abs(u[s] - r["u"]).max(), np.abs(x0[s] - r["x0array"]).max(), np.abs(rp[s] - r_ref).max())
    print(f"inter-frame ties at {ties} reference points; worst |gpu - oracle| = {worst:.3e}")
    assert worst <= TOL
