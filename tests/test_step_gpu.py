"""GPU: one whole control step (amk_step_batch: dual KD queries -> pack P -> solve, <= 3 passes)
against the CPU oracle's restatement of AvoidanceStateMachine::Step's TASK branch.

Neighbour sets are bit-exact (see test_kd_gpu.py), so both sides solve identical problems; the
control/trajectory tolerance is the one of test_mpc_gpu.py (1e-6, fp64)."""
import numpy as np
import pytest

from tests import _oracle
from avoid_mpc_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def run_both(torch, scenes, prm, n_steps=1, tie_order=0):
    """scenes: list of dict(cloud, edge, pos, vel, acc, yaw, ref_path).  Returns (gpu, cpu) lists of
    per-step results."""
    from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch
    S, N = len(scenes), prm.N
    nmax = max(max(len(sc["cloud"]) for sc in scenes), 1)
    emax = max(max(len(sc["edge"]) for sc in scenes), 1)
    cl = np.zeros((S, nmax, 3), np.float32); ed = np.zeros((S, emax, 3), np.float32)
    cn = np.zeros(S, np.int32); en = np.zeros(S, np.int32)
    for s, sc in enumerate(scenes):
        cl[s, :len(sc["cloud"])] = sc["cloud"]; cn[s] = len(sc["cloud"])
        ed[s, :len(sc["edge"])] = sc["edge"]; en[s] = len(sc["edge"])
    kd_o, kd_e = KdBatch(S, nmax), KdBatch(S, emax)
    kd_o.set_tie_order(tie_order); kd_e.set_tie_order(tie_order)
    kd_o.build(torch.from_numpy(cl).cuda(), torch.from_numpy(cn).cuda())
    kd_e.build(torch.from_numpy(ed).cuda(), torch.from_numpy(en).cuda())
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    sq = np.stack([_oracle.scene_state_quads(sc, prm) for sc in scenes])
    ref = torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()
    pos_x = torch.from_numpy(np.array([sc["pos"][0] for sc in scenes])).cuda()
    gpu = []
    for _ in range(n_steps):
        out = step_batch(kd_o, kd_e, mpc, prm, torch.from_numpy(sq).cuda(), pos_x, ref)
        torch.cuda.synchronize()
        gpu.append(dict(u=out["u"].cpu().numpy().copy(), x0array=out["x0array"].cpu().numpy().copy(),
                        flags=out["flags"].cpu().numpy().copy(), ref_path=ref.cpu().numpy().copy()))
    cpu = [[] for _ in range(n_steps)]
    for s, sc in enumerate(scenes):
        ko, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
        m = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m.configure(prm)
        rp = sc["ref_path"].copy()
        for t in range(n_steps):
            r = _oracle.step_oracle(ko, ke, m, prm, sq[s], sc["pos"][0], rp)
            r["ref_path"] = rp.copy()
            cpu[t].append(r)
    return gpu, cpu


def compare(gpu, cpu, tol=TOL):
    """Scene by scene: identical flags {isSafety, solves, status, interior-point iterations} and |gpu - oracle| <= tol.
    A rounding-level branch flip inside a solve (tests/test_mpc_gpu.py) may change the iteration COUNT of a scene; it must
    then keep isSafety / solves / status (first step only: later steps inherit the warm start).  Such a scene normally still
    ends at the same optimum (1e-4 asserted); the problem is non-convex and non-smooth, so once in a few hundred scenes the
    flipped branch leads to ANOTHER local minimum of the same NLP (both sides converged, status 0); they are printed.
    Allowances = ~3 x what the census over the 2048 scenes of configs[3] observed (4 flipped = 0.2 %, 3 of them in another
    minimum = 0.15 %): <= 1 % flipped, <= 0.5 % in another minimum, and one scene of each kind in a
    batch too small for the percentages to mean anything."""
    worst, flipped, other_basin, total = 0.0, 0, [], 0
    diverged = set()
    for t in range(len(gpu)):
        for s, r in enumerate(cpu[t]):
            if s in diverged:
                continue
            total += 1
            du = np.abs(gpu[t]["u"][s] - r["u"]).max()
            dx = np.abs(gpu[t]["x0array"][s] - r["x0array"]).max() if r["flags"][1] > 0 else 0.0
            dr = np.abs(gpu[t]["ref_path"][s] - r["ref_path"]).max()
            if np.array_equal(gpu[t]["flags"][s], r["flags"]):
                worst = max(worst, du, dx, dr)
                assert max(du, dx, dr) <= tol, (t, s, du, dx, dr)
            else:
                flipped += 1
                diverged.add(s)
                assert np.array_equal(gpu[t]["flags"][s][:3], r["flags"][:3]), (t, s, gpu[t]["flags"][s], r["flags"])
                if max(du, dx, dr) > 1e-4:
                    assert r["flags"][2] == 0, (t, s, r["flags"])      # both converged
                    other_basin.append((t, s, gpu[t]["flags"][s].tolist(), r["flags"].tolist(), float(du)))
    if other_basin:
        print("scenes whose flipped branch ended in another local minimum:", other_basin)
    assert flipped <= max(1, total // 100), (flipped, total)
    assert len(other_basin) <= max(1, total // 200), other_basin
    return worst


@pytest.mark.parametrize("cfg,n,scenes", [("C1", 5000, 12), ("C2", 50000, 12), ("C5", 200000, 4)])
def test_step_matches_oracle(cfg, n, scenes, torch_cuda):
    """BASELINE configs at their full sizes (C5 = 200k-point cloud + 20k-point edge cloud, N = 30), fp64."""
    c = synth.CONFIGS[cfg]
    prm = synth.MpcParams(T=c["T"], K=c["K"])
    scenes = [synth.make_scene(n, 300 + i, prm) for i in range(scenes)]
    gpu, cpu = run_both(torch_cuda, scenes, prm, n_steps=2)    # second step: warm start carried over
    w = compare(gpu, cpu)
    print(f"{cfg}: worst |gpu - oracle| = {w:.3e}; solves/step = {[r['flags'][1] for r in cpu[0]]}")


def test_reference_default_problem_size(torch_cuda):
    """The reference's own default configuration (AM/config/mpc_parameters.yaml:1-2,5,59-63): 640x480 depth / 10 ->
    <= 3072 points per frame, N = 30 (T = 1.0, dt = 0.033), K = 3 -- BASELINE's configs are scale-ups of this."""
    prm = synth.MpcParams(T=1.0, K=3)
    scenes = [synth.make_scene(3072, 700 + i, prm) for i in range(16)]
    gpu, cpu = run_both(torch_cuda, scenes, prm, n_steps=3)    # three control periods, warm start carried over
    w = compare(gpu, cpu)
    print(f"yaml default size: worst |gpu - oracle| = {w:.3e}; solves/step = {[r['flags'][1] for r in cpu[0]]}")


def test_step_in_nanoflann_tie_order_on_quantised_clouds(torch_cuda):
    """Clouds and reference paths on a 0.25 m / 0.125 m lattice: equal squared distances everywhere (the K-th neighbour of a
    reference point and the nearest edge point are regularly tied), so WHICH point nanoflann keeps decides the problem the
    solver sees.  With amk_kd_set_tie_order(AMK_TIES_NANOFLANN) on both handles the fused step -- first queries, the edge
    snap and its re-query -- follows the reference's traversal and matches the oracle's step (which searches the oracle's
    nanoflann-shaped tree) to the usual tolerance; the default lowest-index policy picks other members of the ties."""
    prm = synth.MpcParams(T=1.0, K=3)
    scenes = []
    for i in range(16):
        sc = synth.make_scene(3072, 1500 + i, prm)
        sc["cloud"] = (np.round(sc["cloud"] * 4) / 4).astype(np.float32)
        sc["edge"] = (np.round(sc["edge"] * 4) / 4).astype(np.float32)
        sc["ref_path"] = sc["ref_path"].copy(); sc["ref_path"][:, :3] = np.round(sc["ref_path"][:, :3] * 8) / 8
        if i % 2:   # reference point 0 within the safety distance of an obstacle -> snap to a (tied) nearest edge point
            sc["cloud"] = np.concatenate([sc["cloud"], (sc["ref_path"][0, :3] + [0.125, 0, 0])[None].astype(np.float32)])
            e0 = sc["ref_path"][0, :3]
            ring = np.float32([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]) * 0.5 + e0
            sc["edge"] = np.concatenate([sc["edge"], ring.astype(np.float32)])
        scenes.append(sc)
    gpu, cpu = run_both(torch_cuda, scenes, prm, n_steps=2, tie_order=1)
    w = compare(gpu, cpu)
    gpu0, _ = run_both(torch_cuda, scenes, prm, n_steps=1, tie_order=0)
    differ = sum(np.abs(gpu0[0]["u"][s] - cpu[0][s]["u"]).max() > 1e-6 or
                 np.abs(gpu0[0]["ref_path"][s] - cpu[0][s]["ref_path"]).max() > 0 for s in range(len(scenes)))
    print(f"quantised clouds, nanoflann tie order: worst |gpu - oracle| = {w:.3e}; "
          f"default tie policy differs from the reference on {differ}/{len(scenes)} scenes")
    assert differ >= 1      # the mode is what makes these scenes agree


def test_full_batch_c3_and_a_c4_shard(torch_cuda):
    """BASELINE configs[2] at its full batch (256 scenes x 50k points in ONE call: the XCD-aware scene mapping
    s = (j / bps) * 8 + xcd and every size_t stride at S = 256) and the scenes a rank of configs[3] owns (2048 scenes
    block-partitioned over 8 ranks, tests/_shard_torch.py: rank 5 holds scenes 1280..1535; its first 64)."""
    from tests import _shard_torch as shard
    prm = synth.MpcParams(T=0.66, K=8)
    scenes = [synth.make_scene(50000, 100000 + s, prm) for s in range(256)]
    gpu, cpu = run_both(torch_cuda, scenes, prm, n_steps=1)
    w = compare(gpu, cpu)
    print(f"C3, S = 256: worst |gpu - oracle| = {w:.3e}")
    lo, hi = shard.scene_range(5, 8, 2048)
    assert (lo, hi) == (1280, 1536)
    scenes = [synth.make_scene(50000, 100000 + g, prm) for g in range(lo, lo + 64)]
    gpu, cpu = run_both(torch_cuda, scenes, prm, n_steps=1)
    w = compare(gpu, cpu)
    print(f"C4, rank 5 of 8, scenes {lo}..{lo + 63}: worst |gpu - oracle| = {w:.3e}")


def test_c4_every_scene_of_the_2048(torch_cuda):
    """BASELINE configs[3] in full: all 2048 scenes (8 shards of 256, the seeds of the shard test above) against the oracle.
    Census on MI355X: 2044 scenes with identical flags -- |du| median 3e-14, p99 9e-13, 5 scenes above 1e-9, worst 6.7e-6
    (same branches, an ill-conditioned scene) -- and 4 that took another branch at a rounding-level tie: 1 to the same
    optimum, 3 to another local minimum (0.15 %).  Asserted: >= 99 % identical flags, those within 1e-4 with a median
    <= 1e-11, at most 1 scene in 64 of each shard in another local minimum (compare())."""
    prm = synth.MpcParams(T=0.66, K=8)
    dus, flipped = [], 0
    for lo in range(0, 2048, 256):
        scenes = [synth.make_scene(50000, 100000 + g, prm) for g in range(lo, lo + 256)]
        gpu, cpu = run_both(torch_cuda, scenes, prm, n_steps=1)
        compare(gpu, cpu, tol=1e-4)
        for s, r in enumerate(cpu[0]):
            if np.array_equal(gpu[0]["flags"][s], r["flags"]):
                dus.append(np.abs(gpu[0]["u"][s] - r["u"]).max())
            else:
                flipped += 1
    dus = np.sort(np.array(dus))
    print(f"C4, 2048 scenes: identical flags {len(dus)}, |du| median {np.median(dus):.1e} p99 {dus[int(0.99 * len(dus))]:.1e} "
          f"max {dus[-1]:.1e}; other branch {flipped}")
    assert len(dus) >= 0.99 * 2048 and np.median(dus) <= 1e-11


def test_edge_snap_and_unsafe_and_tiny_clouds(torch_cuda):
    """PlanWapionts paths: reference point 0 within safety_distance of an obstacle -> snapped to the
    nearest edge point (the Edge-KD-tree warm start); no edge point -> isSafety false; clouds with
    <= K points -> every obstacle padded with 1e4 (AvoidanceStateMachine.cpp:223-226)."""
    prm = synth.MpcParams(T=0.33, K=3)
    scenes = []
    base = synth.make_scene(5000, 900, prm)
    # (a) obstacle 5 cm from reference point 0, edge cloud present -> snap
    sc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in base.items()}
    sc["cloud"] = np.concatenate([sc["cloud"], (sc["ref_path"][0, :3] + [0.05, 0.0, 0.0])[None].astype(np.float32)])
    scenes.append(sc)
    # (b) same but the edge cloud has one point only -> SearchForNearest(.,1) yields nothing -> unsafe
    sc2 = dict(sc); sc2["edge"] = sc["edge"][:1].copy()
    scenes.append(sc2)
    # (c) empty edge cloud
    sc3 = dict(sc); sc3["edge"] = np.zeros((0, 3), np.float32)
    scenes.append(sc3)
    # (d) obstacle cloud with exactly K points, (e) K+1 points, (f) empty
    for m in (prm.K, prm.K + 1, 0):
        s4 = dict(base); s4["cloud"] = base["cloud"][:m].copy()
        scenes.append(s4)
    # (g) free space far from everything: early exit after the first solve
    s5 = dict(base); s5["cloud"] = (base["cloud"] + np.float32([0, 100, 0])).astype(np.float32)
    scenes.append(s5)
    gpu, cpu = run_both(torch_cuda, scenes, prm, n_steps=1)
    compare(gpu, cpu)
    fl = gpu[0]["flags"]
    assert fl[0, 0] == 1 and fl[1, 0] == 0 and fl[2, 0] == 0          # isSafety
    assert fl[6, 1] == 1                                              # one solve, then :333-335 exit
    assert np.allclose(gpu[0]["ref_path"][0][0, :3], cpu[0][0]["ref_path"][0, :3])


def test_results_are_bit_reproducible(torch_cuda):
    """The LDS atomics of the objective evaluation are applied in lane order and every reduction has a fixed shape:
    the same inputs give the same bits, run after run."""
    prm = synth.MpcParams(T=0.66, K=8)
    scenes = [synth.make_scene(20000, 900 + i, prm) for i in range(6)]
    outs = []
    for _ in range(3):
        gpu, _cpu = run_both(torch_cuda, scenes, prm, n_steps=2)
        outs.append(np.concatenate([gpu[1]["u"].ravel(), gpu[1]["x0array"].ravel()]).view(np.uint64))
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
