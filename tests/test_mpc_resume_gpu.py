"""GPU: the resumable solve (amk_mpc_set_solve_budget).  A control step whose solve launches end after B interior-point
iterations per scene -- unfinished solves paused, resumed inside the next round's launch, scenes in different re-plan passes
side by side -- must return the BITS of the plain schedule (one round per pass, every launch run to convergence): the budget
is scheduling, not arithmetic.  Reference semantics kept per scene: one Solve per pass run to convergence, warm start in / out
(AM/src/HighLvlMpc.cpp:93-137), passes in order (AM/src/AvoidanceStateMachine.cpp:322-344)."""
import numpy as np
import pytest

from avoid_mpc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _steps(torch, scenes, prm, budget, rounds, n_steps=2, precision=64):
    from tests import _oracle
    from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch
    S = len(scenes)
    n, ne = len(scenes[0]["cloud"]), len(scenes[0]["edge"])
    kd_o, kd_e = KdBatch(S, n), KdBatch(S, ne)
    kd_o.build(torch.from_numpy(np.stack([sc["cloud"] for sc in scenes])).cuda())
    kd_e.build(torch.from_numpy(np.stack([sc["edge"] for sc in scenes])).cuda())
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    mpc.set_precision(precision)
    mpc.set_solve_budget(budget, rounds)
    sq = torch.from_numpy(np.stack([_oracle.scene_state_quads(sc, prm) for sc in scenes])).cuda()
    ref = torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()
    pos_x = torch.from_numpy(np.array([sc["pos"][0] for sc in scenes])).cuda()
    out = []
    for _ in range(n_steps):   # second step: warm start carried over (mNlpW0 = sol)
        o = step_batch(kd_o, kd_e, mpc, prm, sq, pos_x, ref)
        torch.cuda.synchronize()
        out.append({k: o[k].cpu().numpy().copy() for k in ("u", "x0array", "flags")} | {"ref_path": ref.cpu().numpy().copy(),
                                                                                         "w": mpc.get_warm_start().cpu().numpy().copy()})
    for h in (kd_o, kd_e, mpc):
        h.close()
    return out


@pytest.mark.parametrize("cfg,n,S", [("C2", 50000, 48), ("C1", 5000, 32), ("C5", 20000, 16)])
def test_budgeted_step_returns_the_bits_of_the_plain_schedule(cfg, n, S, torch_cuda):
    c = synth.CONFIGS[cfg]
    prm = synth.MpcParams(T=c["T"], K=c["K"])
    scenes = [synth.make_scene(n, 900 + i, prm) for i in range(S)]
    plain = _steps(torch_cuda, scenes, prm, 0, 0)
    it = plain[0]["flags"][:, 3]
    assert (plain[0]["flags"][:, 1] >= 2).sum() >= (S // 2 if cfg != "C1" else 4), "the workload must re-plan for the test to mean anything"
    paused_somewhere = False
    for budget, rounds in ((2, 0), (5, 0), (5, 6), (1, 16), (13, 1), (100, 0)):
        got = _steps(torch_cuda, scenes, prm, budget, rounds)
        for t in range(len(plain)):
            for k in plain[t]:
                assert np.array_equal(got[t][k].view(np.int64) if got[t][k].dtype == np.float64 else got[t][k],
                                      plain[t][k].view(np.int64) if plain[t][k].dtype == np.float64 else plain[t][k]), (cfg, budget, rounds, t, k)
        paused_somewhere = paused_somewhere or budget * 3 < it.max()
    assert paused_somewhere
    print(f"{cfg}: {S} scenes, iterations per step min / mean / max {it.min()} / {it.mean():.1f} / {it.max()}, "
          f"solves per step {plain[0]['flags'][:, 1].mean():.2f}: budgets 1 ... 100 bit-identical")


def test_budgeted_step_fp32_and_argument_errors(torch_cuda):
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import MpcBatch
    prm = synth.MpcParams(T=0.66, K=8)
    scenes = [synth.make_scene(20000, 950 + i, prm) for i in range(16)]
    a = _steps(torch_cuda, scenes, prm, 0, 0, n_steps=1, precision=32)
    b = _steps(torch_cuda, scenes, prm, 3, 0, n_steps=1, precision=32)
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k], equal_nan=True), k
    m = MpcBatch(prm.T, prm.dt, prm.K, 2)
    lib = capi.load()
    assert lib.amk_mpc_set_solve_budget(m.h, -1, 0) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_mpc_set_solve_budget(m.h, 4, 17) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_mpc_set_solve_budget(None, 4, 0) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_mpc_set_solve_budget(m.h, 4, 16) == 0
    m.close()
