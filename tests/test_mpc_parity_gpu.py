"""GPU: the MPC parity gate on the HIP solve with the SHIPPED options, for the 64 bench scenes of
every BASELINE size: all 64 converged, |u - u*|_inf <= 1e-3 on all of them and |x - x*|_inf <= 1e-3 on >= 98 %, against the converged local
optimum of tests/golden/mpc_parity_golden.npz (interior point to the rounding floor, cross-checked by scipy L-BFGS-B; see
tests/test_mpc_parity.py).  The objective of the returned point comes from the library's own nlp_f (amk_mpc_eval)."""
import numpy as np
import pytest

from avoid_mpc_amd import synth
from tests.test_mpc_parity import G, assert_gate, gate_report

pytestmark = pytest.mark.gpu


def solve_fixture_on_gpu(torch, cfg, precision=64):
    from avoid_mpc_amd.host import MpcBatch
    c = synth.CONFIGS[cfg]
    prm = synth.MpcParams(T=c["T"], K=c["K"])
    refs = torch.from_numpy(G[cfg + ".ref"]).cuda()
    m = MpcBatch(prm.T, prm.dt, prm.K, refs.shape[0]); m.configure(prm); m.set_precision(precision)
    u, x0, info = m.Solve(refs, faster=True)          # zero warm start (constructor), shipped options
    w = m.get_warm_start()
    J = m.eval(w, refs, want=("f",))["f"]
    torch.cuda.synchronize()
    return u.cpu().numpy(), w.cpu().numpy(), J.cpu().numpy(), info.cpu().numpy()


@pytest.mark.parametrize("cfg", ["C1", "C2", "C5"])
def test_gpu_solve_meets_the_gate(cfg):
    import torch
    u, w, J, info = solve_fixture_on_gpu(torch, cfg)
    assert np.array_equal(u, w[:, 10:14])
    rep = gate_report(cfg, u, w, J, info[:, 0])
    print(cfg, rep, "iterations mean %.1f max %d, converged %d/%d" % (info[:, 1].mean(), info[:, 1].max(),
                                                                     int((info[:, 0] == 0).sum()), len(info)))
    assert_gate(rep)
