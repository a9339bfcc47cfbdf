"""GPU: the generator's use_drag_coefficient switch (mpc_obstacle_casadi.py:95-105, yaml :4, off by default) in the reading under which it
is defined -- `rotmat * diag(k, k, k) * rotmat.T * v` as matrix products = k v (tests/test_mpc_oracle.py proves the identity on the
generator's own acc2rotmat), i.e. v' = a - k .* v: amk_mpc_set_drag_coefficient.  The dynamics stay affine with the same sparsity, so the
solver, the plugin functions and the control step are the same code on another A: constraints g and the Jacobian's VALUES against the
oracle with the same switch, the solve and a whole control step against the oracle's, and the default (0, 0, 0) == never having called it."""
import os
import sys

import numpy as np
import pytest

from avoid_mpc_amd import synth
from tests import _oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import mpc_oracle_np as M  # noqa: E402

pytestmark = pytest.mark.gpu
K_REF = 0.033   # the generator's hard-coded coefficient


def test_plugin_functions_with_drag_match_the_oracle():
    import torch
    from avoid_mpc_amd.host import MpcBatch
    from tests.test_mpc_eval_gpu import _points
    S = 8
    prm, W, R = _points("C2", S, 11)
    N, K = prm.N, prm.K
    nx, ng = 10 + 14 * N, 10 + 10 * N
    m = MpcBatch(prm.T, prm.dt, prm.K, S); m.configure(prm)
    base = {k: v.cpu().numpy() for k, v in m.eval(torch.from_numpy(W).cuda(), torch.from_numpy(R).cuda()).items()}
    m.SetDragCoefficient(K_REF)
    out = {k: v.cpu().numpy() for k, v in m.eval(torch.from_numpy(W).cuda(), torch.from_numpy(R).cuda()).items()}
    jc, jr = m.sparsity("jac_g")
    assert len(jr) == 10 + 39 * N                                    # the pattern does not change
    lib = _oracle.load_oracle()
    with _oracle.oracle_drag(K_REF):
        for s in range(S):
            P = np.ascontiguousarray(np.concatenate([R[s], prm.gain, prm.tau, prm.weights, [prm.radius]]))
            w = np.ascontiguousarray(W[s])
            cg = np.zeros(ng); lib.mpco_nlp_g(w, P, N, K, prm.dt, cg)
            assert np.abs(out["g"][s] - cg).max() <= 1e-12 * max(1.0, np.abs(w).max())
            Jd = np.zeros((ng, nx))
            for col in range(nx):
                Jd[jr[jc[col]:jc[col + 1]], col] = out["jac_g"][s][jc[col]:jc[col + 1]]
            assert np.abs(Jd - M.nlp_jac_g(w, P, N, K, prm.dt)).max() <= 1e-14
    assert np.abs(out["g"] - base["g"]).max() > 1e-4                 # the switch does something ...
    assert np.array_equal(out["f"], base["f"]) and np.array_equal(out["grad_f"], base["grad_f"])   # ... to the dynamics only
    m.SetDragCoefficient(0.0)
    again = {k: v.cpu().numpy() for k, v in m.eval(torch.from_numpy(W).cuda(), torch.from_numpy(R).cuda()).items()}
    assert all(np.array_equal(again[k], base[k]) for k in base)      # off == the default, bit for bit
    lib_p = m.lib
    assert lib_p.amk_mpc_set_drag_coefficient(m.h, -1.0, 0.0, 0.0) != 0 and lib_p.amk_mpc_set_drag_coefficient(m.h, float("nan"), 0.0, 0.0) != 0


def test_solve_and_control_step_with_drag_match_the_oracle():
    import torch
    from dataclasses import replace
    from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch
    from avoid_mpc_amd import fsm
    prm = synth.MpcParams(T=0.66, K=8)
    prm_d = replace(prm, drag=(K_REF, K_REF, K_REF))
    S, n = 12, 8000
    scenes = [synth.make_scene(n, 700 + s, prm) for s in range(S)]
    sq = np.stack([_oracle.scene_state_quads(sc, prm) for sc in scenes])
    posx = np.array([sc["pos"][0] for sc in scenes]); ref0 = np.stack([sc["ref_path"] for sc in scenes])
    kd_o, kd_e = KdBatch(S, n), KdBatch(S, n // 10)
    kd_o.build(torch.from_numpy(np.stack([sc["cloud"] for sc in scenes])).cuda()); kd_e.build(torch.from_numpy(np.stack([sc["edge"] for sc in scenes])).cuda())
    outs = {}
    for name, p in (("off", prm), ("drag", prm_d)):
        mpc = MpcBatch(p.T, p.dt, p.K, S); mpc.configure(p)
        ref = torch.from_numpy(ref0.copy()).cuda()
        o = step_batch(kd_o, kd_e, mpc, p, torch.from_numpy(sq).cuda(), torch.from_numpy(posx).cuda(), ref)
        torch.cuda.synchronize()
        outs[name] = {k: v.cpu().numpy() for k, v in o.items()}
    assert np.abs(outs["drag"]["u"] - outs["off"]["u"]).max() > 1e-3          # another problem: the vehicle decelerates on its own
    worst = 0.0; flipped = 0
    with _oracle.oracle_drag(K_REF):
        for s, sc in enumerate(scenes):
            ko, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
            mo = _oracle.MpcOracle(prm.T, prm.dt, prm.K); mo.configure(prm)
            r = _oracle.step_oracle(ko, ke, mo, prm, sq[s], posx[s], ref0[s].copy())
            if np.array_equal(r["flags"], outs["drag"]["flags"][s]):
                worst = max(worst, np.abs(r["u"] - outs["drag"]["u"][s]).max(), np.abs(r["x0array"] - outs["drag"]["x0array"][s]).max())
            else:
                flipped += 1   # a rounding-level branch flip (tests/test_step_gpu.py): same optimum, other counts
                assert r["flags"][0] == outs["drag"]["flags"][s][0] and np.abs(r["u"] - outs["drag"]["u"][s]).max() <= 1e-4
    print(f"control step with drag {K_REF}: |gpu - oracle| <= {worst:.2e}, flipped {flipped}/{S}")
    assert worst <= 1e-6 and flipped <= 1
