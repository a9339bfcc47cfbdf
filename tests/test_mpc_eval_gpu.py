"""GPU: the plugin's NLP functions on the device (amk_mpc_eval) against the oracle's
restatement of mpc_obstacle_casadi.py:153-219 at random points, for the three BASELINE sizes.

Tolerance (fp64): same formulas, different summation order (LDS atomics / wave reductions) and ocml vs libm exp/log:
1e-9 relative to the largest entry of each output (measured ~1e-13)."""
import os
import sys

import numpy as np
import pytest

from avoid_mpc_amd import synth
from tests import _oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import mpc_oracle_np as M  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _points(cfg, S, seed):
    """S random (w, P-prefix) pairs around a plausible flight state, obstacles close enough to matter."""
    c = synth.CONFIGS[cfg]
    prm = synth.MpcParams(T=c["T"], K=c["K"])
    N, K = prm.N, prm.K
    rng = np.random.default_rng(seed)
    nref = 20 + 10 * N + 3 * K * N
    W = np.zeros((S, 10 + 14 * N)); R = np.zeros((S, nref))
    for s in range(S):
        X = rng.normal(size=(N + 1, 10)); X[:, 0] += np.arange(N + 1) * 0.33; X[:, 2] += 1.5; X[:, 4] += 8.0
        U = rng.normal(size=(N, 4)) * 3.0 + np.array([0, 0, 9.81, 0])
        W[s] = M.pack_w(X, U)
        ref = X[1:].copy() + rng.normal(size=(N, 10)) * 0.2; ref[:, 3] = rng.uniform(-0.6, 0.6, N)
        obs = X[1:, None, 0:3] + rng.normal(size=(N, K, 3)) * 0.5
        obs[rng.random((N, K)) < 0.2] = 1e4                                      # padding, AvoidanceStateMachine.cpp:223-226
        R[s] = np.concatenate([X[0] + rng.normal(size=10) * 0.01, ref.reshape(-1), obs.reshape(-1), ref[-1] + 1.0])
    return prm, W, R


@pytest.mark.parametrize("cfg", ["C1", "C2", "C5"])
def test_eval_matches_oracle(cfg):
    import torch
    from avoid_mpc_amd.host import MpcBatch
    S = 24
    prm, W, R = _points(cfg, S, 5)
    N, K = prm.N, prm.K
    nx, ng = 10 + 14 * N, 10 + 10 * N
    m = MpcBatch(prm.T, prm.dt, prm.K, S); m.configure(prm)
    lam_f = np.linspace(0.5, 2.0, S)
    out = m.eval(torch.from_numpy(W).cuda(), torch.from_numpy(R).cuda(), torch.from_numpy(lam_f).cuda())
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    jc, jr = m.sparsity("jac_g"); hc, hr = m.sparsity("hess_l")
    assert len(jr) == 10 + 39 * N and len(hr) == 25 * (N - 1) + 10 + 4 * N
    lib = _oracle.load_oracle()
    for s in range(S):
        P = np.ascontiguousarray(np.concatenate([R[s], prm.gain, prm.tau, prm.weights, [prm.radius]]))
        w = np.ascontiguousarray(W[s])
        f = lib.mpco_nlp_f(w, P, N, K)
        assert abs(out["f"][s] - f) <= RTOL * abs(f)
        g = np.zeros(nx); lib.mpco_nlp_grad_f(w, P, N, K, g)
        assert np.abs(out["grad_f"][s] - g).max() <= RTOL * np.abs(g).max()
        cg = np.zeros(ng); lib.mpco_nlp_g(w, P, N, K, prm.dt, cg)
        assert np.abs(out["g"][s] - cg).max() <= 1e-12 * max(1.0, np.abs(w).max())
        # Jacobian: CCS -> dense against the numpy twin's dense Jacobian, pattern included
        Jd = np.zeros((ng, nx))
        for col in range(nx):
            rows = jr[jc[col]:jc[col + 1]]
            assert np.all(np.diff(rows) > 0)
            Jd[rows, col] = out["jac_g"][s][jc[col]:jc[col + 1]]
        Jn = M.nlp_jac_g(w, P, N, K, prm.dt)
        assert np.abs(Jd - Jn).max() <= 1e-14
        if s == 0:
            assert np.array_equal(Jd != 0, np.abs(Jn) > 0)
        # Hessian: lam_f * upper triangle of the block-diagonal Hessian of f (exact: sign frozen, no majoriser)
        Qs = np.zeros(N * 100); Rs = np.zeros(N * 4); lib.mpco_nlp_hess_blocks(w, P, N, K, Qs, Rs, 0)
        Hd = np.zeros((nx, nx))
        for k in range(N):
            Hd[14 * (k + 1):14 * (k + 1) + 10, 14 * (k + 1):14 * (k + 1) + 10] = Qs[100 * k:100 * k + 100].reshape(10, 10)
            Hd[14 * k + 10:14 * k + 14, 14 * k + 10:14 * k + 14] = np.diag(Rs[4 * k:4 * k + 4])
        Hg = np.zeros((nx, nx))
        for col in range(nx):
            rows = hr[hc[col]:hc[col + 1]]
            assert np.all(np.diff(rows) > 0) and np.all(rows <= col)
            Hg[rows, col] = out["hess_l"][s][hc[col]:hc[col + 1]]
        assert np.abs(Hg - lam_f[s] * np.triu(Hd)).max() <= RTOL * np.abs(Hd).max()
        # nothing of the Hessian falls outside the declared pattern
        mask = np.zeros((nx, nx), bool)
        for col in range(nx):
            mask[hr[hc[col]:hc[col + 1]], col] = True
        assert np.all(np.triu(Hd)[~mask] == 0)


def test_eval_host_and_null_outputs():
    """The host-buffer convenience with only some outputs requested (S = 1: what the plugin symbols call)."""
    import ctypes as C
    from avoid_mpc_amd import capi
    prm, W, R = _points("C1", 1, 9)
    lib = capi.load()
    h = C.c_void_p()
    assert lib.amk_mpc_create(prm.T, prm.dt, prm.K, 1, C.byref(h)) == 0
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for name, val in (("weights", prm.weights), ("tau", prm.tau), ("gains", prm.gain)):
        assert getattr(lib, "amk_mpc_setup_" + name)(h, vp(np.ascontiguousarray(val, np.float64))) == 0
    assert lib.amk_mpc_set_drone_radius(h, prm.radius) == 0
    f = np.zeros(1); g = np.zeros(10 + 14 * prm.N)
    assert lib.amk_mpc_eval_host(h, vp(W), vp(R), None, vp(f), vp(g), None, None, None) == 0
    P = np.ascontiguousarray(np.concatenate([R[0], prm.gain, prm.tau, prm.weights, [prm.radius]]))
    assert abs(f[0] - _oracle.load_oracle().mpco_nlp_f(np.ascontiguousarray(W[0]), P, prm.N, prm.K)) <= RTOL * abs(f[0])
    assert lib.amk_mpc_eval_host(h, None, vp(R), None, vp(f), None, None, None, None) == 1   # AMK_ERR_INVALID_ARG
    assert lib.amk_mpc_destroy(h) == 0
