"""GPU: the short-dependency-chain exp / log / reciprocal of csrc/fast_math.h (used by the fp64 solve for the collision
terms and the barrier sums) against the host's libm: accuracy in ulp over the ranges the solver feeds them, and the
edge behaviour the call sites rely on (exp of a very negative argument -> 0 so that 1 + e^x == 1 and the naive softplus
of the reference, mpc_obstacle_casadi.py:250-251, is exactly 0 for far obstacles)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _probe(x):
    import torch
    from avoid_mpc_amd import capi
    lib = capi.load()
    lib.amk__fast_math_probe.restype = C.c_int
    lib.amk__fast_math_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    d = torch.from_numpy(np.ascontiguousarray(x, np.float64)).cuda()
    out = torch.empty(3 * len(x), dtype=torch.float64, device="cuda")
    assert lib.amk__fast_math_probe(C.c_void_p(d.data_ptr()), C.c_void_p(out.data_ptr()), len(x), None) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(3, -1)
    return o[0], o[1], o[2]


def _ulp_err(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_accuracy_in_ulp():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 12, 400000),                 # exp: -32 (rho - r) around the active range
                        rng.uniform(-745, 709, 100000),
                        np.exp(rng.uniform(-60, 60, 400000)) * rng.choice([-1.0, 1.0], 400000),   # log |x|, 1/x
                        1.0 + np.exp(rng.uniform(-40, 3, 100000))])                              # log(1 + e^x)
    e, l, r = _probe(x)
    with np.errstate(over="ignore"):
        we = np.exp(x)
    ok = np.isfinite(we) & (we > 1e-300)
    ue, ul, ur = _ulp_err(e[ok], we[ok]).max(), _ulp_err(l, np.log(np.abs(x))).max(), _ulp_err(r, 1.0 / x).max()
    print(f"max error in ulp: exp {ue:.2f}  log {ul:.2f}  rcp {ur:.2f}")
    assert ue <= 2.0 and ul <= 2.0 and ur <= 1.0


def test_device_and_cpu_restatement_give_the_same_bits():
    """oracle/mpc_oracle.c restates the two algorithms step by step (sexp / slog): identical results, except where the
    device's reciprocal (v_rcp_f64 + one cubic correction, <= 1 ulp) is not the correctly rounded quotient -- rare."""
    from tests import _oracle
    lib = _oracle.load_oracle()
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-60, 12, 300000), np.exp(rng.uniform(-40, 40, 300000)), 1.0 + np.exp(rng.uniform(-40, 3, 100000))])
    e, l, _ = _probe(x)
    ce, cl = np.empty_like(x), np.empty_like(x)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.mpco_fast_math.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]; lib.mpco_fast_math.restype = None
    lib.mpco_fast_math(vp(x), vp(ce), vp(cl), len(x))
    same_e = np.mean(e.view(np.int64) == ce.view(np.int64)); same_l = np.mean(l.view(np.int64) == cl.view(np.int64))
    print(f"bit-identical: exp {same_e:.6f}  log {same_l:.6f}")
    assert same_e == 1.0 and same_l >= 0.999
    assert _ulp_err(l, cl).max() <= 1.0


def test_edges_the_call_sites_rely_on():
    x = np.array([-1e6, -3.2e5, -800.0, -745.0, -40.0, -36.8, 0.0, 5e-324, 1.0, np.inf, 709.0, 1e-310])
    e, l, r = _probe(x)
    assert np.all(e[:3] == 0.0) or np.all(e[:3] < 1e-320)            # far obstacles (padding points at 1e4): e^x -> 0
    assert np.all(1.0 + e[:6] == 1.0)                                 # so log(1 + e^x) == 0 exactly: a dormant term
    assert l[8] == 0.0 and e[6] == 1.0                                # log(1) = 0, exp(0) = 1 exactly
    assert np.isclose(e[10], np.exp(709.0), rtol=1e-15)
    assert l[9] == np.inf and np.isclose(l[7], np.log(5e-324)) and np.isclose(l[11], np.log(1e-310))   # library fallback
