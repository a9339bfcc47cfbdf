"""GPU: the CasADi code-generation symbols exported by libavoid_mpc_amd.so (the plugin
`casadi::nlpsol("solve", "ipopt", soPath, opts)` loads, AM/src/HighLvlMpc.cpp:50,52), driven through ctypes the way
CasADi's importer drives a generated library: the oracle `nlp` first, then every nlp_* function (names from
include/avoid_mpc_amd/casadi_plugin.h): *_n_in/_n_out/_name_*/_sparsity_*/_work, then the function itself with
`const double** arg, double** res, long long* iw, double* w, int mem`.  Values against the oracle.  CasADi itself is not
in the image: the calling convention is the one recalled in SURVEY.md appendix B, unverified against libcasadi.

One process holds one plugin configuration (N, K are baked into a generated plugin; here they come from AMK_MPC_T /
AMK_MPC_DT / AMK_MPC_K at first use), so every configuration runs in its own interpreter."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, ROOT)
from avoid_mpc_amd import capi, synth
from tests import _oracle
from tests.test_mpc_eval_gpu import _points
lib = capi.load()
cfg = sys.argv[1]
prm, W, R = _points(cfg, 1, 21)
N, K = prm.N, prm.K
nx, ng, npar = 10 + 14 * N, 10 + 10 * N, 54 + 10 * N + 3 * K * N
LL = C.c_longlong
def sparsity(fn, i):
    fn.restype = C.POINTER(LL); fn.argtypes = [LL]
    p = fn(i)
    nrow, ncol = p[0], p[1]
    if p[2] == 1 and ncol == 1 and True:      # dense shorthand of a column vector (colind would start with 0)
        return nrow, ncol, None, None
    colind = [p[2 + j] for j in range(ncol + 1)]
    row = [p[2 + ncol + 1 + j] for j in range(colind[-1])]
    return nrow, ncol, colind, row
def call(name, args, n_out_sizes):
    f = getattr(lib, name)
    f.restype = C.c_int
    f.argtypes = [C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)), C.POINTER(LL), C.POINTER(C.c_double), C.c_int]
    n_in = getattr(lib, name + "_n_in"); n_in.restype = LL
    n_out = getattr(lib, name + "_n_out"); n_out.restype = LL
    assert n_in() == len(args) and n_out() == len(n_out_sizes)
    sz = [LL(), LL(), LL(), LL()]
    assert getattr(lib, name + "_work")(*[C.byref(v) for v in sz]) == 0
    assert sz[0].value >= len(args) and sz[1].value >= len(n_out_sizes)
    argv = (C.POINTER(C.c_double) * max(sz[0].value, 1))()
    keep = [np.ascontiguousarray(a, np.float64) for a in args]
    for i, a in enumerate(keep):
        argv[i] = a.ctypes.data_as(C.POINTER(C.c_double))
    outs = [np.zeros(n) for n in n_out_sizes]
    resv = (C.POINTER(C.c_double) * max(sz[1].value, 1))()
    for i, o in enumerate(outs):
        resv[i] = o.ctypes.data_as(C.POINTER(C.c_double))
    iw = (LL * max(sz[2].value, 1))(); wk = (C.c_double * max(sz[3].value, 1))()
    getattr(lib, name + "_incref")()
    mem = getattr(lib, name + "_checkout")()
    assert f(argv, resv, iw, wk, mem) == 0
    getattr(lib, name + "_release")(mem)
    getattr(lib, name + "_decref")()
    return outs
dims = [C.c_int() for _ in range(5)]
assert lib.amk_plugin_dims(*[C.byref(d) for d in dims]) == 0
assert [d.value for d in dims] == [N, K, nx, npar, ng], [d.value for d in dims]
x = W[0]; p = np.concatenate([R[0], prm.gain, prm.tau, prm.weights, [prm.radius]])
o = _oracle.load_oracle()
res = {}
assert sparsity(lib.nlp_f_sparsity_in, 0)[:2] == (nx, 1) and sparsity(lib.nlp_f_sparsity_in, 1)[:2] == (npar, 1)
assert sparsity(lib.nlp_f_sparsity_out, 0)[:2] == (1, 1)
(f,) = call("nlp_f", [x, p], [1])
fo = o.mpco_nlp_f(np.ascontiguousarray(x), np.ascontiguousarray(p), N, K)
res["f"] = abs(f[0] - fo) / abs(fo)
f2, gr = call("nlp_grad_f", [x, p], [1, nx])
go = np.zeros(nx); o.mpco_nlp_grad_f(np.ascontiguousarray(x), np.ascontiguousarray(p), N, K, go)
res["grad"] = float(np.abs(gr - go).max() / np.abs(go).max()); assert f2[0] == f[0]
(g,) = call("nlp_g", [x, p], [ng])
cg = np.zeros(ng); o.mpco_nlp_g(np.ascontiguousarray(x), np.ascontiguousarray(p), N, K, prm.dt, cg)
res["g"] = float(np.abs(g - cg).max())
nrow, ncol, colind, row = sparsity(lib.nlp_jac_g_sparsity_out, 1)
assert (nrow, ncol) == (ng, nx) and colind[-1] == 10 + 39 * N
g2, jac = call("nlp_jac_g", [x, p], [ng, colind[-1]])
assert np.array_equal(g2, g)
import mpc_oracle_np as M
Jn = M.nlp_jac_g(x, p, N, K, prm.dt)
Jd = np.zeros((ng, nx))
for c in range(nx):
    Jd[row[colind[c]:colind[c + 1]], c] = jac[colind[c]:colind[c + 1]]
res["jac"] = float(np.abs(Jd - Jn).max())
nrow, ncol, colind, row = sparsity(lib.nlp_hess_l_sparsity_out, 0)
assert (nrow, ncol) == (nx, nx) and colind[-1] == 25 * (N - 1) + 10 + 4 * N
assert sparsity(lib.nlp_hess_l_sparsity_in, 2)[:2] == (1, 1) and sparsity(lib.nlp_hess_l_sparsity_in, 3)[:2] == (ng, 1)
lam_f = np.array([1.7]); lam_g = np.linspace(-1, 1, ng)
(hs,) = call("nlp_hess_l", [x, p, lam_f, lam_g], [colind[-1]])
Qs = np.zeros(N * 100); Rs = np.zeros(N * 4); o.mpco_nlp_hess_blocks(np.ascontiguousarray(x), np.ascontiguousarray(p), N, K, Qs, Rs, 0)
Hd = np.zeros((nx, nx))
for k in range(N):
    Hd[14 * (k + 1):14 * (k + 1) + 10, 14 * (k + 1):14 * (k + 1) + 10] = Qs[100 * k:100 * k + 100].reshape(10, 10)
    Hd[14 * k + 10:14 * k + 14, 14 * k + 10:14 * k + 14] = np.diag(Rs[4 * k:4 * k + 4])
Hg = np.zeros((nx, nx))
for c in range(nx):
    Hg[row[colind[c]:colind[c + 1]], c] = hs[colind[c]:colind[c + 1]]
res["hess"] = float(np.abs(Hg - 1.7 * np.triu(Hd)).max() / np.abs(Hd).max())
# a changed parameter tail (weights) is picked up on the next call
p2 = p.copy(); p2[-2] = 2.4      # collide_lambda
(f3,) = call("nlp_f", [x, p2], [1])
res["f_tail"] = abs(f3[0] - o.mpco_nlp_f(np.ascontiguousarray(x), np.ascontiguousarray(p2), N, K)) / abs(f3[0])
# ---- the oracle `nlp` (what nlpsol(name, "ipopt", "<file>.so") loads first) and NULL conventions
fa, ga = call("nlp", [x, p], [1, ng])
assert fa[0] == f[0] and np.array_equal(ga, g)
def raw(name, argl, outl):
    """argl / outl entries may be None (NULL pointer: all-zero input / output not requested)."""
    fn = getattr(lib, name); fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)), C.POINTER(LL), C.POINTER(C.c_double), C.c_int]
    argv = (C.POINTER(C.c_double) * len(argl))(); resv = (C.POINTER(C.c_double) * len(outl))()
    keep = [None if a is None else np.ascontiguousarray(a, np.float64) for a in argl]
    for i, a in enumerate(keep):
        if a is not None: argv[i] = a.ctypes.data_as(C.POINTER(C.c_double))
    for i, o_ in enumerate(outl):
        if o_ is not None: resv[i] = o_.ctypes.data_as(C.POINTER(C.c_double))
    assert fn(argv, resv, None, None, 0) == 0
gz = np.full(ng, 7.0); raw("nlp", [x, p], [None, gz]); assert np.array_equal(gz, g)          # res[0] == NULL: only g
fz = np.zeros(1); raw("nlp_f", [None, p], [fz]); (f0,) = call("nlp_f", [np.zeros(nx), p], [1]); assert fz[0] == f0[0]   # arg[0] == NULL: x = 0
hz = np.full(colind[-1], 3.0); raw("nlp_hess_l", [x, p, None, None], [hz]); assert not hz.any()   # lam_f == NULL: zero Hessian
# ---- every function, every helper: the walk a dlsym-based importer makes
names = {"nlp": (["x", "p"], ["f", "g"]), "nlp_f": (["x", "p"], ["f"]), "nlp_g": (["x", "p"], ["g"]),
         "nlp_grad_f": (["x", "p"], ["f", "grad_f_x"]), "nlp_jac_g": (["x", "p"], ["g", "jac_g_x"]),
         "nlp_hess_l": (["x", "p", "lam_f", "lam_g"], ["triu_hess_gamma_x_x"]),
         "nlp_grad": (["x", "p", "lam_f", "lam_g"], ["f", "g", "grad_gamma_x", "grad_gamma_p"])}
shape = {"x": (nx, 1), "p": (npar, 1), "f": (1, 1), "g": (ng, 1), "lam_f": (1, 1), "lam_g": (ng, 1), "grad_f_x": (nx, 1),
         "jac_g_x": (ng, nx), "triu_hess_gamma_x_x": (nx, nx), "grad_gamma_x": (nx, 1), "grad_gamma_p": (npar, 1)}
assert list(names) == capi.PLUGIN_FUNCTIONS
for fn, (ins, outs) in names.items():
    for h in capi.PLUGIN_HELPERS:
        assert hasattr(lib, fn + "_" + h), fn + "_" + h
    gi = getattr(lib, fn + "_n_in"); gi.restype = LL; go_ = getattr(lib, fn + "_n_out"); go_.restype = LL
    assert gi() == len(ins) and go_() == len(outs)
    for which, lst in (("in", ins), ("out", outs)):
        nm = getattr(lib, fn + "_name_" + which); nm.restype = C.c_char_p; nm.argtypes = [LL]
        for i, want in enumerate(lst):
            assert nm(i) == want.encode(), (fn, which, i, nm(i))
            assert sparsity(getattr(lib, fn + "_sparsity_" + which), i)[:2] == shape[want], (fn, which, i)
        assert nm(len(lst)) is None
    di = getattr(lib, fn + "_default_in"); di.restype = C.c_double; di.argtypes = [LL]; assert di(0) == 0.0
    assert getattr(lib, fn + "_alloc_mem")() == 0 and getattr(lib, fn + "_init_mem")(0) == 0
    getattr(lib, fn + "_free_mem")(0)
# ---- nlp_grad: gamma = lam_f f + lam_g' g
fg_, gg_, gx, gp = call("nlp_grad", [x, p, lam_f, lam_g], [1, ng, nx, npar])
assert fg_[0] == f[0] and np.array_equal(gg_, g)
gx_ref = 1.7 * go + Jn.T @ lam_g
res["grad_gamma_x"] = float(np.abs(gx - gx_ref).max() / np.abs(gx_ref).max())
gamma = lambda pp: 1.7 * M.nlp_f(x, pp, N, K) + lam_g @ M.nlp_g(x, pp, N, K, prm.dt)
fd = np.zeros(npar)
for i in range(npar):
    h = 1e-6 * max(1.0, abs(p[i])); e = np.zeros(npar); e[i] = h
    fd[i] = (gamma(p + e) - gamma(p - e)) / (2 * h)
scale = np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
res["grad_gamma_p"] = float((np.abs(gp - fd) / scale).max())
res["grad_gamma_p_blocks"] = {"x_init": float(np.abs(gp[:10] - fd[:10]).max()), "tau": float(np.abs(gp[-30:-26] - fd[-30:-26]).max()),
                              "weights": float((np.abs(gp[-26:-1] - fd[-26:-1]) / np.maximum(1.0, np.abs(fd[-26:-1]))).max()),
                              "radius": float(abs(gp[-1] - fd[-1]) / max(1.0, abs(fd[-1])))}
# ---- another horizon in the same process: amk_plugin_configure
lib.amk_plugin_configure.restype = C.c_int; lib.amk_plugin_configure.argtypes = [C.c_double, C.c_double, C.c_int]
assert lib.amk_plugin_configure(0.33, 0.033, 3) == 0
assert lib.amk_plugin_dims(*[C.byref(d) for d in dims]) == 0
assert [d.value for d in dims] == [10, 3, 150, 244, 110], [d.value for d in dims]
assert sparsity(lib.nlp_sparsity_in, 0)[:2] == (150, 1)
print("RESULT " + json.dumps(res))
'''


@pytest.mark.parametrize("cfg,T,K", [("C2", 0.66, 8), ("C5", 1.0, 8)])
def test_plugin_symbols(cfg, T, K):
    env = dict(os.environ, AMK_MPC_T=str(T), AMK_MPC_DT="0.033", AMK_MPC_K=str(K))
    code = "ROOT = %r\n" % ROOT + "import sys; sys.path.insert(0, ROOT + '/oracle')\n" + CHILD
    r = subprocess.run([sys.executable, "-c", code, cfg], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    print(cfg, res)
    assert res["f"] <= 1e-9 and res["grad"] <= 1e-9 and res["hess"] <= 1e-9 and res["f_tail"] <= 1e-9
    assert res["g"] <= 1e-11 and res["jac"] <= 1e-14
    assert res["grad_gamma_x"] <= 1e-12
    assert res["grad_gamma_p"] <= 1e-5, res      # against central differences of the numpy twin over the whole of p
