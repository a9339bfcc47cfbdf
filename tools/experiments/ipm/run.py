"""HARNESS (CPU only).  proto_solve (proto.c) on the dumped problems of the bench step: iterations by pass against the
shipped algorithm's, and whether the prototype stops at the shipped algorithm's optimum (|du|, |dw| <= 1e-3)."""
import sys, os, ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from avoid_mpc_amd import synth
from tests import _oracle

def build(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "scratch/ipm")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(ROOT, "tools/experiments/ipm/proto.c"); so = os.path.join(out_dir, "libproto.so")
    subprocess.check_call(["gcc", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return C.CDLL(so)

class ProtoOpts(C.Structure):
    _fields_ = [("diag", C.c_int), ("variant", C.c_int), ("p", C.c_double * 8)]

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"); _i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

class Bench:
    def __init__(self, path=os.path.join(ROOT, "scratch/ipm/problems_c2.npz"), out_dir=None):
        self.lib = build(out_dir)
        self.lib.proto_solve.restype = C.c_int
        self.lib.proto_solve.argtypes = [_f64p, _f64p, _f64p, _f64p, C.c_int, C.c_int, C.c_double, C.POINTER(_oracle.MpcoOpts),
                                         C.POINTER(ProtoOpts), _f64p, _i32p, _f64p]
        D = np.load(path); self.D = D
        self.prm = prm = synth.MpcParams(T=float(D["T"]), K=int(D["K"])); self.N, self.K = prm.N, prm.K
        self.lbu = np.array([-prm.a_max_xy, -prm.a_max_xy, prm.a_min_z, -prm.a_max_yaw_dot]); self.ubu = np.array([prm.a_max_xy, prm.a_max_xy, prm.a_max_z, prm.a_max_yaw_dot])
        self.P = [np.concatenate([r, prm.gain, prm.tau, prm.weights, [prm.radius]]) for r in D["ref"]]
        self.w0 = D["w0"]; self.pas = D["pas"]
        self.base = None

    def solve(self, idx, variant=0, p=(), diag=0, **kw):
        ol = _oracle.load_oracle()
        opt = _oracle.MpcoOpts(); ol.mpco_default_opts(C.byref(opt))
        for k, v in kw.items(): setattr(opt, k, v)
        po = ProtoOpts(); po.diag = diag; po.variant = variant
        for i, v in enumerate(p): po.p[i] = v
        w = np.zeros(10 + 14 * self.N); info = np.zeros(8, np.int32); st = np.zeros(4)
        self.lib.proto_solve(self.P[idx], np.ascontiguousarray(self.w0[idx]), self.lbu, self.ubu, self.N, self.K, self.prm.dt, C.byref(opt), C.byref(po), w, info, st)
        return w, info, st

    def run(self, variant=0, p=(), label="", **kw):
        n = len(self.P)
        W = np.zeros((n, 10 + 14 * self.N)); I = np.zeros((n, 8), np.int32); J = np.zeros(n)
        for i in range(n):
            W[i], I[i], st = self.solve(i, variant, p, **kw); J[i] = st[0]
        if self.base is None and variant == 0 and not kw:
            self.base = (W.copy(), I.copy(), J.copy())
        out = "%-34s" % label
        for ps in range(3):
            m = self.pas == ps
            out += " | p%d it %5.2f reg %4.2f ls %5.2f" % (ps, I[m, 1].mean(), I[m, 2].mean(), I[m, 4].mean())
        S = len(np.unique(self.D["scene"]))
        out += " | /step it %5.1f sweeps %5.1f evals %5.1f | conv %d/%d" % (I[:, 1].sum() / S, (I[:, 1] + I[:, 2]).sum() / S, I[:, 4].sum() / S, (I[:, 0] == 0).sum(), n)
        if self.base is not None:
            du = np.abs(W[:, 10:14] - self.base[0][:, 10:14]).max(axis=1); dw = np.abs(W - self.base[0]).max(axis=1)
            out += " | du>1e-3: %d dw>1e-3: %d (du med %.1e max %.1e) J lower/higher %d/%d" % ((du > 1e-3).sum(), (dw > 1e-3).sum(), np.median(du), du.max(),
                    (J < self.base[2] * (1 - 1e-6)).sum(), (J > self.base[2] * (1 + 1e-6)).sum())
        print(out, flush=True)
        return W, I, J

if __name__ == "__main__":
    b = Bench()
    W, I, J = b.run(label="shipped (proto, all switches off)")
    # the copy is the oracle: same bits
    ok = 0
    for i in range(len(b.P)):
        w, info, st = _oracle.mpco_solve(b.P[i], b.w0[i], b.lbu, b.ubu, b.N, b.K, b.prm.dt)
        ok += int(np.array_equal(w, W[i]) and info[1] == I[i, 1])
    print("bit-identical to mpco_solve on", ok, "of", len(b.P))
