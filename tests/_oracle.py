"""ctypes access to the oracle (oracle/liboracle.so) and to oracle/_ref.  TEST INFRASTRUCTURE ONLY:
imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the
product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build_oracle(force=False):
    """(Re)build oracle/liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(ORACLE_DIR, "_ref", "libnanoflann_ref_strict.so")
    if os.path.isdir("/root/reference") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    return so


class KdHandle:
    """One tree of either flavour (prefix 'kdo' = C restatement, 'ref_kd' = reference header)."""

    def __init__(self, lib, prefix, xyz, stride=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if stride is None:
            stride = xyz.shape[1]
        self.lib, self.p = lib, prefix
        self.h = getattr(lib, prefix + "_create")(xyz.reshape(-1), xyz.shape[0], stride)

    def size(self):
        return getattr(self.lib, self.p + "_size")(self.h)

    def search(self, q, n):
        """KDTreeTwo::SearchForNearest semantics -> (indices int32, sqdist f64, pts f32[.,3])."""
        m = max(n, 1)
        idx = np.zeros(m, np.int32); d2 = np.zeros(m, np.float64); pts = np.zeros(3 * m, np.float32)
        cnt = getattr(self.lib, self.p + "_search")(self.h, q[0], q[1], q[2], n, idx, d2, pts)
        return idx[:cnt].copy(), d2[:cnt].copy(), pts[:3 * cnt].reshape(-1, 3).copy()

    def search_raw(self, q, n):
        m = max(n, 1)
        idx = np.zeros(m, np.int32); d2 = np.zeros(m, np.float64)
        cnt = getattr(self.lib, self.p + "_search_raw")(self.h, q[0], q[1], q[2], n, idx, d2)
        return idx[:cnt].copy(), d2[:cnt].copy()

    def bruteforce(self, q, k):
        m = max(k, 1)
        idx = np.zeros(m, np.int32); d2 = np.zeros(m, np.float64)
        cnt = self.lib.kdo_bruteforce(self.h, q[0], q[1], q[2], k, idx, d2)
        return idx[:cnt].copy(), d2[:cnt].copy()

    def rebuild(self, reps):
        getattr(self.lib, self.p + "_rebuild")(self.h, reps)

    def close(self):
        if self.h:
            getattr(self.lib, self.p + "_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _decl_kd(lib, p):
    getattr(lib, p + "_create").restype = C.c_void_p
    getattr(lib, p + "_create").argtypes = [_f32p, C.c_int, C.c_int]
    getattr(lib, p + "_size").restype = C.c_int
    getattr(lib, p + "_size").argtypes = [C.c_void_p]
    getattr(lib, p + "_search").restype = C.c_int
    getattr(lib, p + "_search").argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, _i32p, _f64p, _f32p]
    getattr(lib, p + "_search_raw").restype = C.c_int
    getattr(lib, p + "_search_raw").argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, _i32p, _f64p]
    getattr(lib, p + "_rebuild").restype = None
    getattr(lib, p + "_rebuild").argtypes = [C.c_void_p, C.c_int]
    getattr(lib, p + "_destroy").restype = None
    getattr(lib, p + "_destroy").argtypes = [C.c_void_p]


_ORACLE = None


def load_oracle():
    global _ORACLE
    if _ORACLE is None:
        lib = C.CDLL(build_oracle())
        _decl_kd(lib, "kdo")
        lib.kdo_bruteforce.restype = C.c_int
        lib.kdo_bruteforce.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, _i32p, _f64p]
        if hasattr(lib, "mpco_p_len"):
            _decl_mpc(lib)
        _ORACLE = lib
    return _ORACLE


def load_ref(strict=True):
    name = "libnanoflann_ref_strict.so" if strict else "libnanoflann_ref.so"
    path = os.path.join(ORACLE_DIR, "_ref", name)
    if not os.path.exists(path):
        if os.path.isdir("/root/reference"):
            build_oracle()
        if not os.path.exists(path):
            return None
    lib = C.CDLL(path)
    _decl_kd(lib, "ref_kd")
    return lib


def kd_oracle(xyz, stride=None):
    return KdHandle(load_oracle(), "kdo", xyz, stride)


def kd_ref(xyz, stride=None, strict=True):
    lib = load_ref(strict)
    return None if lib is None else KdHandle(lib, "ref_kd", xyz, stride)


def _decl_mpc(lib):
    pass
