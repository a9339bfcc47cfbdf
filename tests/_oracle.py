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

    def keyframe_sweep(self, current, th_dist, th_count):
        """self = last keyframe's tree (kdo only).  -> (rebuilt, n_outliers)"""
        n = C.c_int(0)
        r = self.lib.kdo_keyframe_sweep(self.h, current.h, th_dist, th_count, C.byref(n))
        return r, n.value

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
        lib.kdo_keyframe_sweep.restype = C.c_int
        lib.kdo_keyframe_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int)]
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


class MpcoOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int), ("mu_init", C.c_double),
                ("bound_push", C.c_double), ("bound_frac", C.c_double), ("kappa_mu", C.c_double),
                ("theta_mu", C.c_double), ("kappa_eps", C.c_double), ("mu_min_fac", C.c_double), ("maj", C.c_double),
                ("tau_min", C.c_double), ("eta_phi", C.c_double), ("max_ls", C.c_int),
                ("s_max", C.c_double), ("kappa_sigma", C.c_double), ("trace", C.POINTER(C.c_double))]


def _decl_mpc(lib):
    i, d, vp = C.c_int, C.c_double, C.c_void_p
    lib.mpco_default_opts.argtypes = [C.POINTER(MpcoOpts)]
    lib.mpco_p_len.restype = i; lib.mpco_p_len.argtypes = [i, i]
    lib.mpco_rk4_step.argtypes = [_f64p, _f64p, _f64p, d, _f64p]
    lib.mpco_affine.argtypes = [_f64p, d, _f64p, _f64p, _f64p]
    lib.mpco_nlp_f.restype = d; lib.mpco_nlp_f.argtypes = [_f64p, _f64p, i, i]
    lib.mpco_nlp_grad_f.argtypes = [_f64p, _f64p, i, i, _f64p]
    lib.mpco_nlp_hess_blocks.argtypes = [_f64p, _f64p, i, i, _f64p, _f64p, i]
    lib.mpco_nlp_g.argtypes = [_f64p, _f64p, i, i, d, _f64p]
    lib.mpco_solve.restype = i
    lib.mpco_solve.argtypes = [_f64p, _f64p, _f64p, _f64p, i, i, d, C.POINTER(MpcoOpts), _f64p, _i32p, _f64p]
    lib.mpco_create.restype = vp; lib.mpco_create.argtypes = [d, d, i]
    lib.mpco_destroy.argtypes = [vp]
    lib.mpco_horizon.restype = i; lib.mpco_horizon.argtypes = [vp]
    for n in ("weights", "tau", "gains"):
        getattr(lib, "mpco_setup_" + n).argtypes = [vp, _f64p]
    lib.mpco_set_drone_radius.argtypes = [vp, d]
    lib.mpco_set_drone_accel_limits.argtypes = [vp, d, d, d, d]
    lib.mpco_set_solver_options.argtypes = [vp, d, i]
    lib.mpco_set_drag.argtypes = [_f64p]
    lib.mpco_warm_start.restype = C.POINTER(C.c_double); lib.mpco_warm_start.argtypes = [vp]
    lib.mpco_last_info.restype = C.POINTER(C.c_int); lib.mpco_last_info.argtypes = [vp]
    lib.mpco_last_stats.restype = C.POINTER(C.c_double); lib.mpco_last_stats.argtypes = [vp]
    lib.mpco_Solve.restype = i; lib.mpco_Solve.argtypes = [vp, _f64p, _f64p, _f64p, i]
    if hasattr(lib, "stepo_run"):
        lib.stepo_run.restype = i
        lib.stepo_run.argtypes = [vp, vp, vp, i, d, d, d, i, _f64p, d, _f64p, _f64p, _f64p, _i32p, vp]
        lib.stepo_cur_state_quad.argtypes = [_f64p, _f64p, _f64p, d, d, i, _f64p]
        lib.stepo_get_init_path.argtypes = [_f64p, i, d, d, d, d, d]


class oracle_drag:
    """`with oracle_drag(k):` -- the oracle's dynamics with v' = a - k .* v (oracle/mpc_oracle.c: the use_drag_coefficient switch read
    as matrix products) in the C restatement AND the numpy twin; restored to the default (off) on exit."""

    def __init__(self, k):
        self.k = np.ascontiguousarray(np.broadcast_to(np.asarray(k, np.float64), (3,)).copy())

    @staticmethod
    def _twins():   # the numpy twin is imported as `mpc_oracle_np` (oracle/ on sys.path) by the tests: every loaded instance
        import sys
        return [m for n, m in sys.modules.items() if n.split(".")[-1] == "mpc_oracle_np" and hasattr(m, "DRAG")]

    def __enter__(self):
        load_oracle().mpco_set_drag(self.k)
        for m in self._twins(): m.DRAG[:] = self.k
        return self

    def __exit__(self, *exc):
        load_oracle().mpco_set_drag(np.zeros(3))
        for m in self._twins(): m.DRAG[:] = 0.0
        return False


MPC_DEFAULT_MAX_ITER = 100   # oracle/mpc_oracle.c MPCO_DEFAULT_MAX_ITER == the product's default (amk_mpc_create)


def mpco_solve(P, w0, lbu, ubu, N, K, dt, tol=1e-4, max_iter=MPC_DEFAULT_MAX_ITER, trace=None, **kw):
    """-> (w, info[4], stats[4]) from the C solver.  trace: optional float64 [max_iter, 8] array filled per iteration
    with (phi, E_0, mu, E_mu, alpha, alpha_dual, delta, dphi); kw: other mpco_opts fields."""
    lib = load_oracle()
    opt = MpcoOpts(); lib.mpco_default_opts(C.byref(opt)); opt.tol = tol; opt.max_iter = max_iter
    for k, v in kw.items():
        setattr(opt, k, v)
    if trace is not None:
        assert trace.dtype == np.float64 and trace.shape == (max_iter, 8) and trace.flags.c_contiguous
        opt.trace = trace.ctypes.data_as(C.POINTER(C.c_double))
    w = np.zeros(10 + 14 * N); info = np.zeros(4, np.int32); stats = np.zeros(4)
    lib.mpco_solve(np.ascontiguousarray(P, np.float64), np.ascontiguousarray(w0, np.float64),
                   np.ascontiguousarray(lbu, np.float64), np.ascontiguousarray(ubu, np.float64), N, K, dt,
                   C.byref(opt), w, info, stats)
    return w, info, stats


class MpcOracle:
    """ObstacleAvoidanceMPC restated (oracle/mpc_oracle.c mpco_create ... mpco_Solve)."""

    def __init__(self, T, dt, K):
        self.lib = load_oracle()
        self.h = self.lib.mpco_create(T, dt, K)
        self.N = self.lib.mpco_horizon(self.h); self.K = K
        self.nx = 10 + 14 * self.N

    def configure(self, prm):
        f = lambda v: np.ascontiguousarray(v, np.float64)
        self.lib.mpco_setup_weights(self.h, f(prm.weights)); self.lib.mpco_setup_tau(self.h, f(prm.tau))
        self.lib.mpco_setup_gains(self.h, f(prm.gain)); self.lib.mpco_set_drone_radius(self.h, prm.radius)
        self.lib.mpco_set_drone_accel_limits(self.h, prm.a_min_z, prm.a_max_z, prm.a_max_xy, prm.a_max_yaw_dot)

    def set_solver_options(self, tol=1e-4, max_iter=MPC_DEFAULT_MAX_ITER):
        self.lib.mpco_set_solver_options(self.h, tol, max_iter)

    def Solve(self, ref_states, faster=False):
        u = np.zeros(4); x0 = np.zeros((self.N, 14))
        st = self.lib.mpco_Solve(self.h, np.ascontiguousarray(ref_states, np.float64), u, x0.reshape(-1), int(faster))
        info = np.ctypeslib.as_array(self.lib.mpco_last_info(self.h), (4,)).copy()
        return u, x0, info

    @property
    def warm_start(self):
        return np.ctypeslib.as_array(self.lib.mpco_warm_start(self.h), (self.nx,))

    def close(self):
        if self.h:
            self.lib.mpco_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cur_state_quad(pos, vel, acc, yaw, dt, use_odom_est=True):
    """GetCurStateQuad restated (oracle/step_oracle.c)."""
    sq = np.zeros(10)
    f = lambda v: np.ascontiguousarray(v, np.float64)
    load_oracle().stepo_cur_state_quad(f(pos), f(vel), f(acc), float(yaw), float(dt), int(use_odom_est), sq)
    return sq


def scene_state_quads(scene, prm):
    """[max_iter][10] per-iteration initial states: the clock model of this project -- outer
    iteration i starts (i+1)*decay after the odometry stamp (AvoidanceStateMachine.cpp:327-330,343
    with every iteration taking `decay` seconds)."""
    return np.stack([cur_state_quad(scene["pos"], scene["vel"], scene["acc"], scene["yaw"], prm.decay * (i + 1))
                     for i in range(prm.max_iter)])


def step_oracle(kd_obs, kd_edge, mpc, prm, state_quad, pos_x, ref_path, want_log=False):
    """One control step on the CPU oracle.  kd_*: KdHandle (kdo), mpc: MpcOracle.  ref_path is
    updated in place.  -> dict(u, x0array, flags, ref_log)"""
    lib = load_oracle()
    N, K = mpc.N, mpc.K
    nref = 20 + 10 * N + 3 * K * N
    u = np.zeros(4); x0 = np.zeros((N, 14)); flags = np.zeros(4, np.int32)
    log = np.zeros((prm.max_iter, nref)) if want_log else None
    sq = np.ascontiguousarray(state_quad, np.float64)
    assert sq.shape == (prm.max_iter, 10) and ref_path.shape == (N, 10) and ref_path.flags.c_contiguous
    lib.stepo_run(kd_obs.h, kd_edge.h if kd_edge is not None else None, mpc.h, K, prm.speed, prm.T,
                  prm.safety_distance, prm.max_iter, sq.reshape(-1), float(pos_x), ref_path.reshape(-1), u,
                  x0.reshape(-1), flags, log.ctypes.data_as(C.c_void_p) if want_log else None)
    return dict(u=u, x0array=x0, flags=flags, ref_log=log)


def step_oracle_frames(kd_obs_frames, kd_edge_frames, mpc, prm, state_quad, pos_x, ref_path, Twc=None, cam=None):
    """The control step on a multi-frame map (oracle/step_oracle.c stepo_run_frames).  kd_*_frames: lists of KdHandle, index 0
    = current frame; Twc 4x4 or None; cam = (fx, fy, cx, cy, depth_max, width, height).  ref_path is updated in place."""
    lib = load_oracle()
    lib.stepo_run_frames.restype = C.c_int
    lib.stepo_run_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                     C.c_double, C.c_double, C.c_int, _f64p, C.c_double, _f64p, _f64p, _f64p, _i32p]
    F = len(kd_obs_frames)
    oa = (C.c_void_p * F)(*[k.h for k in kd_obs_frames]); ea = (C.c_void_p * F)(*[k.h for k in kd_edge_frames])
    N, K = mpc.N, mpc.K
    u = np.zeros(4); x0 = np.zeros((N, 14)); flags = np.zeros(4, np.int32)
    T = np.ascontiguousarray(Twc, np.float64).reshape(-1) if Twc is not None else None
    cm = np.ascontiguousarray(cam, np.float64) if cam is not None else None
    sq = np.ascontiguousarray(state_quad, np.float64)
    lib.stepo_run_frames(oa, ea, F, T.ctypes.data_as(C.c_void_p) if T is not None else None,
                         cm.ctypes.data_as(C.c_void_p) if cm is not None else None, mpc.h, K, prm.speed, prm.T,
                         prm.safety_distance, prm.max_iter, sq.reshape(-1), float(pos_x), ref_path.reshape(-1), u,
                         x0.reshape(-1), flags)
    return dict(u=u, x0array=x0, flags=flags)


# ---- depth image -> cloud (oracle/depth_oracle.c) ------------------------------------------------
class DepthoParams(C.Structure):
    _fields_ = [("pixel2meter", C.c_double), ("depth_min", C.c_double), ("depth_max", C.c_double),
                ("resize_scale", C.c_double), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("Tbc", C.c_double * 16)]


def depth_oracle(depth, prm, Twb, stride=3):
    """One scene.  depth [rows, cols] uint16 / float32; prm: dict of the amk_depth_params fields (Tbc 4x4);
    -> (cloud float32 [count, stride], inverse-depth image float32 [H, W])."""
    lib = load_oracle()
    lib.deptho_process.restype = C.c_int
    lib.deptho_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DepthoParams), _f64p,
                                   C.c_void_p, C.c_int, C.c_void_p]
    depth = np.ascontiguousarray(depth)
    assert depth.dtype in (np.uint16, np.float32) and depth.ndim == 2
    rows, cols = depth.shape
    p = DepthoParams(prm["pixel2meter"], prm["depth_min"], prm["depth_max"], prm["resize_scale"], prm["fx"],
                     prm["fy"], prm["cx"], prm["cy"])
    T = np.asarray(prm.get("Tbc", np.eye(4)), np.float64).reshape(4, 4)
    for i in range(16):
        p.Tbc[i] = float(T.flat[i])
    W, H = int(cols / prm["resize_scale"]), int(rows / prm["resize_scale"])
    cloud = np.zeros((W * H, stride), np.float32)
    inv = np.zeros((H, W), np.float32)
    n = lib.deptho_process(depth.ctypes.data_as(C.c_void_p), 0 if depth.dtype == np.uint16 else 1, rows, cols,
                           C.byref(p), np.ascontiguousarray(Twb, np.float64).reshape(-1),
                           cloud.ctypes.data_as(C.c_void_p), stride, inv.ctypes.data_as(C.c_void_p))
    return cloud[:n], inv


def depth_edge_oracle(depth, prm, Twc, stride=3):
    """BuildEdgeCloud for one scene -> (cloud float32 [count, stride], quant, eroded, edges uint8 [H, W])."""
    lib = load_oracle()
    lib.deptho_edge.restype = C.c_int
    lib.deptho_edge.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DepthoParams), _f64p, C.c_void_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    depth = np.ascontiguousarray(depth)
    assert depth.dtype in (np.uint16, np.float32) and depth.ndim == 2
    rows, cols = depth.shape
    p = DepthoParams(prm["pixel2meter"], prm["depth_min"], prm["depth_max"], prm["resize_scale"], prm["fx"],
                     prm["fy"], prm["cx"], prm["cy"])
    T = np.asarray(prm.get("Tbc", np.eye(4)), np.float64).reshape(4, 4)
    for i in range(16):
        p.Tbc[i] = float(T.flat[i])
    W, H = int(cols / prm["resize_scale"]), int(rows / prm["resize_scale"])
    cloud = np.zeros((W * H, stride), np.float32)
    quant = np.zeros((H, W), np.uint8); eroded = np.zeros((H, W), np.uint8); edges = np.zeros((H, W), np.uint8)
    winv = np.zeros(H * W, np.float32); wmag = np.zeros(H * W, np.int16)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib.deptho_edge(vp(depth), 0 if depth.dtype == np.uint16 else 1, rows, cols, C.byref(p),
                        np.ascontiguousarray(Twc, np.float64).reshape(-1), vp(cloud), stride, vp(quant), vp(eroded),
                        vp(edges), vp(winv), vp(wmag))
    return cloud[:n], quant, eroded, edges
