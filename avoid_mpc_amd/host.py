"""Thin Python objects over the C ABI (include/avoid_mpc_amd.h) used by tests and bench.py.

torch is plumbing here (device buffers, streams); every computation happens in the HIP library.
The reference-shaped C++ adapters (KDTreeTwo<double>, FrameKDMap, ObstacleAvoidanceMPC) live in
include/avoid_mpc_amd/*.hpp; these classes are their batched Python twins:

  KdBatch   <- S x KDTreeTwo<double>          AM/include/kd_tree_two.h:53-144
  MpcBatch  <- S x ObstacleAvoidanceMPC       AM/include/HighLvlMpc.h:4-33
  step_batch<- TASK branch of Step            AM/src/AvoidanceStateMachine.cpp:322-355
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def _dev():
    if not torch.cuda.is_available():
        raise capi.AmkError("no GPU visible: the hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class KdBatch:
    def __init__(self, n_scenes, max_points, handle=None):
        """handle: a borrowed amk_kd* (a pipeline slot's); it is then not destroyed by this object."""
        self.lib = capi.load()
        self.S, self.max_points = int(n_scenes), int(max_points)
        self.owned = handle is None
        if handle is None:
            handle = C.c_void_p()
            capi.check(self.lib.amk_kd_create(self.S, self.max_points, C.byref(handle)), "amk_kd_create")
        self.h = handle if isinstance(handle, C.c_void_p) else C.c_void_p(handle)

    def close(self):
        if getattr(self, "h", None):
            if self.owned:
                self.lib.amk_kd_destroy(self.h)
            self.h = None

    __del__ = close

    def set_tie_order(self, mode):
        """capi.AMK_TIES_LOWEST_INDEX (default) or capi.AMK_TIES_NANOFLANN; takes effect at the next build."""
        capi.check(self.lib.amk_kd_set_tie_order(self.h, int(mode)), "amk_kd_set_tie_order")

    def exact_status(self, stream=None):
        """amk_kd_exact_status: int32 device tensor [S] -- -1 mode off / 0 the reference-shaped tree answers / 1 its build gave up /
        2 deeper than the traversal stack (1, 2: the bucketed index answers)."""
        st = torch.empty(self.S, dtype=torch.int32, device="cuda")
        capi.check(self.lib.amk_kd_exact_status(self.h, capi.dptr(st), capi.stream_ptr(stream)), "amk_kd_exact_status")
        return st

    def build(self, xyz, counts=None, stream=None):
        """InitializeNew: xyz float32 device tensor [S, max_points, 3|4]; counts int32 [S] or None."""
        assert xyz.dtype == torch.float32 and xyz.dim() == 3 and xyz.shape[0] == self.S
        assert xyz.shape[1] >= self.max_points or self.max_points == 0
        if counts is not None:
            assert counts.dtype == torch.int32 and counts.numel() == self.S
        capi.check(self.lib.amk_kd_build(self.h, capi.dptr(xyz), int(xyz.shape[2]),
                                         int(xyz.shape[1] * xyz.shape[2]), capi.dptr(counts),
                                         capi.stream_ptr(stream)), "amk_kd_build")

    def sizes(self, stream=None):
        out = np.zeros(self.S, np.int32)
        capi.check(self.lib.amk_kd_sizes(self.h, out.ctypes.data_as(C.c_void_p), capi.stream_ptr(stream)),
                   "amk_kd_sizes")
        return out

    def search(self, queries, k, want_pts=True, stream=None, out=None):
        """SearchForNearest for queries float64 [S, Q, 3] -> dict(indices, sqdist, pts, counts)."""
        assert queries.dtype == torch.float64 and queries.shape[0] == self.S and queries.shape[2] == 3
        Q = int(queries.shape[1])
        dev = queries.device
        if out is None:
            out = dict(indices=torch.empty((self.S, Q, k), dtype=torch.int32, device=dev),
                       sqdist=torch.empty((self.S, Q, k), dtype=torch.float64, device=dev),
                       pts=torch.empty((self.S, Q, k, 3), dtype=torch.float32, device=dev) if want_pts else None,
                       counts=torch.empty((self.S, Q), dtype=torch.int32, device=dev))
        capi.check(self.lib.amk_kd_search(self.h, capi.dptr(queries), Q, int(k), capi.dptr(out["indices"]),
                                          capi.dptr(out["sqdist"]), capi.dptr(out["pts"]),
                                          capi.dptr(out["counts"]), capi.stream_ptr(stream)), "amk_kd_search")
        return out

    def keyframe_sweep(self, current, th_dist, th_count, stream=None):
        """KeyframeThreadWorker's sweep: self = last keyframe, rebuilt from its outliers w.r.t. `current`.
        -> (outliers int32 [S], rebuilt int32 [S]) device tensors"""
        dev = _dev()
        outl = torch.empty(self.S, dtype=torch.int32, device=dev)
        reb = torch.empty(self.S, dtype=torch.int32, device=dev)
        capi.check(self.lib.amk_kd_keyframe_sweep(self.h, current.h, float(th_dist), int(th_count), capi.dptr(outl),
                                                  capi.dptr(reb), capi.stream_ptr(stream)), "amk_kd_keyframe_sweep")
        return outl, reb

    # host-buffer conveniences ---------------------------------------------------------------
    def build_host(self, xyz, counts=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        assert xyz.ndim == 3 and xyz.shape[0] == self.S
        cp = None
        if counts is not None:
            counts = np.ascontiguousarray(counts, np.int32)
            cp = counts.ctypes.data_as(C.c_void_p)
        capi.check(self.lib.amk_kd_build_host(self.h, xyz.ctypes.data_as(C.c_void_p), int(xyz.shape[2]),
                                              int(xyz.shape[1] * xyz.shape[2]), cp), "amk_kd_build_host")

    def search_host(self, queries, k):
        q = np.ascontiguousarray(queries, np.float64)
        assert q.ndim == 3 and q.shape[0] == self.S and q.shape[2] == 3
        Q = q.shape[1]
        idx = np.zeros((self.S, Q, k), np.int32); d2 = np.zeros((self.S, Q, k), np.float64)
        pts = np.zeros((self.S, Q, k, 3), np.float32); cnt = np.zeros((self.S, Q), np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        capi.check(self.lib.amk_kd_search_host(self.h, vp(q), Q, int(k), vp(idx), vp(d2), vp(pts), vp(cnt)),
                   "amk_kd_search_host")
        return dict(indices=idx, sqdist=d2, pts=pts, counts=cnt)


def kd_build_pair(kd_obstacle, xyz, kd_edge, edge_xyz, counts=None, edge_counts=None, stream=None):
    """amk_kd_build_pair: FrameKDMap::AddVertex's two InitializeNew calls as one launch (clouds packed [S, max_points, 3|4])."""
    assert xyz.dtype == torch.float32 and edge_xyz.dtype == torch.float32 and xyz.shape[2] == edge_xyz.shape[2]
    assert xyz.shape[1] == kd_obstacle.max_points and edge_xyz.shape[1] == kd_edge.max_points
    capi.check(capi.load().amk_kd_build_pair(kd_obstacle.h, capi.dptr(xyz), capi.dptr(counts), kd_edge.h, capi.dptr(edge_xyz),
                                             capi.dptr(edge_counts), int(xyz.shape[2]), capi.stream_ptr(stream)), "amk_kd_build_pair")


def kd_tie_flags(kd, queries, k, query_stride=3, stream=None):
    """amk_kd_tie_flags: int32 [S, n_queries], 1 where the k nearest (or the k-th and the best rejected) hold an exact tie."""
    S = kd.S if hasattr(kd, "S") else queries.shape[0]
    nq = int(queries.shape[1])
    out = torch.empty((S, nq), dtype=torch.int32, device=queries.device)
    capi.check(capi.load().amk_kd_tie_flags(kd.h, capi.dptr(queries), int(query_stride), nq, int(k), capi.dptr(out),
                                            capi.stream_ptr(stream)), "amk_kd_tie_flags")
    return out


class MpcBatch:
    def __init__(self, T, dt, nearest_point_num, n_scenes, handle=None):
        """handle: a borrowed amk_mpc* (a pipeline slot's); it is then not destroyed by this object."""
        self.lib = capi.load()
        self.owned = handle is None
        if handle is None:
            handle = C.c_void_p()
            capi.check(self.lib.amk_mpc_create(float(T), float(dt), int(nearest_point_num), int(n_scenes),
                                               C.byref(handle)), "amk_mpc_create")
        h = handle if isinstance(handle, C.c_void_p) else C.c_void_p(handle)
        self.h = h
        self.S, self.K = int(n_scenes), int(nearest_point_num)
        self.N = self.lib.amk_mpc_horizon(h)
        self.nx = self.lib.amk_mpc_nx(h)
        self.ref_len = self.lib.amk_mpc_ref_len(h)

    def close(self):
        if getattr(self, "h", None):
            if self.owned:
                self.lib.amk_mpc_destroy(self.h)
            self.h = None

    __del__ = close

    def _arr(self, v, n):
        a = np.ascontiguousarray(v, np.float64)
        assert a.size == n
        return a.ctypes.data_as(C.c_void_p), a

    def SetupWeights(self, w):
        p, _k = self._arr(w, 25); capi.check(self.lib.amk_mpc_setup_weights(self.h, p), "SetupWeights")

    def SetupTau(self, tau):
        p, _k = self._arr(tau, 4); capi.check(self.lib.amk_mpc_setup_tau(self.h, p), "SetupTau")

    def SetupGains(self, g):
        p, _k = self._arr(g, 4); capi.check(self.lib.amk_mpc_setup_gains(self.h, p), "SetupGains")

    def SetDroneRadius(self, r):
        capi.check(self.lib.amk_mpc_set_drone_radius(self.h, float(r)), "SetDroneRadius")

    def SetDragCoefficient(self, kx, ky=None, kz=None):
        """v' = a - k .* v (the generator's use_drag_coefficient switch read as matrix products: include/avoid_mpc_amd.h); one value = isotropic."""
        ky = kx if ky is None else ky; kz = kx if kz is None else kz
        capi.check(self.lib.amk_mpc_set_drag_coefficient(self.h, float(kx), float(ky), float(kz)), "SetDragCoefficient")

    def SetDroneAccelLimits(self, aMinZ, aMaxZ, aMaxXy, aMaxYawDot):
        capi.check(self.lib.amk_mpc_set_drone_accel_limits(self.h, float(aMinZ), float(aMaxZ), float(aMaxXy),
                                                           float(aMaxYawDot)), "SetDroneAccelLimits")

    def set_solver_options(self, tol=1e-4, max_iter=capi.AMK_MPC_DEFAULT_MAX_ITER):
        capi.check(self.lib.amk_mpc_set_solver_options(self.h, float(tol), int(max_iter)), "set_solver_options")

    def set_solve_budget(self, budget, budget_rounds=0):
        """amk_mpc_set_solve_budget: interior-point iterations per solve LAUNCH inside amk_step_batch (0: off); unfinished solves
        pause and are resumed inside the next round's launch.  A scheduling knob: results are unchanged bit for bit."""
        capi.check(self.lib.amk_mpc_set_solve_budget(self.h, int(budget), int(budget_rounds)), "set_solve_budget")

    def set_precision(self, bits):
        """64 (default) or 32: arithmetic of the solve (BASELINE config C5's fp32 tolerance check)."""
        capi.check(self.lib.amk_mpc_set_precision(self.h, int(bits)), "set_precision")

    def configure(self, prm):
        """SetupMPC (AvoidanceStateMachine.cpp:55-70) from a synth.MpcParams."""
        self.SetupWeights(prm.weights); self.SetupTau(prm.tau); self.SetupGains(prm.gain)
        self.SetDroneAccelLimits(prm.a_min_z, prm.a_max_z, prm.a_max_xy, prm.a_max_yaw_dot)
        self.SetDroneRadius(prm.radius)
        if any(getattr(prm, "drag", (0.0, 0.0, 0.0))):
            self.SetDragCoefficient(*prm.drag)

    def Solve(self, ref_states, faster=False, stream=None, want_traj=True):
        assert ref_states.dtype == torch.float64 and tuple(ref_states.shape) == (self.S, self.ref_len)
        dev = ref_states.device
        u = torch.empty((self.S, 4), dtype=torch.float64, device=dev)
        x0 = torch.empty((self.S, self.N, 14), dtype=torch.float64, device=dev) if want_traj else None
        info = torch.empty((self.S, 4), dtype=torch.int32, device=dev)
        capi.check(self.lib.amk_mpc_solve(self.h, capi.dptr(ref_states), capi.dptr(u), capi.dptr(x0),
                                          capi.dptr(info), int(bool(faster)), capi.stream_ptr(stream)),
                   "amk_mpc_solve")
        return u, x0, info

    def eval(self, w, ref_states, lam_f=None, stream=None, want=("f", "grad_f", "g", "jac_g", "hess_l")):
        """amk_mpc_eval: the plugin's nlp_f / nlp_grad_f / nlp_g / nlp_jac_g / nlp_hess_l at w [S, nx] -> dict of tensors."""
        assert w.dtype == torch.float64 and tuple(w.shape) == (self.S, self.nx)
        assert ref_states.dtype == torch.float64 and tuple(ref_states.shape) == (self.S, self.ref_len)
        dev = w.device
        shp = dict(f=(self.S,), grad_f=(self.S, self.nx), g=(self.S, self.lib.amk_mpc_ng(self.h)),
                   jac_g=(self.S, self.lib.amk_mpc_jac_nnz(self.h)), hess_l=(self.S, self.lib.amk_mpc_hess_nnz(self.h)))
        out = {k: torch.empty(shp[k], dtype=torch.float64, device=dev) for k in want}
        capi.check(self.lib.amk_mpc_eval(self.h, capi.dptr(w), capi.dptr(ref_states), capi.dptr(lam_f),
                                         *[capi.dptr(out.get(k)) for k in ("f", "grad_f", "g", "jac_g", "hess_l")],
                                         capi.stream_ptr(stream)), "amk_mpc_eval")
        return out

    def sparsity(self, which):
        """CCS pattern ('jac_g' or 'hess_l') -> (colind int32 [nx + 1], row int32 [nnz])."""
        nnz = (self.lib.amk_mpc_jac_nnz if which == "jac_g" else self.lib.amk_mpc_hess_nnz)(self.h)
        colind = np.zeros(self.nx + 1, np.int32); row = np.zeros(nnz, np.int32)
        fn = self.lib.amk_mpc_jac_sparsity if which == "jac_g" else self.lib.amk_mpc_hess_sparsity
        capi.check(fn(self.h, colind.ctypes.data_as(C.c_void_p), row.ctypes.data_as(C.c_void_p)), which + " sparsity")
        return colind, row

    def get_warm_start(self, stream=None):
        w = torch.empty((self.S, self.nx), dtype=torch.float64, device=_dev())
        capi.check(self.lib.amk_mpc_get_warm_start(self.h, capi.dptr(w), capi.stream_ptr(stream)), "get_warm_start")
        return w

    def set_warm_start(self, w, stream=None):
        assert w.dtype == torch.float64 and tuple(w.shape) == (self.S, self.nx)
        capi.check(self.lib.amk_mpc_set_warm_start(self.h, capi.dptr(w), capi.stream_ptr(stream)), "set_warm_start")

    def reset_warm_start(self, stream=None):
        capi.check(self.lib.amk_mpc_reset_warm_start(self.h, capi.stream_ptr(stream)), "reset_warm_start")


class KfMap:
    """amk_kfmap: FrameKDMap's keyframe list for S scenes on the device (FrameKDMap.cpp:29-74,233-252,428-488)."""

    def __init__(self, n_scenes, max_points, max_edge_points, max_frame_count, th_dist, th_count, depth_min, Tbc):
        self.lib = capi.load()
        p = capi.KfmapParams(int(max_frame_count), int(th_count), float(th_dist), float(depth_min),
                             (C.c_double * 16)(*[float(v) for v in np.asarray(Tbc, np.float64).reshape(-1)]))
        h = C.c_void_p()
        capi.check(self.lib.amk_kfmap_create(int(n_scenes), int(max_points), int(max_edge_points), C.byref(p), C.byref(h)), "amk_kfmap_create")
        self.h, self.S, self.F = h, int(n_scenes), int(max_frame_count) + 1
        self.max_points, self.max_edge_points = int(max_points), int(max_edge_points)

    def close(self):
        if getattr(self, "h", None):
            self.lib.amk_kfmap_destroy(self.h)
            self.h = None

    __del__ = close

    def add_vertex(self, clouds, edges, Twc, counts=None, edge_counts=None, first_scene=0, stream=None):
        """clouds [n, max_points, 3|4] f32, edges likewise, Twc [n, 4, 4] f64 = Twb * Tbc of the frame; counts int32 [n] or None."""
        assert clouds.dtype == torch.float32 and edges.dtype == torch.float32 and Twc.dtype == torch.float64
        assert clouds.shape[1] == self.max_points and edges.shape[1] == self.max_edge_points and clouds.shape[2] == edges.shape[2]
        capi.check(self.lib.amk_kfmap_add_vertex(self.h, int(first_scene), int(clouds.shape[0]), capi.dptr(clouds), capi.dptr(counts),
                                                 capi.dptr(edges), capi.dptr(edge_counts), int(clouds.shape[2]), capi.dptr(Twc),
                                                 capi.stream_ptr(stream)), "amk_kfmap_add_vertex")

    def update(self, stream=None):
        capi.check(self.lib.amk_kfmap_update(self.h, capi.stream_ptr(stream)), "amk_kfmap_update")

    def step(self, mpc, prm, state_quad, pos_x, ref_path, cam=None, stream=None, out=None):
        """amk_kfmap_step; ref_path is refilled in place.  -> dict(u, x0array, flags)"""
        S, N = self.S, mpc.N
        dev = _dev()
        if out is None:
            out = dict(u=torch.empty((S, 4), dtype=torch.float64, device=dev), x0array=torch.empty((S, N, 14), dtype=torch.float64, device=dev),
                       flags=torch.empty((S, 4), dtype=torch.int32, device=dev))
        sp = capi.StepParams(float(prm.speed), float(prm.safety_distance), int(prm.max_iter), 0)
        capi.check(self.lib.amk_kfmap_step(self.h, C.byref(cam) if cam is not None else None, mpc.h, C.byref(sp), capi.dptr(state_quad),
                                           capi.dptr(pos_x), capi.dptr(ref_path), capi.dptr(out["u"]), capi.dptr(out["x0array"]),
                                           capi.dptr(out["flags"]), capi.stream_ptr(stream)), "amk_kfmap_step")
        return out

    def state(self):
        """-> dict(n_keyframes [S], n_query_frames [S], last_outliers [S], frame_sizes [S, F]) (synchronises)"""
        nk = np.zeros(self.S, np.int32); nq = np.zeros(self.S, np.int32); out = np.zeros(self.S, np.int32)
        sz = np.zeros((self.S, self.F), np.int32)
        capi.check(self.lib.amk_kfmap_state_host(self.h, nk.ctypes.data_as(C.c_void_p), nq.ctypes.data_as(C.c_void_p),
                                                 out.ctypes.data_as(C.c_void_p), sz.ctypes.data_as(C.c_void_p)), "amk_kfmap_state_host")
        return dict(n_keyframes=nk, n_query_frames=nq, last_outliers=out, frame_sizes=sz)


class Pipeline:
    """amk_pipeline: n_slots launches in flight, each slot = {HIP stream, obstacle + edge index, MPC batch, outputs};
    gang = frames (of n_scenes scenes) that share one set of launches -- the slot's handles then hold gang * n_scenes scenes."""

    def __init__(self, n_slots, n_scenes, max_points, max_edge_points, prm, queue_depth=0, gang=0, farest_point=500.0,
                 slow_down_kp=0.3, slow_down_kd=0.3, iter_time=0.0, use_odom_est=True, depth=None, task="forward", keyframes=None):
        """keyframes: dict(max_frame_count, th_dist, th_count, depth_min[, Tbc]) -> every slot keeps a keyframe map (amk_kfmap)."""
        self.lib = capi.load()
        task = capi.TaskParams(float(prm.decay), float(iter_time), float(farest_point), float(prm.height), float(slow_down_kp),
                               float(slow_down_kd), float(prm.a_max_xy), float(prm.a_max_z), int(bool(use_odom_est)),
                               {"forward": 0, "global_goal": 1}[task])
        cfg = capi.PipelineConfig(int(n_slots), int(n_scenes), int(max_points), int(max_edge_points), float(prm.T), float(prm.dt),
                                  int(prm.K), int(queue_depth), int(gang),
                                  capi.StepParams(float(prm.speed), float(prm.safety_distance), int(prm.max_iter), 0), task,
                                  depth if depth is not None else capi.DepthParams(),
                                  self._kfmap_params(keyframes))
        h = C.c_void_p()
        capi.check(self.lib.amk_pipeline_create(C.byref(cfg), C.byref(h)), "amk_pipeline_create")
        self.h, self.n_slots, self.S, self.prm = h, int(n_slots), int(n_scenes), prm
        self.N = self.lib.amk_mpc_horizon(self.lib.amk_pipeline_mpc(h, 0))
        self.gang = self.lib.amk_pipeline_gang(h)
        self._events = []
        hs = n_scenes * self.gang   # scenes of a slot's handles
        self._mpc = [MpcBatch(prm.T, prm.dt, prm.K, hs, handle=self.lib.amk_pipeline_mpc(h, i)) for i in range(n_slots)]
        self._kd = [(KdBatch(hs, max_points, handle=self.lib.amk_pipeline_kd(h, i, 0)),
                     KdBatch(hs, max_edge_points, handle=self.lib.amk_pipeline_kd(h, i, 1))) for i in range(n_slots)]
        for m in self._mpc:
            m.configure(prm)

    @staticmethod
    def _kfmap_params(keyframes):
        """amk_kfmap_params of the configuration; keyframes["Tbc"] (4 x 4, optional) is mParamTbc -- absent: the depth configuration's."""
        if not keyframes:
            return capi.KfmapParams()
        kp = capi.KfmapParams(int(keyframes["max_frame_count"]), int(keyframes["th_count"]), float(keyframes["th_dist"]),
                              float(keyframes["depth_min"]))
        if keyframes.get("Tbc") is not None:
            T = np.asarray(keyframes["Tbc"], np.float64).reshape(16)
            for i in range(16):
                kp.Tbc[i] = float(T[i])
        return kp

    def kfmap_state(self, slot):
        """State of the slot's keyframe map (amk_kfmap_state_host): dict(n_keyframes, n_query_frames, last_outliers, frame_sizes)."""
        h = self.lib.amk_pipeline_kfmap(self.h, int(slot))
        assert h, "the pipeline was created without keyframes"
        S = self.S * self.gang
        F = self.lib.amk_kfmap_frames(h)
        nk = np.zeros(S, np.int32); nq = np.zeros(S, np.int32); out = np.zeros(S, np.int32); sz = np.zeros((S, F), np.int32)
        capi.check(self.lib.amk_kfmap_state_host(h, nk.ctypes.data_as(C.c_void_p), nq.ctypes.data_as(C.c_void_p),
                                                 out.ctypes.data_as(C.c_void_p), sz.ctypes.data_as(C.c_void_p)), "amk_kfmap_state_host")
        return dict(n_keyframes=nk, n_query_frames=nq, last_outliers=out, frame_sizes=sz)

    def mpc(self, slot):
        return self._mpc[slot]

    def kd(self, slot, which):
        return self._kd[slot][which]

    def close(self):
        if getattr(self, "h", None):
            self.lib.amk_pipeline_destroy(self.h)
            self.h = None

    __del__ = close

    def submit(self, clouds, edges, state_quad=None, pos_x=None, ref_path_init=None, cloud_counts=None, edge_counts=None, u_out=None,
               keep_warm_start=False, order_after_current_stream=True, odom=None, odom_age=0.0, cmd_out=None, depth=None, Twb=None,
               keyframes=None, Twc_cur=None, cam=None, global_goal=None):
        """One fresh frame + control step on the next slot; returns its ticket at once (blocks only when that slot's queue
        is full).  ticket % n_slots = slot; with a gang the frame is staged until the gang is full (or wait / drain).
        All tensors are device tensors that must stay alive until the frame finished.  A slot runs on its own stream: by default
        an event recorded on torch's current stream is handed over (amk_pipeline_frame.input_ready) so that the slot reads the
        inputs only after the work queued on that stream so far -- like every other entry point of this module, which run ON the
        current stream.  order_after_current_stream=False: the caller guarantees the inputs are complete (bench.py: frames made
        once, synchronised).
        TASK mode: odom float64 [S, 10] = [mPos, yaw, mVel, mAcc] instead of state_quad / pos_x; the slot then runs GetInitPath on
        its own mRefPath (ref_path_init, when given, re-initialises it first), the clock model, the step and PubCmd /
        PubSlowDownCmd (cmd_out float64 [S, 3])."""
        opt = lambda t: t.data_ptr() if t is not None else None
        dinfo = (None, 0, 0, 0, 0, None)
        if depth is not None:   # a frame that starts at the raw depth image: depth [S, rows, cols] uint16 / int16 / float32, Twb [S, 4, 4]
            assert depth.dim() == 3 and depth.is_contiguous() and Twb.dtype == torch.float64 and Twb.is_contiguous()
            kind = capi.AMK_DEPTH_F32 if depth.dtype == torch.float32 else capi.AMK_DEPTH_U16
            dinfo = (depth.data_ptr(), kind, int(depth.shape[1]), int(depth.shape[2]), 0, Twb.data_ptr())
        else:
            assert clouds.dtype == torch.float32 and edges.dtype == torch.float32 and clouds.shape[2] == edges.shape[2]
        self._kf_keep = (None, None, cam)
        kinfo = (None, None, 0, 0, opt(Twc_cur), C.cast(C.pointer(cam), C.c_void_p) if cam is not None else None)
        if keyframes:   # [(KdBatch obstacle, KdBatch edge), ...]: the multi-frame map [this frame, keyframes ...] (gang 1 only)
            ko = (C.c_void_p * len(keyframes))(*[k[0].h for k in keyframes]); ke = (C.c_void_p * len(keyframes))(*[k[1].h for k in keyframes])
            self._kf_keep = (ko, ke, cam)
            kinfo = (C.cast(ko, C.c_void_p), C.cast(ke, C.c_void_p), len(keyframes), 0) + kinfo[4:]
        ev_ptr = None
        if order_after_current_stream:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            ev_ptr = ev.cuda_event
            self._events.append(ev)                      # alive until the frame has been launched: keep the newest ones
            if len(self._events) > 4 * self.n_slots * self.gang + 8:
                del self._events[:len(self._events) // 2]
        fr = capi.PipelineFrame(opt(clouds), opt(cloud_counts), opt(edges), opt(edge_counts),
                                int(clouds.shape[2]) if clouds is not None else 3, int(bool(keep_warm_start)), opt(state_quad), opt(pos_x),
                                opt(ref_path_init), opt(u_out), opt(odom), float(odom_age), opt(cmd_out), *dinfo, *kinfo, ev_ptr,
                                opt(global_goal))
        slot = C.c_int(-1)
        capi.check(self.lib.amk_pipeline_submit(self.h, C.byref(fr), C.byref(slot)), "amk_pipeline_submit")
        return slot.value

    def wait_stream(self, ticket, stream=None):
        """Device-side wait (amk_pipeline_wait_stream): work queued on `stream` (default: torch's current stream) after this call
        runs after the frame's step; the host does not block."""
        capi.check(self.lib.amk_pipeline_wait_stream(self.h, int(ticket), capi.stream_ptr(stream)), "amk_pipeline_wait_stream")

    def output_tensors(self, ticket):
        """The frame's results as torch tensors ALIASING the slot's buffers (no copy, no synchronisation): valid to read on a
        stream that waited for the frame (wait / wait_stream) until the slot's next launch overwrites them.
        -> dict(u [S,4], x0array [S,N,14], flags [S,4] int32, ref_path [S,N,10])"""
        ptr = [C.c_void_p() for _ in range(4)]
        capi.check(self.lib.amk_pipeline_outputs(self.h, int(ticket), *[C.byref(p) for p in ptr]), "amk_pipeline_outputs")
        S, N = self.S, self.N
        return dict(u=capi.tensor_from_ptr(ptr[0].value, (S, 4), torch.float64), x0array=capi.tensor_from_ptr(ptr[1].value, (S, N, 14), torch.float64),
                    flags=capi.tensor_from_ptr(ptr[2].value, (S, 4), torch.int32), ref_path=capi.tensor_from_ptr(ptr[3].value, (S, N, 10), torch.float64))

    def wait(self, ticket):
        capi.check(self.lib.amk_pipeline_wait(self.h, int(ticket)), "amk_pipeline_wait")

    def drain(self):
        capi.check(self.lib.amk_pipeline_drain(self.h), "amk_pipeline_drain")

    def outputs(self, ticket):
        """Host copies of the frame's results (after wait): dict(u [S,4], x0array [S,N,14], flags [S,4], ref_path [S,N,10])."""
        ptr = [C.c_void_p() for _ in range(4)]
        capi.check(self.lib.amk_pipeline_outputs(self.h, int(ticket), *[C.byref(p) for p in ptr]), "amk_pipeline_outputs")
        S, N = self.S, self.N
        shapes = [((S, 4), np.float64), ((S, N, 14), np.float64), ((S, 4), np.int32), ((S, N, 10), np.float64)]
        out = {}
        for name, p, (shp, dt) in zip(("u", "x0array", "flags", "ref_path"), ptr, shapes):
            a = np.empty(shp, dt)
            capi.check_hip(capi.hip_memcpy_dtoh(a, p.value), name)
            out[name] = a
        return out


class Shard:
    """amk_shard: one process per GPU, RCCL bound by the library (ncclCommInitRank / ncclAllGather / ncclAllReduce)."""

    def __init__(self, rank, world, unique_id):
        self.lib = capi.load()
        h = C.c_void_p()
        capi.check(self.lib.amk_shard_create(unique_id, int(rank), int(world), C.byref(h)), "amk_shard_create")
        self.h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        capi.check(capi.load().amk_shard_unique_id(buf), "amk_shard_unique_id")
        return buf.raw

    def gather(self, local, out, stream=None):
        """out[r * n : (r + 1) * n] = rank r's `local` (float64 device tensors, n = local.numel())."""
        assert local.dtype == torch.float64 and out.dtype == torch.float64 and out.numel() == local.numel() * self.world
        capi.check(self.lib.amk_shard_gather(self.h, capi.dptr(local), int(local.numel()), capi.dptr(out),
                                             capi.stream_ptr(stream)), "amk_shard_gather")
        return out

    def max(self, values, stream=None):
        capi.check(self.lib.amk_shard_max(self.h, capi.dptr(values), int(values.numel()), capi.stream_ptr(stream)), "amk_shard_max")
        return values

    def wait(self, stream=None, timeout_s=120.0):
        """amk_shard_wait: the watchdog of the exchange step -- returns capi.AMK_OK or capi.AMK_ERR_TIMEOUT (a hung collective)."""
        return self.lib.amk_shard_wait(self.h, capi.stream_ptr(stream), float(timeout_s))

    @staticmethod
    def rccl_info():
        """-> dict(path, version, was_loaded) of the RCCL the library bound, or None when no librccl can be loaded."""
        buf = C.create_string_buffer(1024); ver = C.c_int(0); was = C.c_int(0)
        if capi.load().amk_shard_rccl_info(buf, 1024, C.byref(ver), C.byref(was)) != capi.AMK_OK:
            return None
        return dict(path=buf.value.decode(), version=ver.value, loaded_before_amk=bool(was.value))

    def close(self):
        if getattr(self, "h", None):
            self.lib.amk_shard_destroy(self.h)
            self.h = None

    __del__ = close


def step_batch(kd_obstacle, kd_edge, mpc, prm, state_quad, pos_x, ref_path, stream=None, out=None):
    """One control step for every scene (amk_step_batch).  ref_path is updated in place."""
    S, N = mpc.S, mpc.N
    assert state_quad.dtype == torch.float64 and tuple(state_quad.shape) == (S, prm.max_iter, 10)
    assert ref_path.dtype == torch.float64 and tuple(ref_path.shape) == (S, N, 10)
    assert pos_x.dtype == torch.float64 and pos_x.numel() == S
    dev = ref_path.device
    if out is None:
        out = dict(u=torch.empty((S, 4), dtype=torch.float64, device=dev),
                   x0array=torch.empty((S, N, 14), dtype=torch.float64, device=dev),
                   flags=torch.empty((S, 4), dtype=torch.int32, device=dev))
    sp = capi.StepParams(float(prm.speed), float(prm.safety_distance), int(prm.max_iter), 0)
    capi.check(capi.load().amk_step_batch(kd_obstacle.h, kd_edge.h, mpc.h, C.byref(sp), capi.dptr(state_quad),
                                          capi.dptr(pos_x), capi.dptr(ref_path), capi.dptr(out["u"]),
                                          capi.dptr(out["x0array"]), capi.dptr(out["flags"]),
                                          capi.stream_ptr(stream)), "amk_step_batch")
    return out


def step_batch_frames(kd_obstacle_frames, kd_edge_frames, mpc, prm, state_quad, pos_x, ref_path, Twc=None, cam=None,
                      stream=None, out=None):
    """amk_step_batch_frames: the control step over a multi-frame map.  kd_*_frames: lists of KdBatch, index 0 = current
    frame; Twc float64 [S, 4, 4] (or None); cam: capi.FrameCamera."""
    S, N = mpc.S, mpc.N
    F = len(kd_obstacle_frames)
    assert F == len(kd_edge_frames) and F >= 1
    dev = ref_path.device
    if out is None:
        out = dict(u=torch.empty((S, 4), dtype=torch.float64, device=dev),
                   x0array=torch.empty((S, N, 14), dtype=torch.float64, device=dev),
                   flags=torch.empty((S, 4), dtype=torch.int32, device=dev))
    sp = capi.StepParams(float(prm.speed), float(prm.safety_distance), int(prm.max_iter), 0)
    oa = (C.c_void_p * F)(*[k.h for k in kd_obstacle_frames]); ea = (C.c_void_p * F)(*[k.h for k in kd_edge_frames])
    capi.check(capi.load().amk_step_batch_frames(oa, ea, F, capi.dptr(Twc), C.byref(cam) if cam is not None else None, mpc.h,
                                                 C.byref(sp), capi.dptr(state_quad), capi.dptr(pos_x), capi.dptr(ref_path),
                                                 capi.dptr(out["u"]), capi.dptr(out["x0array"]), capi.dptr(out["flags"]),
                                                 capi.stream_ptr(stream)), "amk_step_batch_frames")
    return out


def depth_params(pixel2meter=1.0, depth_min=0.1, depth_max=100.0, resize_scale=10.0, fx=320.0, fy=320.0, cx=320.0,
                 cy=240.0, Tbc=None):
    """amk_depth_params with the defaults of AM/config/mpc_parameters.yaml:59-66."""
    p = capi.DepthParams(float(pixel2meter), float(depth_min), float(depth_max), float(resize_scale), float(fx),
                         float(fy), float(cx), float(cy))
    T = np.eye(4) if Tbc is None else np.asarray(Tbc, np.float64).reshape(4, 4)
    for i in range(16):
        p.Tbc[i] = float(T.flat[i])
    return p


def depth_to_cloud(depth, params, Twb, point_stride=3, stream=None, edge=False):
    """FrameKDMap::ProcessDepth (edge=True: FrameKDMap::BuildEdgeCloud, Twb then = mCurFrame.Twc) for a batch: depth [S, rows, cols] uint16 / float32 device tensor, Twb [S, 4, 4]
    float64 -> (cloud float32 [S, W*H, point_stride], counts int32 [S]); the cloud feeds KdBatch.build(cloud, counts)."""
    assert depth.dim() == 3 and depth.dtype in (torch.uint16, torch.int16, torch.float32) and depth.is_contiguous()
    S, rows, cols = (int(v) for v in depth.shape)
    assert Twb.dtype == torch.float64 and tuple(Twb.shape) == (S, 4, 4) and Twb.is_contiguous()
    lib = capi.load()
    w, h = C.c_int(), C.c_int()
    capi.check(lib.amk_depth_out_size(rows, cols, params.resize_scale, C.byref(w), C.byref(h)), "amk_depth_out_size")
    cap = w.value * h.value
    cloud = torch.zeros((S, cap, point_stride), dtype=torch.float32, device=depth.device)
    counts = torch.empty(S, dtype=torch.int32, device=depth.device)
    kind = capi.AMK_DEPTH_F32 if depth.dtype == torch.float32 else capi.AMK_DEPTH_U16
    fn = lib.amk_depth_to_edge_cloud if edge else lib.amk_depth_to_cloud
    capi.check(fn(capi.dptr(depth), kind, rows, cols, rows * cols, S, C.byref(params), capi.dptr(Twb), capi.dptr(cloud),
                  int(point_stride), cap * point_stride, capi.dptr(counts), capi.stream_ptr(stream)),
               "amk_depth_to_edge_cloud" if edge else "amk_depth_to_cloud")
    return cloud, counts
