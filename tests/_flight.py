"""Closed-loop flight drivers (TEST INFRASTRUCTURE): the same flights on the CPU oracle, on the GPU through the C ABI's
pipeline, and on the IPOPT-shaped emulation capped at the reference's 10 iterations.

Everything around the step (world, frames, GetInitPath, clock model, command, vehicle) is avoid_mpc_amd/flight.py and is
shared by the three drivers; what differs is who runs the step:

  oracle_flights   oracle/kd_oracle.c + step_oracle.c + mpc_oracle.c (warm start kept inside the oracle's MPC object)
  gpu_flights      amk_pipeline_submit(keep_warm_start = 1) on one slot per batch of flights          (needs a GPU)
  ipopt_flights    the oracle's KD queries + oracle/ipopt_emul.py with max_iter = 10 as the Solve of HighLvlMpc.cpp:93-137

Reference: AM/src/AvoidanceStateMachine.cpp:24-54,183-203,322-355,369-397; AM/src/HighLvlMpc.cpp:17-23,109-129.
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from avoid_mpc_amd import flight, synth  # noqa: E402


def make_prm(cfg="C2", **kw):
    c = synth.CONFIGS[cfg]
    return synth.MpcParams(T=c["T"], K=c["K"], **kw), c["n"]


def _log_arrays(periods):
    return dict(x=np.zeros((periods + 1, 10)), u=np.zeros((periods, 4)), flags=np.zeros((periods, 4), np.int32),
                cmd=np.zeros((periods, 3)))


class _GivenFrames:
    """Frames handed over by the caller (bench.py --workload flight: the frames its GPU flights saw)."""

    def __init__(self, clouds, edges, cyl=None):
        self.clouds, self.edges, self.cyl = clouds, edges, cyl

    def frame(self, t):
        return self.clouds[t], self.edges[t]

    def clearance(self, p):
        if self.cyl is None:
            return np.full(p.shape[:-1], np.nan)
        cx, cy, cr = self.cyl
        return (np.sqrt((p[..., 0:1] - cx) ** 2 + (p[..., 1:2] - cy) ** 2) - cr).min(axis=-1)


def _oracle_flight(job):
    seed, cfg, periods, n_points, world_kw = job[:5]
    task_kw = (job[5] if len(job) > 5 else None) or {}   # dict(task="global_goal", global_goal=[3]): GetInitPath's other task (:34-45)
    map_kw = job[6] if len(job) > 6 else None     # dict(max_frame_count, th_dist, th_count, depth_min, cam): fly with the keyframe map;
    # cloud frames carry no camera pose: mCurFrame.Twc = Twb * T_b_c with Twb = the odometry position (R = I) and the yaml's camera
    # extrinsic (flight.TBC_YAML: the camera looks along the body's +x), as bench.py and tests/cpp/flight_driver.cpp do
    from tests import _oracle
    prm, n = make_prm(cfg) if isinstance(cfg, str) else (synth.MpcParams(T=cfg[0], K=cfg[1]), n_points)
    n = n_points or n
    if isinstance(seed, dict):   # explicit frames and start (oracle_flights_on_frames)
        world = _GivenFrames(seed["clouds"], seed["edges"], seed.get("cyl"))
        x, ref = seed["x0"].copy(), seed["ref0"].copy()
    else:
        world = flight.FlightWorld(seed, prm, n, **world_kw)
        x, ref = flight.initial_state(seed, prm)
    mpc = _oracle.MpcOracle(prm.T, prm.dt, prm.K); mpc.configure(prm)
    log = _log_arrays(periods)
    log["x"][0] = x
    kmap = None
    if map_kw:
        from tests import _kfmap
        kmap = _kfmap.MapOracle(map_kw["max_frame_count"], map_kw["th_dist"], map_kw["th_count"], map_kw["depth_min"], flight.TBC_YAML)
        log["n_keyframes"] = np.zeros(periods, np.int32); log["n_query_frames"] = np.zeros(periods, np.int32)
    for t in range(periods):
        cloud, edge = world.frame(t)
        sq, px = flight.period_inputs(x[None], ref[None], prm, task=task_kw.get("task", "forward"),
                                      global_goal=None if task_kw.get("global_goal") is None else np.asarray(task_kw["global_goal"])[None])
        if kmap is not None:
            Twc = cloud_frame_twc(x[None])[0]
            kmap.add_vertex(cloud, edge, Twc, stamp=t)
            kmap.update()
            log["n_keyframes"][t], log["n_query_frames"][t] = len(kmap.kfs), len(kmap.frames())
            r = kmap.step(mpc, prm, sq[0], px[0], ref, map_kw.get("cam"))
        else:
            kd, ke = _oracle.kd_oracle(cloud), _oracle.kd_oracle(edge)
            r = _oracle.step_oracle(kd, ke, mpc, prm, sq[0], px[0], ref)
            kd.close(); ke.close()
        a = flight.command(r["u"][None], r["flags"][None], x[None], prm)
        x = flight.apply_command(x[None], a, prm)[0]
        log["x"][t + 1] = x; log["u"][t] = r["u"]; log["flags"][t] = r["flags"]; log["cmd"][t] = a[0]
    mpc.close()
    log["clearance"] = world.clearance(log["x"][:, 0:3])
    return log


def _stack(logs):
    return {k: np.stack([l[k] for l in logs]) for k in logs[0]}


def _pool_map(fn, jobs, workers):
    workers = max(1, min(workers or (os.cpu_count() or 1), len(jobs)))
    if workers == 1:
        return [fn(j) for j in jobs]
    # one BLAS / OpenMP thread per worker: W workers x (all cores) LAPACK threads is 100 x slower than W x 1
    keys = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
    old = {k: os.environ.get(k) for k in keys}
    os.environ.update({k: "1" for k in keys})
    try:
        with mp.get_context("spawn").Pool(workers) as pool:
            return pool.map(fn, jobs, chunksize=1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def usable_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return cores


CLOUD_CAM = (32.0, 32.0, 32.0, 24.0, 100.0, 64, 48)   # PtIsInFrame's camera for cloud frames: the yaml's 640 x 480 / 10 sensor


def cloud_frame_twc(x):
    """mCurFrame.Twc of cloud frames, [n, 4, 4]: Twb * T_b_c (FrameKDMap.cpp:50) with Twb = [I | odometry position] and the yaml's
    extrinsic (mpc_parameters.yaml:67-71): the camera sits 5 cm ahead of the body's origin and looks along its +x."""
    x = np.asarray(x, np.float64)
    T = np.tile(flight.TBC_YAML, (len(x), 1, 1))
    T[:, :3, 3] += x[:, 0:3]
    return T


def oracle_flights(seeds, cfg="C2", periods=100, n_points=None, world_kw=None, workers=None, task_kw=None, keyframes=None):
    """-> dict of arrays [F, ...]: x [F, periods + 1, 10], u, flags, cmd, clearance [F, periods + 1].
    task_kw: dict(task="global_goal", global_goal=[F, 3]) flies GetInitPath's other task.
    keyframes: dict(max_frame_count, th_dist, th_count) flies with FrameKDMap's keyframe list (mCurFrame.Twc = cloud_frame_twc of
    the odometry; PtIsInFrame through CLOUD_CAM) and logs n_keyframes / n_query_frames."""
    from tests import _oracle
    _oracle.build_oracle()
    tk = lambda i: {} if not task_kw else dict(task=task_kw["task"], global_goal=None if task_kw.get("global_goal") is None else np.asarray(task_kw["global_goal"])[i])
    mk = dict(keyframes, depth_min=0.1, cam=CLOUD_CAM) if keyframes else None
    jobs = [(int(s), cfg, periods, n_points, world_kw or {}, tk(i), mk) for i, s in enumerate(seeds)]
    return _stack(_pool_map(_oracle_flight, jobs, workers or usable_cores()))


def oracle_flights_on_frames(clouds, edges, x0, ref0, T, K, cyl=None, workers=None, keyframes=None):
    """CPU-oracle flights on given frames: clouds [F, P, n, 3], edges [F, P, ne, 3] float32, x0 [F, 10], ref0 [F, N, 10];
    cyl: optional (cx, cy, cr) [F, ncyl] for the clearance.  -> the same dict of arrays as oracle_flights.
    keyframes: dict(max_frame_count, th_dist, th_count, depth_min, cam) flies with FrameKDMap's keyframe list."""
    from tests import _oracle
    _oracle.build_oracle()
    F, P = clouds.shape[0], clouds.shape[1]
    jobs = [(dict(clouds=clouds[f], edges=edges[f], x0=x0[f], ref0=ref0[f], cyl=None if cyl is None else tuple(c[f] for c in cyl)),
             (T, K), P, clouds.shape[2], {}, None, keyframes) for f in range(F)]
    return _stack(_pool_map(_oracle_flight, jobs, workers or usable_cores()))


def gpu_flights(seeds, cfg="C2", periods=100, n_points=None, world_kw=None, batch=None, tie_order=0, precision=64, mode="host",
                gang=1, task_kw=None, keyframes=None):
    """The same flights through amk_pipeline_*: one (slot, gang position) per batch of flights, one submit(keep_warm_start) per
    period.  mode "host": GetInitPath / clock model / command on the host (avoid_mpc_amd/flight.py), the pipeline gets
    state_quad, pos_x and the shifted path; mode "task": the pipeline's TASK mode -- the slot keeps mRefPath, the caller hands
    over the odometry and gets the command (prologue / epilogue kernels of csrc/pipeline.hip)."""
    import torch
    from avoid_mpc_amd.host import Pipeline
    prm, n = make_prm(cfg)
    n = n_points or n
    F = len(seeds)
    B = batch or F
    assert F % B == 0
    nb = F // B
    assert nb % gang == 0
    dev = torch.device("cuda", torch.cuda.current_device())
    worlds = [flight.FlightWorld(int(s), prm, n, **(world_kw or {})) for s in seeds]
    st = [flight.initial_state(int(s), prm) for s in seeds]
    x = np.stack([a for a, _ in st]); ref = np.stack([b for _, b in st])
    task = (task_kw or {}).get("task", "forward")
    goal = None if not task_kw or task_kw.get("global_goal") is None else np.ascontiguousarray(task_kw["global_goal"], np.float64)
    from avoid_mpc_amd import capi
    pl = Pipeline(nb // gang, B, n, n // 10, prm, queue_depth=1, gang=gang, task=task,
                  keyframes=dict(keyframes, depth_min=0.1, Tbc=flight.TBC_YAML) if keyframes else None)
    kcam = capi.FrameCamera(*CLOUD_CAM[:5], int(CLOUD_CAM[5]), int(CLOUD_CAM[6])) if keyframes else None
    if keyframes:
        logs_kf = dict(n_keyframes=np.zeros((F, periods), np.int32), n_query_frames=np.zeros((F, periods), np.int32))
    for i in range(nb // gang):
        pl.kd(i, 0).set_tie_order(tie_order); pl.kd(i, 1).set_tie_order(tie_order); pl.mpc(i).set_precision(precision)
    goal_d = None if goal is None else torch.from_numpy(goal).to(dev)
    logs = dict(x=np.zeros((F, periods + 1, 10)), u=np.zeros((F, periods, 4)), flags=np.zeros((F, periods, 4), np.int32),
                cmd=np.zeros((F, periods, 3)))
    logs["x"][:, 0] = x
    for t in range(periods):
        keep, tickets = [], []
        if mode == "host":
            sq, px = flight.period_inputs(x, ref, prm, task=task, global_goal=goal)
        for b in range(nb):   # every batch of the period in flight, then collect
            sl = slice(b * B, (b + 1) * B)
            fr = [worlds[i].frame(t) for i in range(sl.start, sl.stop)]
            clouds = torch.from_numpy(np.stack([c for c, _ in fr])).to(dev); edges = torch.from_numpy(np.stack([e for _, e in fr])).to(dev)
            Twc = None
            if keyframes:   # mCurFrame.Twc of the frame
                Twc = torch.from_numpy(cloud_frame_twc(x[sl])).to(dev)
            if mode == "host":
                bufs = (clouds, edges, torch.from_numpy(sq[sl]).to(dev), torch.from_numpy(px[sl]).to(dev), torch.from_numpy(ref[sl]).to(dev))
                keep.append(bufs + (Twc,))
                tickets.append(pl.submit(*bufs, keep_warm_start=t > 0, Twc_cur=Twc, cam=kcam))
            else:
                odom = torch.from_numpy(x[sl]).to(dev); cmd = torch.empty((B, 3), dtype=torch.float64, device=dev)
                ref0 = torch.from_numpy(ref[sl]).to(dev) if t == 0 else None     # InitCircleState's role; afterwards the slot's own
                keep.append((clouds, edges, odom, cmd, ref0, Twc))
                tickets.append(pl.submit(clouds, edges, ref_path_init=ref0, odom=odom, cmd_out=cmd, keep_warm_start=t > 0,
                                         global_goal=None if goal_d is None else goal_d[sl], Twc_cur=Twc, cam=kcam))
        for b, tk in enumerate(tickets):
            sl = slice(b * B, (b + 1) * B)
            pl.wait(tk)
            o = pl.outputs(tk)
            ref[sl] = o["ref_path"]
            a = flight.command(o["u"], o["flags"], x[sl], prm)
            if mode == "task":   # the device's command is the host twin's, bit for bit
                a_dev = keep[b][3].cpu().numpy()
                assert np.array_equal(a_dev, a), (t, b, np.abs(a_dev - a).max())
                a = a_dev
            x[sl] = flight.apply_command(x[sl], a, prm)
            logs["u"][sl, t] = o["u"]; logs["flags"][sl, t] = o["flags"]; logs["cmd"][sl, t] = a
            if keyframes:
                ms = pl.kfmap_state(tk % pl.n_slots)
                ps = slice((tk // pl.n_slots) * B, (tk // pl.n_slots + 1) * B)
                logs_kf["n_keyframes"][sl, t] = ms["n_keyframes"][ps]; logs_kf["n_query_frames"][sl, t] = ms["n_query_frames"][ps]
        logs["x"][:, t + 1] = x
    pl.close()
    if keyframes:
        logs.update(logs_kf)
    logs["clearance"] = np.stack([w.clearance(logs["x"][i, :, 0:3]) for i, w in enumerate(worlds)])
    return logs


# ---- the reference's solver regime, emulated: IPOPT stopped after 10 iterations, primal warm start -------------------------------
def _ipopt_flight(job):
    seed, cfg, periods, n_points, world_kw, max_iter = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ipopt_emul as IE
    from tests import _oracle
    prm, n = make_prm(cfg)
    n = n_points or n
    N, K = prm.N, prm.K
    world = flight.FlightWorld(seed, prm, n, **world_kw)
    x, ref = flight.initial_state(seed, prm)
    lbu = np.array([-prm.a_max_xy, -prm.a_max_xy, prm.a_min_z, -prm.a_max_yaw_dot])
    ubu = np.array([prm.a_max_xy, prm.a_max_xy, prm.a_max_z, prm.a_max_yaw_dot])
    tail = np.concatenate([prm.gain, prm.tau, prm.weights, [prm.radius]])   # HighLvlMpc.cpp:97-107
    w0 = np.zeros(10 + 14 * N)                                              # mNlpW0, HighLvlMpc.cpp:26-27,35
    log = _log_arrays(periods)
    log["x"][0] = x
    log["ipopt_iters"] = np.zeros(periods, np.int32); log["ipopt_status"] = np.zeros((periods, prm.max_iter), np.int32) - 1
    lib = _oracle.load_oracle()
    for t in range(periods):
        cloud, edge = world.frame(t)
        kd, ke = _oracle.kd_oracle(cloud), _oracle.kd_oracle(edge)
        sq, px = flight.period_inputs(x[None], ref[None], prm)
        # TASK branch (:322-355) with the emulated solver in the place of ObstacleAvoidanceMPC::Solve: the oracle's step with
        # mpc_max_iter = 1 would hide the solver, so the loop is restated here on the oracle's KD queries
        u = np.zeros(4); is_safety = 1; solves = 0; iters = 0
        for it in range(prm.max_iter):
            is_safety = 1
            p1 = ref[0, 0:3]
            i1, d1, _ = kd.search(p1, 1)
            nd = np.sqrt(d1[0]) if len(d1) else np.sqrt(np.finfo(np.float64).max)
            if not (nd > prm.safety_distance):
                ie, de, pe = ke.search(p1, 1)
                if len(ie) == 0:
                    is_safety = 0
                else:
                    ref[0, 0:3] = pe[0]
            obst = np.full((N, K, 3), 10000.0); need = False
            for i in range(N):
                ii, dd, pp = kd.search(ref[i, 0:3], K)
                obst[i, :len(ii)] = pp
                if len(ii) == 0 or np.sqrt(dd[0]) <= prm.safety_distance:
                    need = True
            if not need and it > 0 and is_safety:
                break
            tg = ref[N - 1].copy()
            tg[0] += max(0.0, prm.speed * prm.T - max(0.0, tg[0] - px[0])); tg[1] = 0.0
            P = np.concatenate([sq[0, it], ref.reshape(-1), obst.reshape(-1), tg, tail])
            nlp = IE.ShootingNlp(P, N, K, prm.dt, lbu, ubu)
            res = IE.solve(nlp, w0, max_iter=max_iter)
            w0 = res["x"]; u = w0[10:14].copy()
            log["ipopt_status"][t, it] = res["status"]; iters += res["iters"]; solves += 1
            ref[:] = w0[:14 * N].reshape(N, 14)[:, :10]
        kd.close(); ke.close()
        flags = np.array([is_safety, solves, 0, iters], np.int32)
        a = flight.command(u[None], flags[None], x[None], prm)
        x = flight.apply_command(x[None], a, prm)[0]
        log["x"][t + 1] = x; log["u"][t] = u; log["flags"][t] = flags; log["cmd"][t] = a[0]; log["ipopt_iters"][t] = iters
    log["clearance"] = world.clearance(log["x"][:, 0:3])
    del lib
    return log


def ipopt_flights(seeds, cfg="C2", periods=100, n_points=None, world_kw=None, workers=None, max_iter=10):
    jobs = [(int(s), cfg, periods, n_points, world_kw or {}, max_iter) for s in seeds]
    return _stack(_pool_map(_ipopt_flight, jobs, workers or usable_cores()))


# ---- comparison ----------------------------------------------------------------------------------------------------------------
def compare(a, b, pos_tol=1e-6):
    """Period-by-period comparison of two sets of flights (dicts of [F, ...] arrays).  A flight SEPARATES at the first period
    whose flags differ or whose position differs by more than pos_tol.  -> dict of statistics"""
    F, P = a["u"].shape[0], a["u"].shape[1]
    dpos = np.abs(a["x"][:, :, 0:3] - b["x"][:, :, 0:3]).max(axis=2)        # [F, P + 1]
    flag_diff = np.any(a["flags"] != b["flags"], axis=2)                    # [F, P]
    sep = np.full(F, -1)
    for f in range(F):
        bad = np.nonzero(flag_diff[f] | (dpos[f, 1:] > pos_tol))[0]
        if bad.size:
            sep[f] = bad[0]
    together = sep < 0
    dpos_before = np.array([dpos[f, :(sep[f] + 1 if sep[f] >= 0 else P + 1)].max() for f in range(F)])
    return dict(flights=F, periods=P, separated=int((~together).sum()), separation_period=sep,
                dpos_max_while_together=float(dpos_before.max()), dpos_final=dpos[:, -1],
                dpos_final_max_separated=float(dpos[~together, -1].max()) if (~together).any() else 0.0,
                du_max_together=float(np.abs(a["u"] - b["u"])[together].max()) if together.any() else None)


def flight_stats(log, prm, con_dt=None):
    """Clearance / collision statistics of a set of flights: clearance = horizontal distance from the vehicle's centre to the
    nearest cylinder surface; a flight collides when it drops below the drone radius."""
    cl = log["clearance"].min(axis=1)
    fl = log["flags"]
    radius = prm.radius
    t = np.arange(log["x"].shape[1])
    lag = log["x"][:, :, 0] - log["x"][:, :1, 0] - prm.speed * (con_dt or prm.dt) * t[None]   # flown x against the frames' nominal x
    return dict(flights=int(cl.size), min_clearance_median=float(np.median(cl)), min_clearance_min=float(cl.min()),
                collided=int((cl < radius).sum()), hit_surface=int((cl < 0).sum()),
                unsafe_periods=int((fl[:, :, 0] == 0).sum()), solves_per_period=float(fl[:, :, 1].mean()),
                iters_per_period=float(fl[:, :, 3].mean()), capped_periods=int((fl[:, :, 2] > 0).sum()),
                x_final_mean=float(log["x"][:, -1, 0].mean()), x_behind_nominal_max=float(-lag.min()),
                x_ahead_of_nominal_max=float(lag.max()))


# ---- flights from the raw depth image: the reference's real regime (640 x 480 / 10 -> 3072-point frames in the yaml; here a
# 320 x 240 sensor / 5), FrameKDMap::AddVertex's ProcessDepth + BuildEdgeCloud included (FrameKDMap.cpp:34-52,75-214) --------------
DEPTH_CAM = dict(rows=240, cols=320, pixel2meter=1e-3, depth_min=0.1, depth_max=60.0, resize_scale=5.0, fx=160.0, fy=160.0, cx=160.0,
                 cy=120.0, Tbc=flight.TBC_YAML)


YAML_CAM = dict(rows=480, cols=640, pixel2meter=1e-3, depth_min=0.1, depth_max=100.0, resize_scale=10.0, fx=320.0, fy=320.0, cx=320.0,
                cy=240.0, Tbc=flight.TBC_YAML)   # mpc_parameters.yaml:59-70: the reference's own sensor, 640 x 480 / 10 -> <= 3072 points


def _depth_frame(world, x, cam=None):
    """(depth uint16 [rows, cols] in millimetres, Twb) as the depth callback would hand them over: the pose is the odometry rounded
    to a micrometre, so that two drivers whose states agree to 1e-9 m render identical images."""
    c = cam or DEPTH_CAM
    Twb = np.eye(4); Twb[:3, 3] = np.round(x[0:3], 6)          # Command.yaw = 0 holds the heading: R = I
    sel = (world.cx > x[0] - 2.0) & (world.cx < x[0] + 40.0)
    d = flight.render_depth((world.cx[sel], world.cy[sel], world.cr[sel]), Twb, c["Tbc"], c["rows"], c["cols"], c["fx"], c["fy"], c["cx"], c["cy"])
    return np.clip(np.round(d / c["pixel2meter"]), 0, 65535).astype(np.uint16), Twb


def depth_camera(cam=None):
    """PtIsInFrame's camera for DEPTH_CAM: the intrinsics divided by the resize scale (FrameKDMap.cpp:21-24), the down-scaled image
    size of ProcessDepth (:106-107).  -> (fx, fy, cx, cy, depth_max, width, height)"""
    c = cam or DEPTH_CAM
    sc = c["resize_scale"]
    return (c["fx"] / sc, c["fy"] / sc, c["cx"] / sc, c["cy"] / sc, c["depth_max"], int(c["cols"] / sc), int(c["rows"] / sc))


def _oracle_depth_flight(job):
    seed, cfg, periods, world_kw = job[:4]
    map_kw = job[4] if len(job) > 4 else None     # dict(max_frame_count, th_dist, th_count): fly with the keyframe map
    DEPTH_CAM = (job[5] if len(job) > 5 else None) or globals()["DEPTH_CAM"]   # the sensor (default: the test sensor)
    from tests import _oracle
    prm, _ = make_prm(cfg)
    world = flight.FlightWorld(seed, prm, 1000, **world_kw)
    x, ref = flight.initial_state(seed, prm)
    mpc = _oracle.MpcOracle(prm.T, prm.dt, prm.K); mpc.configure(prm)
    log = _log_arrays(periods)
    log["x"][0] = x
    log["n_cloud"] = np.zeros(periods, np.int32); log["n_edge"] = np.zeros(periods, np.int32)
    Twc = np.eye(4); kd = ke = None
    kmap = None
    if map_kw:
        from tests import _kfmap
        kmap = _kfmap.MapOracle(map_kw["max_frame_count"], map_kw["th_dist"], map_kw["th_count"], DEPTH_CAM["depth_min"], DEPTH_CAM["Tbc"])
        log["n_keyframes"] = np.zeros(periods, np.int32); log["n_query_frames"] = np.zeros(periods, np.int32)
        log["outliers"] = np.zeros(periods, np.int32); log["map_points"] = np.zeros(periods, np.int64)
    for t in range(periods):
        img, Twb = _depth_frame(world, x, DEPTH_CAM)
        cloud, _ = _oracle.depth_oracle(img, DEPTH_CAM, Twb)
        if kmap is not None:
            if len(cloud):                                      # AddVertex (:39-51), then KeyframeThreadWorker's body (:443-486)
                edge = _oracle.depth_edge_oracle(img, DEPTH_CAM, kmap.Twc)[0]   # the stale pose (:209)
                kmap.add_vertex(cloud, edge, Twb @ DEPTH_CAM["Tbc"], stamp=t)
            kmap.update()
            nk, sizes = kmap.summary()
            log["n_keyframes"][t] = nk; log["n_query_frames"][t] = len(sizes); log["outliers"][t] = max(kmap.last_outliers, 0)
            log["map_points"][t] = int(np.sum(sizes))
            log["n_cloud"][t] = kmap.cur.kd.size(); log["n_edge"][t] = kmap.cur.ke.size()
            sq, px = flight.period_inputs(x[None], ref[None], prm)
            r = kmap.step(mpc, prm, sq[0], px[0], ref, depth_camera(DEPTH_CAM))
        else:
            if len(cloud):                                          # AddVertex, FrameKDMap.cpp:39-51
                edge = _oracle.depth_edge_oracle(img, DEPTH_CAM, Twc)[0]
                if kd is not None:
                    kd.close(); ke.close()
                kd, ke = _oracle.kd_oracle(cloud), _oracle.kd_oracle(edge)
                Twc = Twb @ DEPTH_CAM["Tbc"]
            log["n_cloud"][t] = kd.size(); log["n_edge"][t] = ke.size()
            sq, px = flight.period_inputs(x[None], ref[None], prm)
            r = _oracle.step_oracle(kd, ke, mpc, prm, sq[0], px[0], ref)
        a = flight.command(r["u"][None], r["flags"][None], x[None], prm)
        x = flight.apply_command(x[None], a, prm)[0]
        log["x"][t + 1] = x; log["u"][t] = r["u"]; log["flags"][t] = r["flags"]; log["cmd"][t] = a[0]
    log["clearance"] = world.clearance(log["x"][:, 0:3])
    return log


def oracle_depth_flights(seeds, cfg="C1", periods=40, world_kw=None, workers=None, keyframes=None, cam=None):
    """keyframes: dict(max_frame_count, th_dist, th_count) flies with FrameKDMap's keyframe list (tests/_kfmap.py) and logs
    n_keyframes / n_query_frames / outliers / map_points per period."""
    from tests import _oracle
    _oracle.build_oracle()
    jobs = [(int(s), cfg, periods, world_kw or {}, keyframes, cam) for s in seeds]
    return _stack(_pool_map(_oracle_depth_flight, jobs, workers or usable_cores()))


def gpu_depth_flights(seeds, cfg="C1", periods=40, world_kw=None, gang=1, batch=None, keyframes=None, cam=None):
    """The same flights through amk_pipeline frames that START at the depth image (d_depth + d_Twb) in TASK mode.
    keyframes: dict(max_frame_count, th_dist, th_count): every slot keeps a keyframe map (amk_pipeline_config.keyframes)."""
    import torch
    from avoid_mpc_amd.host import Pipeline, depth_params
    prm, _ = make_prm(cfg)
    c = cam or DEPTH_CAM
    F = len(seeds); B = batch or F; nb = F // B
    assert F % B == 0 and nb % gang == 0
    dev = torch.device("cuda", torch.cuda.current_device())
    worlds = [flight.FlightWorld(int(s), prm, 1000, **(world_kw or {})) for s in seeds]
    st = [flight.initial_state(int(s), prm) for s in seeds]
    x = np.stack([a for a, _ in st]); ref0 = np.stack([b for _, b in st])
    cap = int(c["cols"] / c["resize_scale"]) * int(c["rows"] / c["resize_scale"])
    dp = depth_params(c["pixel2meter"], c["depth_min"], c["depth_max"], c["resize_scale"], c["fx"], c["fy"], c["cx"], c["cy"], c["Tbc"])
    pl = Pipeline(nb // gang, B, cap, cap, prm, queue_depth=1, gang=gang, depth=dp,
                  keyframes=dict(keyframes, depth_min=c["depth_min"]) if keyframes else None)
    logs = dict(x=np.zeros((F, periods + 1, 10)), u=np.zeros((F, periods, 4)), flags=np.zeros((F, periods, 4), np.int32),
                cmd=np.zeros((F, periods, 3)))
    if keyframes:
        logs.update(n_keyframes=np.zeros((F, periods), np.int32), n_query_frames=np.zeros((F, periods), np.int32),
                    outliers=np.zeros((F, periods), np.int32), map_points=np.zeros((F, periods), np.int64),
                    n_cloud=np.zeros((F, periods), np.int32))
    logs["x"][:, 0] = x
    for t in range(periods):
        keep, tickets = [], []
        for b in range(nb):
            sl = slice(b * B, (b + 1) * B)
            fr = [_depth_frame(worlds[i], x[i], c) for i in range(sl.start, sl.stop)]
            depth = torch.from_numpy(np.stack([d for d, _ in fr]).view(np.int16)).to(dev)
            Twb = torch.from_numpy(np.stack([T for _, T in fr])).to(dev)
            odom = torch.from_numpy(x[sl]).to(dev); cmd = torch.empty((B, 3), dtype=torch.float64, device=dev)
            r0 = torch.from_numpy(ref0[sl]).to(dev) if t == 0 else None
            keep.append((depth, Twb, odom, cmd, r0))
            tickets.append(pl.submit(None, None, ref_path_init=r0, odom=odom, cmd_out=cmd, keep_warm_start=t > 0, depth=depth, Twb=Twb))
        for b, tk in enumerate(tickets):
            sl = slice(b * B, (b + 1) * B)
            pl.wait(tk)
            o = pl.outputs(tk)
            a = keep[b][3].cpu().numpy()
            x[sl] = flight.apply_command(x[sl], a, prm)
            logs["u"][sl, t] = o["u"]; logs["flags"][sl, t] = o["flags"]; logs["cmd"][sl, t] = a
            if keyframes:   # batch b = position b % gang of slot b // gang ... in submission order: slot = ticket % n_slots
                slot, pos = tk % pl.n_slots, tk // pl.n_slots
                ms = pl.kfmap_state(slot)
                ps = slice(pos * B, (pos + 1) * B)
                logs["n_keyframes"][sl, t] = ms["n_keyframes"][ps]; logs["n_query_frames"][sl, t] = ms["n_query_frames"][ps]
                logs["outliers"][sl, t] = ms["last_outliers"][ps]
                logs["map_points"][sl, t] = np.maximum(ms["frame_sizes"][ps], 0).sum(axis=1); logs["n_cloud"][sl, t] = ms["frame_sizes"][ps, 0]
        logs["x"][:, t + 1] = x
    pl.close()
    logs["clearance"] = np.stack([w.clearance(logs["x"][i, :, 0:3]) for i, w in enumerate(worlds)])
    return logs
