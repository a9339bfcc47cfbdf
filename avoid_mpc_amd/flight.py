"""BENCH AND TEST HARNESS (not part of the product path): closed-loop flights around the control step -- the world, the sensor
frames, the vehicle and the host twins of the TASK prologue / epilogue that bench.py --workload flight and tests/_flight.py share.
Nothing under csrc/ or include/ depends on it.

The reference is a 30 Hz loop (timer of con_dt = 0.033 s, AM/src/AvoidanceStateMachine.cpp:110-111,
AM/launch/mpc_obstacle_avoidance_sim.launch:8): every period takes a fresh depth frame
(DepthCallback -> FrameKDMap::AddVertex, :153-164), shifts mRefPath (GetInitPath, :24-54), re-reads the odometry
(GetCurStateQuad, :183-203), runs the TASK branch (:322-355) from the previous period's solution
(mNlpW0 = sol, AM/src/HighLvlMpc.cpp:129) and publishes the first control (PubCmd :369-378) or the slow-down
command (PubSlowDownCmd :379-397).  This module holds what a flight needs AROUND the step and is shared by the
GPU driver (tests/_flight.py, bench.py --workload flight) and the CPU-oracle driver (tests/_flight.py):

  plant_step / affine_plant   the vehicle = the MPC's own model (mpc_obstacle_casadi.py:106-122 integrated by :338-357),
                              driven by the published command; the reference flies AirSim + bfctrl, absent here
  FlightWorld                 a static corridor of vertical cylinders; frame(t) = what a forward-looking sensor
                              returns in period t: surface points RESAMPLED every period (fresh frame, fresh indices)
  period_inputs / apply_command   GetInitPath + the clock model on one side of the step, PubCmd / PubSlowDownCmd on the other

Nothing here computes neighbours or solves anything: that is the step's job (amk_step_batch / the oracle's stepo_run).
"""
import numpy as np

from . import fsm, synth

GZ = 9.81   # mpc_obstacle_casadi.py:33 (self._gz)


def model_f(x, u, tau):
    """x_dot of mpc_obstacle_casadi.py:106-122 (drag off): x = [p(3), yaw, v(3), a(3)], u = [a_cmd(3), yaw_dot].
    Vectorised over leading dimensions."""
    x, u = np.asarray(x, np.float64), np.asarray(u, np.float64)
    d = np.empty_like(x)
    d[..., 0:3] = x[..., 4:7]
    d[..., 3] = u[..., 3]
    d[..., 4:7] = x[..., 7:10]
    d[..., 7] = (u[..., 0] - x[..., 7]) * tau[0]
    d[..., 8] = (u[..., 1] - x[..., 8]) * tau[1]
    d[..., 9] = (u[..., 2] - GZ - x[..., 9]) * tau[2]
    return d


def plant_step(x, u, tau, dt, refine=4):
    """F(x, u): `refine` classic RK4 sub-steps over dt (sys_dynamics, mpc_obstacle_casadi.py:338-357)."""
    h = dt / refine
    x = np.array(x, np.float64)
    for _ in range(refine):
        k1 = h * model_f(x, u, tau)
        k2 = h * model_f(x + 0.5 * k1, u, tau)
        k3 = h * model_f(x + 0.5 * k2, u, tau)
        k4 = h * model_f(x + k3, u, tau)
        x = x + (k1 + 2 * k2 + 2 * k3 + k4) / 6
    return x


def affine_plant(tau, dt):
    """(A [10,10], B [10,4], c [10]) with F(x, u) = A x + B u + c -- exact, the model is affine for fixed tau."""
    c = plant_step(np.zeros(10), np.zeros(4), tau, dt)
    A = np.stack([plant_step(e, np.zeros(4), tau, dt) - c for e in np.eye(10)], axis=1)
    B = np.stack([plant_step(np.zeros(10), e, tau, dt) - c for e in np.eye(4)], axis=1)
    return A, B, c


class FlightWorld:
    """One flight's world: vertical cylinders scattered over a corridor (x from `x_first` on, |y| <= 8 m, 4 m high), the
    obstacle model of synth.make_cloud stretched along the flight.  frame(t) is the sensor return of period t around the
    NOMINAL position x_nom(t) = x0 + speed * con_dt * t: surface points of the cylinders whose axis lies in
    [x_nom - back, x_nom + ahead], resampled every period (rng seeded by (seed, t): the GPU driver and the CPU oracle see
    identical frames whatever their trajectories do), plus ground returns.  The window follows the nominal position, not the
    flown one, so that a frame is a pure function of (seed, t); drivers report how far a flight strays from it."""

    def __init__(self, seed, prm, n_points, cyl_per_m=1.0, length=80.0, x_first=8.0, back=6.0, ahead=30.0, ground_frac=0.15,
                 con_dt=None):
        self.seed, self.prm, self.n = int(seed), prm, int(n_points)
        self.back, self.ahead, self.ground_frac = float(back), float(ahead), float(ground_frac)
        self.con_dt = prm.dt if con_dt is None else float(con_dt)
        rng = np.random.default_rng([self.seed, 0xF11])
        ncyl = max(1, int(round(cyl_per_m * (length - x_first))))
        self.cx = np.sort(rng.uniform(x_first, length, ncyl))
        self.cy = rng.uniform(-8.0, 8.0, ncyl)
        self.cr = rng.uniform(0.1, 0.5, ncyl)

    def x_nom(self, t):
        return self.prm.speed * self.con_dt * t

    def frame(self, t):
        """-> (cloud float32 [n, 3], edge float32 [n // 10, 3]) of period t."""
        n, ne = self.n, self.n // 10
        rng = np.random.default_rng([self.seed, 0xF12, int(t)])
        xn = self.x_nom(t)
        lo, hi = xn - self.back, xn + self.ahead
        sel = np.nonzero((self.cx >= lo) & (self.cx <= hi))[0]
        if sel.size == 0:   # open space: everything is a ground return
            sel = None
        n_g = n if sel is None else int(self.ground_frac * n)
        n_c = n - n_g
        parts = []
        if n_c:
            ci = sel[rng.integers(0, sel.size, n_c)]
            th = rng.uniform(0.0, 2.0 * np.pi, n_c)
            parts.append(np.stack([self.cx[ci] + self.cr[ci] * np.cos(th), self.cy[ci] + self.cr[ci] * np.sin(th),
                                   rng.uniform(0.0, 4.0, n_c)], axis=1))
        parts.append(np.stack([rng.uniform(lo, hi, n_g), rng.uniform(-8.0, 8.0, n_g), np.zeros(n_g)], axis=1))
        cloud = np.concatenate(parts).astype(np.float32)
        cloud = cloud[rng.permutation(n)]
        # edge cloud: the silhouette lines of the cylinders ahead, seen from the nominal camera position
        if sel is None:
            edge = np.stack([rng.uniform(lo, hi, ne), rng.uniform(-8.0, 8.0, ne), np.zeros(ne)], axis=1).astype(np.float32)
            return cloud, edge
        ahead = sel[self.cx[sel] > xn + 0.6]
        src = ahead if ahead.size else sel
        ei = src[rng.integers(0, src.size, ne)]
        side = rng.integers(0, 2, ne) * 2 - 1
        dx, dy = self.cx[ei] - xn, self.cy[ei]
        dist = np.sqrt(dx * dx + dy * dy)
        ang = np.arctan2(dy, dx) + side * (np.pi / 2.0 + np.arcsin(np.clip(self.cr[ei] / dist, -1.0, 1.0)))
        edge = np.stack([self.cx[ei] + self.cr[ei] * np.cos(ang), self.cy[ei] + self.cr[ei] * np.sin(ang),
                         rng.uniform(0.0, 4.0, ne)], axis=1).astype(np.float32)
        return cloud, edge

    def clearance(self, p):
        """Distance from position(s) p [..., 3] to the nearest cylinder SURFACE in the horizontal plane (negative: inside)."""
        p = np.asarray(p, np.float64)
        d = np.sqrt((p[..., 0:1] - self.cx) ** 2 + (p[..., 1:2] - self.cy) ** 2) - self.cr
        return d.min(axis=-1)


def initial_state(seed, prm):
    """Plant state [p, yaw, v, a] at t = 0 and mRefPath as GetInitPath's "forward" task leaves it in cruise
    (synth.make_odom / make_ref_path: the bench scenes' start)."""
    pos, vel, acc, yaw = synth.make_odom(seed, prm)
    x = np.concatenate([pos, [yaw], vel, acc])
    return x, synth.make_ref_path(pos, prm)


def period_inputs(x, ref_path, prm, farest=500.0, shift=True, task="forward", global_goal=None):
    """Host side of one period BEFORE the step, for a batch: x [S, 10] plant states (= odometry: perfect sensing),
    ref_path [S, N, 10] in/out.  GetInitPath (:24-54; task "forward", goal_x = 500 of mpc_parameters.yaml:55), then the
    per-pass initial states of the clock model (fsm.state_quads).  -> (state_quad [S, max_iter, 10], pos_x [S])"""
    S = x.shape[0]
    sq = np.empty((S, prm.max_iter, 10))
    for s in range(S):
        if shift:
            fsm.get_init_path(ref_path[s], prm.speed, prm.T, x[s, 0], farest, prm.height, task=task,
                              global_goal=None if global_goal is None else global_goal[s], dt=prm.dt)
        sq[s] = fsm.state_quads(x[s, 0:3], x[s, 4:7], x[s, 7:10], x[s, 3], prm.decay, prm.max_iter)
    return sq, x[:, 0].copy()


def command(u, flags, x, prm, kp=0.3, kd=0.3):
    """What the node publishes after the step (:345-350): PubCmd(u) when isSafety, else PubSlowDownCmd (:379-397;
    slow_down_kp / kd of mpc_parameters.yaml:79-80; z clamped to +-aMaxZ like the reference).  Batched:
    u [S, 4], flags [S, 4], x [S, 10] -> a_cmd [S, 3]"""
    u, x = np.asarray(u, np.float64), np.asarray(x, np.float64)
    slow = -kp * x[:, 4:7] - kd * x[:, 7:10] + np.array([0.0, 0.0, 9.8])
    slow[:, 0:2] = np.clip(slow[:, 0:2], -prm.a_max_xy, prm.a_max_xy)
    slow[:, 2] = np.clip(slow[:, 2], -prm.a_max_z, prm.a_max_z)
    safe = (np.asarray(flags)[:, 0] != 0)[:, None]
    return np.where(safe, u[:, 0:3], slow)


def apply_command(x, a_cmd, prm, con_dt=None):
    """The vehicle over one control period: the model driven by the published acceleration command; Command.yaw = 0
    (:376) holds the heading, so yaw_dot = 0."""
    uu = np.concatenate([a_cmd, np.zeros((a_cmd.shape[0], 1))], axis=1)
    return plant_step(x, uu, prm.tau, prm.dt if con_dt is None else con_dt)


class FlightWorldsTorch:
    """S flights' worlds and frames generated ON the device (bench.py --workload flight: frames are made before the clock starts
    and stay resident in HBM).  The scene model of FlightWorld -- vertical cylinders along a corridor, a window of surface points
    around the nominal position resampled every period, ground returns, silhouette points as the edge cloud -- on torch's random
    stream (values differ from the numpy generator's; the drivers that compare GPU and CPU flights hand the SAME frames to
    both)."""

    def __init__(self, S, n_points, prm, seed, device, cyl_per_m=1.5, length=80.0, x_first=8.0, back=6.0, ahead=30.0,
                 ground_frac=0.15, con_dt=None):
        import torch
        self.S, self.n, self.prm, self.seed, self.dev = int(S), int(n_points), prm, int(seed), device
        self.back, self.ahead, self.ground_frac = float(back), float(ahead), float(ground_frac)
        self.con_dt = prm.dt if con_dt is None else float(con_dt)
        g = torch.Generator(device=device); g.manual_seed(self.seed)
        self.ncyl = max(1, int(round(cyl_per_m * (length - x_first))))
        r = lambda: torch.rand((self.S, self.ncyl), generator=g, device=device, dtype=torch.float64)
        self.cx = torch.sort(x_first + (length - x_first) * r(), dim=1).values.contiguous()
        self.cy = -8.0 + 16.0 * r()
        self.cr = 0.1 + 0.4 * r()

    def x_nom(self, t):
        return self.prm.speed * self.con_dt * t

    def frame(self, t):
        """-> (cloud float32 [S, n, 3], edge float32 [S, n // 10, 3]) of period t."""
        import math
        import torch
        S, n, ne, dev = self.S, self.n, self.n // 10, self.dev
        g = torch.Generator(device=dev); g.manual_seed(self.seed * 1000003 + 7919 * int(t) + 1)
        rnd = lambda *shape: torch.rand(shape, generator=g, device=dev, dtype=torch.float32)
        xn = self.x_nom(t)
        lo, hi = xn - self.back, xn + self.ahead
        lo_i = torch.searchsorted(self.cx, torch.full((S, 1), lo, device=dev, dtype=torch.float64))
        hi_i = torch.searchsorted(self.cx, torch.full((S, 1), hi, device=dev, dtype=torch.float64), right=True)
        cnt = (hi_i - lo_i).clamp_(min=1)
        n_g = int(self.ground_frac * n); n_c = n - n_g
        ci = (lo_i + (rnd(S, n_c) * cnt).long()).clamp_(max=self.ncyl - 1)
        th = 2.0 * math.pi * rnd(S, n_c)
        gx, gy, gr = (v.gather(1, ci).float() for v in (self.cx, self.cy, self.cr))
        on_cyl = torch.stack([gx + gr * torch.cos(th), gy + gr * torch.sin(th), 4.0 * rnd(S, n_c)], dim=2)
        ground = torch.stack([lo + (hi - lo) * rnd(S, n_g), -8.0 + 16.0 * rnd(S, n_g), torch.zeros((S, n_g), device=dev)], dim=2)
        cloud = torch.cat([on_cyl, ground], dim=1)
        perm = torch.argsort(rnd(S, n), dim=1)
        cloud = cloud.gather(1, perm[:, :, None].expand(S, n, 3)).contiguous()
        a_i = torch.searchsorted(self.cx, torch.full((S, 1), xn + 0.6, device=dev, dtype=torch.float64), right=True)
        a_i = torch.minimum(a_i, hi_i - 1).clamp_(min=0)
        cnt2 = (hi_i - a_i).clamp_(min=1)
        ei = (a_i + (rnd(S, ne) * cnt2).long()).clamp_(max=self.ncyl - 1)
        side = (rnd(S, ne) < 0.5).float() * 2.0 - 1.0
        ex0, ey0, er = (v.gather(1, ei).float() for v in (self.cx, self.cy, self.cr))
        dx = ex0 - xn
        dist = torch.sqrt(dx * dx + ey0 * ey0)
        ang = torch.atan2(ey0, dx) + side * (math.pi / 2.0 + torch.asin((er / dist).clamp(-1.0, 1.0)))
        edge = torch.stack([ex0 + er * torch.cos(ang), ey0 + er * torch.sin(ang), 4.0 * rnd(S, ne)], dim=2).contiguous()
        return cloud, edge

    def clearance(self, pos):
        """pos float64 [S, P, 3] (numpy) -> horizontal distance to the nearest cylinder surface [S, P]."""
        cx, cy, cr = (v.cpu().numpy() for v in (self.cx, self.cy, self.cr))
        d = np.sqrt((pos[:, :, 0:1] - cx[:, None, :]) ** 2 + (pos[:, :, 1:2] - cy[:, None, :]) ** 2) - cr[:, None, :]
        return d.min(axis=2)


# T_b_c of AM/config/mpc_parameters.yaml:67-71: camera (x right, y down, z forward) in the body frame (x forward, y left, z up)
TBC_YAML = np.array([[0.0, 0.0, 1.0, 0.05], [-1.0, 0.0, 0.0, 0.0], [0.0, -1.0, 0.0, 0.01], [0.0, 0.0, 0.0, 1.0]])


def render_depth(cyl, Twb, Tbc, rows, cols, fx, fy, cx, cy, z_top=4.0, max_range=60.0):
    """Planar depth image (metres, float32 [rows, cols]; 0 = no return) of vertical cylinders cyl = (cx, cy, r) arrays standing
    on the ground plane z = 0, seen by a pinhole camera at Twb * Tbc -- the synthetic counterpart of the depth frames the
    reference receives (AirSim's DepthPlanar, AM/src/AvoidanceStateMachine.cpp:153-164).  Pixel (u, v) looks along
    ((u - cx) / fx, (v - cy) / fy, 1) in the camera frame; the value is the z coordinate of the first hit in that frame, which is
    what FrameKDMap::UV2Camera back-projects (FrameKDMap.cpp:131-138)."""
    ccx, ccy, cr = (np.asarray(v, np.float64) for v in cyl)
    Twc = np.asarray(Twb, np.float64).reshape(4, 4) @ np.asarray(Tbc, np.float64).reshape(4, 4)
    R, o = Twc[:3, :3], Twc[:3, 3]
    v, u = np.meshgrid(np.arange(rows, dtype=np.float64), np.arange(cols, dtype=np.float64), indexing="ij")
    dc = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)      # camera-frame ray, z component 1: t = planar depth
    d = dc @ R.T                                                                    # world-frame ray
    best = np.full((rows, cols), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(d[..., 2] < 0, -o[2] / d[..., 2], np.inf)                     # ground plane
        best = np.minimum(best, np.where(tg > 0, tg, np.inf))
        a = d[..., 0] ** 2 + d[..., 1] ** 2
        for k in range(ccx.size):
            ox, oy = o[0] - ccx[k], o[1] - ccy[k]
            b = ox * d[..., 0] + oy * d[..., 1]
            c = ox * ox + oy * oy - cr[k] ** 2
            disc = b * b - a * c
            t = (-b - np.sqrt(np.where(disc >= 0, disc, np.nan))) / a
            z = o[2] + t * d[..., 2]
            ok = (disc >= 0) & (t > 0) & (z >= 0) & (z <= z_top)
            best = np.where(ok & (t < best), t, best)
    return np.where(best <= max_range, best, 0.0).astype(np.float32)
