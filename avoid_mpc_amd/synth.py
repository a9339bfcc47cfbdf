"""Deterministic synthetic inputs for the hot path.

The reference has no dataset: its clouds come from a depth camera through
FrameKDMap::ProcessDepth (AM/src/FrameKDMap.cpp:90-130).  BASELINE.json's configs are synthetic
scale-ups, generated here on the host with numpy (seed = base + scene id):

  * obstacle cloud: float32 xyz in a 30 m x 16 m x 4 m corridor, 70 % of the points on 20-60
    random vertical cylinders (r in [0.1, 0.5] m), 30 % uniform clutter, continuous coordinates;
  * edge cloud: n/10 points on the cylinders' silhouettes as seen from the origin (stand-in for
    the Canny edges of FrameKDMap::BuildEdgeCloud, FrameKDMap.cpp:176-214);
  * odometry: position (0, y0, height), forward speed `speed`, zero acceleration, yaw 0;
  * reference path: straight along +x at spacing speed*dt -- what GetInitPath's "forward" task
    converges to (AM/src/AvoidanceStateMachine.cpp:24-54).

Parameters mirror AM/config/mpc_parameters.yaml.
"""
from dataclasses import dataclass, field

import numpy as np

# AM/config/mpc_parameters.yaml:7-52 in the weightsName order of ParameterManager.cpp:63-68
DEFAULT_WEIGHTS = [50.0, 50.0, 100.0, 100.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0,    # goal_*
                   0.0, 10.0, 50.0, 100.0, 0.0, 1.0, 1.0, 0.0, 1.0, 1.0,      # path_*
                   0.3, 0.3, 0.5, 1.0,                                        # u_*
                   1.2]                                                       # collide_lambda
DEFAULT_TAU = [6.09837416, 6.21675029, 15.79816293, 0.0]
DEFAULT_GAIN = [0.999999, 0.999999, 0.999999, 1.0]


@dataclass
class MpcParams:
    """The yaml keys that define the problem (mpc_parameters.yaml:1-57)."""
    T: float = 0.66
    dt: float = 0.033
    max_iter: int = 3                      # mpc_max_iter
    K: int = 8                             # nearest_point_num
    weights: list = field(default_factory=lambda: list(DEFAULT_WEIGHTS))
    tau: list = field(default_factory=lambda: list(DEFAULT_TAU))
    gain: list = field(default_factory=lambda: list(DEFAULT_GAIN))
    speed: float = 10.0
    radius: float = 0.5                    # drone_radius
    a_min_z: float = 5.0
    a_max_z: float = 15.0
    a_max_xy: float = 10.0
    a_max_yaw_dot: float = 10.0
    height: float = 1.5
    safety_distance: float = 0.2
    decay: float = 0.015
    drag: tuple = (0.0, 0.0, 0.0)         # use_drag_coefficient read as k v (0.033 in the generator); 0 = the yaml's default

    @property
    def N(self):
        return int(self.T / self.dt)       # HighLvlMpc.cpp:9, mpc_obstacle_casadi.py:36


CONFIGS = {
    # BASELINE.json configs
    "C1": dict(n=5000, T=0.33, K=3),
    "C2": dict(n=50000, T=0.66, K=8),
    "C5": dict(n=200000, T=1.0, K=8),
    # the reference's own default problem (mpc_parameters.yaml:1-2,5,59-63): N = 30, K = 3, 640 x 480 / 10 -> <= 3072-point frames
    "YAML": dict(n=3072, T=1.0, K=3),
}


def make_cloud(n, seed):
    """float32 [n,3] obstacle cloud + float32 [n//10,3] edge cloud for one scene."""
    rng = np.random.default_rng(seed)
    ncyl = int(rng.integers(20, 61))
    cx = rng.uniform(2.0, 30.0, ncyl)
    cy = rng.uniform(-8.0, 8.0, ncyl)
    cr = rng.uniform(0.1, 0.5, ncyl)
    n1 = int(0.7 * n)
    ci = rng.integers(0, ncyl, n1)
    th = rng.uniform(0.0, 2.0 * np.pi, n1)
    on_cyl = np.stack([cx[ci] + cr[ci] * np.cos(th), cy[ci] + cr[ci] * np.sin(th),
                       rng.uniform(0.0, 4.0, n1)], axis=1)
    n2 = n - n1
    clutter = np.stack([rng.uniform(0.0, 30.0, n2), rng.uniform(-8.0, 8.0, n2),
                        rng.uniform(0.0, 4.0, n2)], axis=1)
    cloud = np.concatenate([on_cyl, clutter]).astype(np.float32)
    cloud = cloud[rng.permutation(n)]
    # silhouette points: the two tangent lines of each cylinder seen from the origin
    ne = n // 10
    ei = rng.integers(0, ncyl, ne)
    side = rng.integers(0, 2, ne) * 2 - 1
    dist = np.sqrt(cx[ei] ** 2 + cy[ei] ** 2)
    base = np.arctan2(cy[ei], cx[ei])
    off = np.arcsin(np.clip(cr[ei] / dist, -1.0, 1.0))
    ang = base + side * (np.pi / 2.0 + off)
    ex = cx[ei] + cr[ei] * np.cos(ang)
    ey = cy[ei] + cr[ei] * np.sin(ang)
    edge = np.stack([ex, ey, rng.uniform(0.0, 4.0, ne)], axis=1).astype(np.float32)
    return cloud, edge


def make_odom(seed, prm: MpcParams):
    """(pos, vel, acc, yaw) of one scene."""
    rng = np.random.default_rng(seed + 7919)
    y0 = float(rng.uniform(-1.0, 1.0))
    pos = np.array([0.0, y0, prm.height])
    vel = np.array([prm.speed, 0.0, 0.0])
    acc = np.zeros(3)
    return pos, vel, acc, 0.0


def make_ref_path(pos, prm: MpcParams):
    """[N,10] straight reference path along +x (steady state of GetInitPath 'forward')."""
    N = prm.N
    ref = np.zeros((N, 10))
    for i in range(N):
        ref[i] = [pos[0] + prm.speed * prm.dt * (i + 1), pos[1], prm.height, 0.0,
                  prm.speed, 0.0, 0.0, 0.0, 0.0, 0.0]
    # last point as GetInitPath writes it (AvoidanceStateMachine.cpp:29-33,53)
    ref[N - 1] = [prm.speed * prm.T + pos[0], 0.0, prm.height, 0.0, prm.speed, 0.0, 0.0, 0.0, 0.0, 0.0]
    return ref


def make_scene(n, seed, prm: MpcParams):
    cloud, edge = make_cloud(n, seed)
    pos, vel, acc, yaw = make_odom(seed, prm)
    return dict(cloud=cloud, edge=edge, pos=pos, vel=vel, acc=acc, yaw=yaw,
                ref_path=make_ref_path(pos, prm))


def make_clouds_torch(n, S, seed, device):
    """Batch of S synthetic frames generated ON the device (bench.py: every in-flight step gets its own frames, so the
    working set is far beyond the 256 MiB Infinity Cache and the builds stream from HBM).  Same scene model as make_cloud
    (20-60 vertical cylinders carrying 70 % of the points, 30 % clutter, randomly permuted; n/10 silhouette points as the
    edge cloud); different random stream, so the values differ from the numpy generator's.
    -> (cloud float32 [S, n, 3], edge float32 [S, n // 10, 3])"""
    import math
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    rnd = lambda *shape: torch.rand(shape, generator=g, device=device, dtype=torch.float32)
    ncyl = torch.randint(20, 61, (S, 1), generator=g, device=device)
    cx = 2.0 + 28.0 * rnd(S, 60); cy = -8.0 + 16.0 * rnd(S, 60); cr = 0.1 + 0.4 * rnd(S, 60)
    n1 = int(0.7 * n)
    ci = (rnd(S, n1) * ncyl).long().clamp_(max=59)
    th = 2.0 * math.pi * rnd(S, n1)
    gx, gy, gr = cx.gather(1, ci), cy.gather(1, ci), cr.gather(1, ci)
    on_cyl = torch.stack([gx + gr * torch.cos(th), gy + gr * torch.sin(th), 4.0 * rnd(S, n1)], dim=2)
    n2 = n - n1
    clutter = torch.stack([30.0 * rnd(S, n2), -8.0 + 16.0 * rnd(S, n2), 4.0 * rnd(S, n2)], dim=2)
    cloud = torch.cat([on_cyl, clutter], dim=1)
    perm = torch.argsort(rnd(S, n), dim=1)
    cloud = cloud.gather(1, perm[:, :, None].expand(S, n, 3)).contiguous()
    ne = n // 10
    ei = (rnd(S, ne) * ncyl).long().clamp_(max=59)
    side = (rnd(S, ne) < 0.5).float() * 2.0 - 1.0
    ex0, ey0, er = cx.gather(1, ei), cy.gather(1, ei), cr.gather(1, ei)
    dist = torch.sqrt(ex0 * ex0 + ey0 * ey0)
    ang = torch.atan2(ey0, ex0) + side * (math.pi / 2.0 + torch.asin((er / dist).clamp(-1.0, 1.0)))
    edge = torch.stack([ex0 + er * torch.cos(ang), ey0 + er * torch.sin(ang), 4.0 * rnd(S, ne)], dim=2).contiguous()
    return cloud, edge
