"""SURVEY.md section 8 row f4 end to end on the GPU: message-shaped Image / Odometry / Imu -> ros_glue::OnOdometry / OnImu /
OnDepth -> FrameKDMap::AddVertex(Twb, depth) (depth -> obstacle cloud + edge cloud -> both KD indices) ->
AvoidanceTaskStep::GetInitPath + Step -> ros_glue::FillStepCmd, in ONE C++ process (tests/cpp/f4_e2e.cpp), against the oracle
chain depth_oracle -> kd_oracle -> step_oracle on the same messages.  Reference: AM/src/AvoidanceStateMachine.cpp:118-164,
322-355,369-397; AM/src/FrameKDMap.cpp:34-52,75-214."""
import os
import struct
import subprocess

import numpy as np
import pytest

from avoid_mpc_amd import flight, fsm, synth
from tests import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DPRM = dict(pixel2meter=1e-3, depth_min=0.1, depth_max=60.0, resize_scale=5.0, fx=160.0, fy=160.0, cx=160.0, cy=120.0,
            Tbc=flight.TBC_YAML)     # pixel2meter: 16UC1 frames are in millimetres, 32FC1 frames in metres (set per test)
ROWS, COLS = 240, 320
WALL = 0.4


def quat_rot(w, x, y, z):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def make_messages(prm, depth_type=0, wall_frame=2):
    """A short flight's worth of messages: odometry / IMU / depth per control period.  Frame `wall_frame` looks at a flat wall
    one step ahead (no depth edges -> empty edge cloud, obstacle points on the first reference point -> PlanWapionts fails)."""
    rng = np.random.default_rng(12)
    world = flight.FlightWorld(41, prm, 1000, cyl_per_m=2.0, x_first=3.0, length=40.0)
    msgs = []
    pos = np.array([0.0, 0.3, prm.height]); vel = np.array([prm.speed, 0.4, 0.0])
    for f in range(4):
        yaw = 0.05 * f
        q = np.array([np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)])
        t0 = 100.0 + 0.033 * f
        accb = quat_rot(*q).T @ (np.array([0.5, -0.2, 0.1]) + [0, 0, 9.81])     # body-frame specific force for a world acceleration
        m = dict(pos=pos.copy(), quat=q, vel=vel.copy(), t_odom=t0, accb=accb, t_imu=t0 + 0.004, t_depth=t0 + 0.006,
                 t_step=t0 + (0.006 if f != 1 else 0.011))
        # the pose the depth callback will compute (closed form below) is what the camera looks from
        Twb = np.eye(4); Twb[:3, :3] = quat_rot(*q); Twb[:3, 3] = pos
        if f == wall_frame:
            depth = np.full((ROWS, COLS), WALL, np.float32)
        else:
            cyl = (world.cx, world.cy, world.cr)
            depth = flight.render_depth(cyl, Twb, DPRM["Tbc"], ROWS, COLS, DPRM["fx"], DPRM["fy"], DPRM["cx"], DPRM["cy"])
        m["depth_m"] = depth
        m["img"] = np.clip(np.round(depth / DPRM["pixel2meter"]), 0, 65535).astype(np.uint16) if depth_type == 0 else depth
        msgs.append(m)
        pos = pos + vel * 0.033
    return msgs


def oracle_chain(msgs, prm, ref0):
    """What the node computes per period, on the CPU oracle."""
    Tbc = DPRM["Tbc"]
    mpc = _oracle.MpcOracle(prm.T, prm.dt, prm.K); mpc.configure(prm)
    ref = ref0.copy()
    o_pos, o_vel, o_acc, o_stamp, quat = np.zeros(3), np.zeros(3), np.zeros(3), 0.0, np.array([1.0, 0, 0, 0])
    Twc = np.eye(4)
    kd = ke = None
    out = []
    for m in msgs:
        o_pos, quat, o_vel, o_stamp = m["pos"].copy(), m["quat"], m["vel"].copy(), m["t_odom"]            # OdomCallback :118-134
        yaw = np.arctan2(2 * (quat[0] * quat[3] + quat[1] * quat[2]), 1 - 2 * (quat[2] ** 2 + quat[3] ** 2))
        dt = m["t_imu"] - o_stamp                                                                          # IMUCallback :136-152
        o_pos = o_pos + o_vel * dt + 0.5 * o_acc * dt * dt; o_vel = o_vel + o_acc * dt; o_stamp = m["t_imu"]
        o_acc = quat_rot(*quat) @ m["accb"] - np.array([0, 0, 9.81])
        dt = m["t_depth"] - o_stamp                                                                        # DepthCallback :153-164
        Twb = np.eye(4); Twb[:3, :3] = quat_rot(*quat); Twb[:3, 3] = o_pos + o_vel * dt + 0.5 * o_acc * dt * dt
        cloud, _ = _oracle.depth_oracle(m["img"], DPRM, Twb)                                               # AddVertex FrameKDMap.cpp:34-52
        if len(cloud):
            edge = _oracle.depth_edge_oracle(m["img"], DPRM, Twc)[0]
            kd, ke = _oracle.kd_oracle(cloud), _oracle.kd_oracle(edge)
            Twc = Twb @ Tbc
        r = dict(have=int(kd is not None), n_cloud=kd.size() if kd else 0, n_edge=ke.size() if kd else 0, Twb=Twb)
        if kd is not None:                                                                                 # Step, TASK :322-355
            fsm.get_init_path(ref, prm.speed, prm.T, o_pos[0], 500.0, prm.height)
            age = m["t_step"] - o_stamp
            sq = fsm.state_quads(o_pos, o_vel, o_acc, yaw, prm.decay, prm.max_iter, iter_time=prm.decay, age=age)
            s = _oracle.step_oracle(kd, ke, mpc, prm, sq, o_pos[0], ref)
            x = np.concatenate([o_pos, [yaw], o_vel, o_acc])
            r.update(flags=s["flags"], u=s["u"], cmd=flight.command(s["u"][None], s["flags"][None], x[None], prm)[0], ref=ref.copy())
        out.append(r)
    return out


def write_input(path, msgs, prm, ref0, depth_type, pad):
    with open(path, "wb") as f:
        f.write(struct.pack("7i", len(msgs), ROWS, COLS, depth_type, pad, prm.max_iter, prm.K))
        f.write(np.array([prm.T, prm.dt, prm.speed, prm.safety_distance, prm.decay, prm.height, DPRM["pixel2meter"], DPRM["depth_min"],
                          DPRM["depth_max"], DPRM["resize_scale"], DPRM["fx"], DPRM["fy"], DPRM["cx"], DPRM["cy"]]).tobytes())
        f.write(np.ascontiguousarray(DPRM["Tbc"], np.float64).tobytes())
        f.write(np.array(prm.weights, np.float64).tobytes()); f.write(np.array(prm.tau, np.float64).tobytes())
        f.write(np.array(prm.gain, np.float64).tobytes())
        f.write(np.array([prm.a_min_z, prm.a_max_z, prm.a_max_xy, prm.a_max_yaw_dot, prm.radius, 0.3, 0.3, 500.0]).tobytes())
        f.write(np.ascontiguousarray(ref0, np.float64).tobytes())
        for m in msgs:
            f.write(np.concatenate([m["pos"], m["quat"], m["vel"], [m["t_odom"]], m["accb"], [m["t_imu"], m["t_depth"], m["t_step"]]]).tobytes())
            raw = np.ascontiguousarray(m["img"]).view(np.uint8).reshape(ROWS, -1)
            f.write(np.concatenate([raw, np.full((ROWS, pad), 0xEE, np.uint8)], axis=1).tobytes())   # row padding: step > width * bpp


@pytest.mark.gpu
@pytest.mark.parametrize("depth_type,pad", [(0, 6), (1, 0)])
def test_f4_end_to_end(tmp_path, depth_type, pad):
    exe = str(tmp_path / "f4_e2e")
    libdir = os.path.join(ROOT, "avoid_mpc_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "f4_e2e.cpp"),
                           "-o", exe, "-L", libdir, "-lavoid_mpc_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    prm = synth.MpcParams(T=0.33, K=3)
    DPRM["pixel2meter"] = 1e-3 if depth_type == 0 else 1.0
    msgs = make_messages(prm, depth_type)
    ref0 = synth.make_ref_path(msgs[0]["pos"], prm)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_input(fin, msgs, prm, ref0, depth_type, pad)
    subprocess.check_call([exe, fin, fout])
    want = oracle_chain(msgs, prm, ref0)
    buf = open(fout, "rb").read()
    off = 0
    N = prm.N

    def take(dtype, cnt):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=cnt, offset=off)
        off += a.nbytes
        return a

    unsafe = 0
    for i, w in enumerate(want):
        have, n_cloud, n_edge = take(np.int32, 3); flags = take(np.int32, 4); u = take(np.float64, 4)
        mode = int(take(np.int32, 1)[0]); acc = take(np.float64, 4); ref = take(np.float64, 10 * N).reshape(N, 10)
        Twb = take(np.float64, 16).reshape(4, 4)
        assert np.abs(Twb - w["Twb"]).max() <= 1e-12
        assert (have, n_cloud, n_edge) == (w["have"], w["n_cloud"], w["n_edge"]), i    # depth -> both clouds, point for point
        assert n_cloud > 500
        assert np.array_equal(flags, w["flags"]), (i, flags, w["flags"])
        assert np.abs(u - w["u"]).max() <= 1e-6 and np.abs(ref - w["ref"]).max() <= 1e-6
        assert mode == 1 and acc[3] == 0.0                                               # ACCELERATION_MODE, yaw = 0  (:372-376)
        assert np.abs(acc[:3] - w["cmd"]).max() <= (1e-6 if flags[0] else 1e-12)
        unsafe += int(flags[0] == 0)
    assert off == len(buf)
    assert 0 < unsafe < len(want)          # both branches of the TASK tail ran: PubCmd and PubSlowDownCmd
