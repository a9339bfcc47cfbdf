"""Closed-loop parity: the same flights on the GPU (amk_pipeline_submit, keep_warm_start = 1) and on the CPU oracle.

north_star: "identical obstacle-avoidance trajectories".  The reference's regime is a 30 Hz loop (GetInitPath shift,
fresh frame, re-read state, TASK branch from mNlpW0; AM/src/AvoidanceStateMachine.cpp:24-54,183-203,322-355,
AM/src/HighLvlMpc.cpp:129).  Per step GPU and oracle agree to ~1e-12 except where a rounding-level tie flips a branch
(0.2 % of the cold-started scene-steps, tests/test_step_gpu.py); this test measures what that does over a flight:
positions must agree to 1e-6 m for as long as the flags agree, and the flights that separate are counted and reported.
"""
import json
import os

import numpy as np
import pytest

from tests import _flight

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(name, obj):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        json.dump(obj, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))


def test_flights_c2_gpu_equals_oracle():
    """64 flights x 100 periods at BASELINE configs[1] size (50 k-point frames, N = 20, K = 8)."""
    F, P = 64, 100
    seeds = list(range(5000, 5000 + F))
    kw = dict(cyl_per_m=1.5)
    g = _flight.gpu_flights(seeds, "C2", P, world_kw=kw)
    o = _flight.oracle_flights(seeds, "C2", P, world_kw=kw)
    prm, _ = _flight.make_prm("C2")
    cmp = _flight.compare(g, o, pos_tol=1e-6)
    sg, so = _flight.flight_stats(g, prm), _flight.flight_stats(o, prm)
    rep = dict(config="C2: 50k-point frames, N=20, K=8, 64 flights x 100 periods, 1.5 cylinders/m", compare=cmp, gpu=sg, oracle=so)
    _report("flight_c2_gpu_vs_oracle.json", rep)
    print("\nflights GPU vs oracle:", {k: v for k, v in cmp.items() if k not in ("separation_period", "dpos_final")})
    print("separated at period:", cmp["separation_period"][cmp["separation_period"] >= 0], "final |dpos| of those:",
          np.round(cmp["dpos_final"][cmp["separation_period"] >= 0], 4))
    print("GPU   ", sg)
    print("oracle", so)
    assert cmp["dpos_max_while_together"] <= 1e-6
    # a separated flight took another branch at a rounding-level tie (another iteration count, rarely another local minimum):
    # allowed on <= 3 % of the flights (the census over 1024 flights x 150 periods observed 1.2 %, profiles/r04_flight_census_1024.json);
    # the closed loop must bring them back: the census saw 11 of 12 separated flights end within 3e-5 m of the oracle's
    assert cmp["separated"] <= max(1, (3 * F) // 100), cmp["separation_period"]
    assert sg["capped_periods"] == 0 and so["capped_periods"] == 0
    # the same flying: clearance statistics agree (a separated flight may differ in the last digits of its minimum)
    assert abs(sg["min_clearance_median"] - so["min_clearance_median"]) < 0.02
    assert abs(sg["collided"] - so["collided"]) <= cmp["separated"] and abs(sg["hit_surface"] - so["hit_surface"]) <= cmp["separated"]


def test_flights_c1_and_batches():
    """C1-sized flights split over two pipeline slots (two batches in flight per period) = the same flights in one batch = the oracle."""
    seeds = list(range(700, 716))
    kw = dict(cyl_per_m=2.0, x_first=3.0)
    g1 = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw)
    g2 = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw, batch=8)
    assert np.array_equal(g1["x"], g2["x"]) and np.array_equal(g1["flags"], g2["flags"])   # scenes are independent: bit-identical
    o = _flight.oracle_flights(seeds, "C1", 40, world_kw=kw)
    cmp = _flight.compare(g1, o)
    print("\nC1 flights:", {k: v for k, v in cmp.items() if k not in ("separation_period", "dpos_final")})
    assert cmp["dpos_max_while_together"] <= 1e-6 and cmp["separated"] <= 1


def test_task_mode_equals_host_mode():
    """The pipeline's TASK mode (GetInitPath on the slot's own mRefPath, GetCurStateQuad per pass, PubCmd / PubSlowDownCmd on the
    device: AvoidanceStateMachine.cpp:24-54,183-203,345-350,379-397) flies the flights of the host-driven loop, bit for bit -- also
    with the batches of a period sharing launches (gang 2) and with a slow-down period forced by an empty edge cloud."""
    seeds = list(range(700, 716))
    kw = dict(cyl_per_m=2.0, x_first=3.0)
    h = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw, batch=4)
    t1 = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw, batch=4, mode="task")
    t2 = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw, batch=4, mode="task", gang=2)
    for t in (t1, t2):
        assert np.array_equal(h["x"], t["x"]) and np.array_equal(h["flags"], t["flags"]) and np.array_equal(h["cmd"], t["cmd"])


def test_global_goal_task_of_get_init_path():
    """GetInitPath's other task (AvoidanceStateMachine.cpp:34-45, mStrTask == "global_goal"): the last point of the path walks
    towards mStateGlobalGoal by at most speed * dt per period, its z goes into every shifted point.  The device prologue
    (csrc/pipeline.hip) against the host twin (avoid_mpc_amd/fsm.py), bit for bit over whole flights, gang 1 and 2, goals 20 m
    ahead / off-axis / the constructor's default {0, 0, height} (d_global_goal = NULL: the goal is BEHIND the start, the path
    folds back); and against the CPU oracle flown with the same task."""
    seeds = list(range(760, 768))
    kw = dict(cyl_per_m=1.0, x_first=3.0)
    prm, _ = _flight.make_prm("C1")
    rng = np.random.default_rng(4)
    goals = np.stack([[20.0, 0.0, prm.height]] * 4 + [[18.0 + rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), prm.height + rng.uniform(-0.3, 0.3)]
                                                      for _ in range(4)])
    for tk in (dict(task="global_goal", global_goal=goals), dict(task="global_goal", global_goal=None)):
        h = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw, batch=4, task_kw=tk)
        assert np.abs(h["x"][:, -1, 0:3] - h["x"][:, 0, 0:3]).max() > 0.5
        for gang in (1, 2):
            t = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw, batch=4, mode="task", gang=gang, task_kw=tk)
            assert np.array_equal(h["x"], t["x"]) and np.array_equal(h["flags"], t["flags"]) and np.array_equal(h["cmd"], t["cmd"])
        o = _flight.oracle_flights(seeds, "C1", 40, world_kw=kw, task_kw=tk)
        cmp = _flight.compare(h, o)
        print("\nglobal_goal flights:", {k: v for k, v in cmp.items() if k not in ("separation_period", "dpos_final")},
              "x final", h["x"][:, -1, 0].round(2))
        assert cmp["dpos_max_while_together"] <= 1e-6 and cmp["separated"] <= 1
    fwd = _flight.gpu_flights(seeds, "C1", 40, world_kw=kw, batch=4)
    assert not np.array_equal(fwd["x"], h["x"])                 # the task does change the flight
    # (the vehicle does NOT stop at the goal in this harness: the path's last point carries vx = mSpeed (:53) and GetRefStates
    # pushes the target on by up to speed * T (:251-254) -- the reference's behaviour, flown identically by all three drivers)


def test_flights_from_the_raw_depth_image():
    """Pipeline frames that start where FrameKDMap::AddVertex starts (FrameKDMap.cpp:34-52): rendered 16UC1 depth images ->
    ProcessDepth + BuildEdgeCloud (through the slot's own stale Twc, :209) -> both index builds -> TASK step, against the oracle
    chain depth_oracle -> kd_oracle -> step_oracle, over whole flights; one launch per batch and two batches per launch."""
    seeds = list(range(900, 908))
    kw = dict(cyl_per_m=2.0, x_first=3.0, length=40.0)
    o = _flight.oracle_depth_flights(seeds, "C1", 30, world_kw=kw)
    assert o["n_cloud"].min() > 300 and o["n_edge"].max() > 50 and o["n_cloud"].max() <= 3072
    prm, _ = _flight.make_prm("C1")
    for gang, batch in ((1, 8), (2, 4)):
        g = _flight.gpu_depth_flights(seeds, "C1", 30, world_kw=kw, gang=gang, batch=batch)
        cmp = _flight.compare(g, o, pos_tol=1e-9)
        print("\ndepth-image flights, gang %d:" % gang, {k: v for k, v in cmp.items() if k not in ("separation_period", "dpos_final")},
              _flight.flight_stats(g, prm))
        assert cmp["dpos_max_while_together"] <= 1e-9 and cmp["separated"] <= 1


def test_cpp_fleet_host_on_task_mode(tmp_path):
    """tests/cpp/flight_driver.cpp: a C++ host that only knows include/avoid_mpc_amd.h flies the flights of the Python driver --
    frames and odometry in, commands out, GetInitPath / warm start / re-plan loop inside the slot -- bit for bit."""
    import struct
    import subprocess
    from avoid_mpc_amd import flight
    exe = str(tmp_path / "flight_driver")
    libdir = os.path.join(ROOT, "avoid_mpc_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "flight_driver.cpp"), "-o", exe, "-L", libdir, "-lavoid_mpc_amd",
                           f"-Wl,-rpath,{libdir}"])
    seeds = list(range(700, 708))
    kw = dict(cyl_per_m=2.0, x_first=3.0)
    P = 30
    prm, n = _flight.make_prm("C1")
    want = _flight.gpu_flights(seeds, "C1", P, world_kw=kw, mode="task")
    worlds = [flight.FlightWorld(s, prm, n, **kw) for s in seeds]
    st = [flight.initial_state(s, prm) for s in seeds]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("6i", len(seeds), P, n, n // 10, prm.K, prm.max_iter))
        f.write(np.array([prm.T, prm.dt, prm.speed, prm.safety_distance, prm.decay, prm.height, 500.0, 0.3, 0.3]).tobytes())
        f.write(np.array(prm.weights, np.float64).tobytes()); f.write(np.array(prm.tau, np.float64).tobytes())
        f.write(np.array(prm.gain, np.float64).tobytes())
        f.write(np.array([prm.a_min_z, prm.a_max_z, prm.a_max_xy, prm.a_max_yaw_dot, prm.radius]).tobytes())
        f.write(np.stack([a for a, _ in st]).tobytes()); f.write(np.stack([b for _, b in st]).tobytes())
        for t in range(P):
            fr = [w.frame(t) for w in worlds]
            f.write(np.stack([c for c, _ in fr]).tobytes()); f.write(np.stack([e for _, e in fr]).tobytes())
    subprocess.check_call([exe, fin, fout])
    buf = open(fout, "rb").read()
    S, off = len(seeds), 0
    for t in range(P):
        x = np.frombuffer(buf, np.float64, S * 10, off).reshape(S, 10); off += x.nbytes
        cmd = np.frombuffer(buf, np.float64, S * 3, off).reshape(S, 3); off += cmd.nbytes
        fl = np.frombuffer(buf, np.int32, S * 4, off).reshape(S, 4); off += fl.nbytes
        assert np.array_equal(fl, want["flags"][:, t]) and np.array_equal(cmd, want["cmd"][:, t]), t
        assert np.array_equal(x, want["x"][:, t + 1]), t
    assert off == len(buf)


def test_depth_frames_that_yield_no_point():
    """FrameKDMap::AddVertex returns before both InitializeNew calls (and before Twc is updated) when ProcessDepth yields no point
    (FrameKDMap.cpp:39-41): a robot whose FIRST frame is empty flies on an empty map (no neighbours: obstacles padded with 1e4,
    every pass re-plans, AvoidanceStateMachine.cpp:223-231), a robot whose LATER frame is empty keeps the previous frame's
    indices and Twc.  Three periods, robot 1 blind in period 0, robot 2 blind in period 1, against the oracle chain."""
    import torch
    from avoid_mpc_amd import flight
    from avoid_mpc_amd.host import Pipeline, depth_params
    from tests import _oracle
    c = _flight.DEPTH_CAM
    prm, _ = _flight.make_prm("C1")
    S, P = 3, 3
    kw = dict(cyl_per_m=2.0, x_first=3.0, length=40.0)
    worlds = [flight.FlightWorld(950 + s, prm, 1000, **kw) for s in range(S)]
    st = [flight.initial_state(950 + s, prm) for s in range(S)]
    blind = {(0, 1), (1, 2)}            # (period, robot) pairs that see nothing
    cap = int(c["cols"] / c["resize_scale"]) * int(c["rows"] / c["resize_scale"])
    dp = depth_params(c["pixel2meter"], c["depth_min"], c["depth_max"], c["resize_scale"], c["fx"], c["fy"], c["cx"], c["cy"], c["Tbc"])

    def frame(t, s, x):
        img, Twb = _flight._depth_frame(worlds[s], x)
        return (np.zeros_like(img) if (t, s) in blind else img), Twb

    # ---- oracle chain
    xo = np.stack([a for a, _ in st]); refo = np.stack([b for _, b in st])
    mp = [_oracle.MpcOracle(prm.T, prm.dt, prm.K) for _ in range(S)]
    for m in mp:
        m.configure(prm)
    empty = np.zeros((0, 3), np.float32)
    kd = [_oracle.kd_oracle(empty) for _ in range(S)]; ke = [_oracle.kd_oracle(empty) for _ in range(S)]; Twc = [np.eye(4) for _ in range(S)]
    want = []
    for t in range(P):
        row = []
        for s in range(S):
            img, Twb = frame(t, s, xo[s])
            cloud, _ = _oracle.depth_oracle(img, c, Twb)
            if len(cloud):
                kd[s], ke[s] = _oracle.kd_oracle(cloud), _oracle.kd_oracle(_oracle.depth_edge_oracle(img, c, Twc[s])[0])
                Twc[s] = Twb @ c["Tbc"]
            sq, px = flight.period_inputs(xo[s][None], refo[s][None], prm)
            r = _oracle.step_oracle(kd[s], ke[s], mp[s], prm, sq[0], px[0], refo[s])
            a = flight.command(r["u"][None], r["flags"][None], xo[s][None], prm)
            xo[s] = flight.apply_command(xo[s][None], a, prm)[0]
            row.append((r["flags"].copy(), r["u"].copy(), a[0].copy(), kd[s].size()))
        want.append(row)
    assert want[0][1][3] == 0 and want[0][1][0][1] == prm.max_iter        # the blind first frame: empty map, every pass re-plans
    assert want[1][2][3] == want[0][2][3] > 300                           # the blind later frame: the previous index stays
    # ---- the pipeline
    dev = torch.device("cuda", torch.cuda.current_device())
    pl = Pipeline(1, S, cap, cap, prm, depth=dp)
    x = np.stack([a for a, _ in st]); ref0 = np.stack([b for _, b in st])
    for t in range(P):
        fr = [frame(t, s, x[s]) for s in range(S)]
        depth = torch.from_numpy(np.stack([d for d, _ in fr]).view(np.int16)).to(dev); Twb = torch.from_numpy(np.stack([T for _, T in fr])).to(dev)
        odom = torch.from_numpy(x).to(dev); cmd = torch.empty((S, 3), dtype=torch.float64, device=dev)
        tk = pl.submit(None, None, ref_path_init=torch.from_numpy(ref0).to(dev) if t == 0 else None, odom=odom, cmd_out=cmd,
                       keep_warm_start=t > 0, depth=depth, Twb=Twb)
        pl.wait(tk)
        o = pl.outputs(tk)
        a = cmd.cpu().numpy()
        sizes = pl.kd(0, 0).sizes()
        for s in range(S):
            assert np.array_equal(o["flags"][s], want[t][s][0]), (t, s, o["flags"][s], want[t][s][0])
            assert np.abs(o["u"][s] - want[t][s][1]).max() <= 1e-6 and np.abs(a[s] - want[t][s][2]).max() <= 1e-6
            assert sizes[s] == want[t][s][3], (t, s, sizes[s], want[t][s][3])
        x = flight.apply_command(x, a, prm)
    pl.close()
