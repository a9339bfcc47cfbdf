"""GPU: the keyframe map (amk_kfmap, csrc/kfmap.hip) against FrameKDMap's keyframe list restated on the CPU oracle's trees
(tests/_kfmap.py: AM/src/FrameKDMap.cpp:29-74,233-252,428-488) -- the regime the reference flies in by default
(max_frame_count = 100, only_trust_vel = false).

  test_map_follows_the_reference_list   the map alone, fed scripted depth frames: keyframe counts, query-vector lengths, sweep
                                        outliers, the size of every query frame and a control step over the map, every period
  test_flights_with_the_keyframe_map    whole closed-loop flights from rendered depth images through amk_pipeline (TASK mode,
                                        keyframes in the slot, gang 1 and 2), max_frame_count 3 and 10
"""
import numpy as np
import pytest

from tests import _flight, _kfmap, _oracle
from avoid_mpc_amd import flight

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _scripted_frames(seed, prm, periods, world_kw):
    """Depth frames along a scripted trajectory (10 m/s ahead, a slow weave): [(cloud, edge-maker, Twb)]"""
    world = flight.FlightWorld(seed, prm, 1000, **world_kw)
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28)
    xs = []
    for t in range(periods):
        x = np.zeros(10)
        x[0] = 0.5 + prm.speed * prm.dt * t
        x[1] = 0.8 * np.sin(0.15 * t + ph)
        x[2] = prm.height + 0.1 * np.sin(0.11 * t)
        xs.append(x)
    return world, xs


@pytest.mark.parametrize("max_frames,th_count,wide,target,order", [(3, 10, 0, 1, 1), (6, 10, 1, 1, 1), (12, 40, 0, 0, 1), (100, 10, 0, 1, 1),
                                                                   (6, 10, 0, 1, 0), (6, 1300, 0, 1, 1)])
def test_map_follows_the_reference_list(max_frames, th_count, wide, target, order, torch_cuda):
    """(wide = 1: the merge of maps with more than 1024 (frame, neighbour) candidates -- candidates re-read every round -- forced
    on a small map; max_frames = 100: the yaml's own max_frame_count, mpc_parameters.yaml:73; target = 0: the sweep against the
    current frame's own index instead of its fine hashed grid, the default; order = 0: the hashed sweep with the keyframe's points always in
    record order -- round 6's default, 1, takes them in the order of last sweep's grid whenever the keyframe IS last sweep's frame; th_count 1300:
    only some sweeps rebuild, so the newest keyframe stays for several periods and the rows alternate between the two orders)"""
    torch = torch_cuda
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import KfMap, MpcBatch
    capi.load().amk__frames_force_wide(int(wide))
    capi.load().amk__sweep_set_target(int(target))
    capi.load().amk__sweep_set_order(int(order))
    prm, _ = _flight.make_prm("C1")
    c = _flight.DEPTH_CAM
    S, P = 6, 36
    cap = int(c["cols"] / c["resize_scale"]) * int(c["rows"] / c["resize_scale"])
    kw = dict(cyl_per_m=2.0, x_first=3.0, length=40.0)
    scripts = [_scripted_frames(1200 + s, prm, P, kw) for s in range(S)]
    th_dist = 0.1
    gmap = KfMap(S, cap, cap, max_frames, th_dist, th_count, c["depth_min"], c["Tbc"])
    omaps = [_kfmap.MapOracle(max_frames, th_dist, th_count, c["depth_min"], c["Tbc"]) for _ in range(S)]
    gmpc = MpcBatch(prm.T, prm.dt, prm.K, S); gmpc.configure(prm)
    ompc = [_oracle.MpcOracle(prm.T, prm.dt, prm.K) for _ in range(S)]
    for m in ompc:
        m.configure(prm)
    cam = _flight.depth_camera()
    gcam = capi.FrameCamera(*cam[:5], int(cam[5]), int(cam[6]))
    dev = torch.device("cuda")
    seen_pop = seen_keep = seen_multi = False
    sweeps_rebuilt = sweeps_kept = 0
    for t in range(P):
        clouds = np.zeros((S, cap, 3), np.float32); edges = np.zeros((S, cap, 3), np.float32)
        cn = np.zeros(S, np.int32); en = np.zeros(S, np.int32); Twc = np.zeros((S, 4, 4))
        refs = np.zeros((S, prm.N, 10)); sqs = np.zeros((S, prm.max_iter, 10)); px = np.zeros(S)
        for s, (world, xs) in enumerate(scripts):
            x = xs[t]
            img, Twb = _flight._depth_frame(world, x)
            if t % 11 == 7 and s == 2:
                img[:] = 0                                       # a frame without a valid pixel: AddVertex returns early (:39-41)
            cloud, _ = _oracle.depth_oracle(img, c, Twb)
            if len(cloud):
                edge = _oracle.depth_edge_oracle(img, c, omaps[s].Twc)[0]
                clouds[s, :len(cloud)] = cloud; cn[s] = len(cloud); edges[s, :len(edge)] = edge; en[s] = len(edge)
                Twc[s] = Twb @ c["Tbc"]
                omaps[s].add_vertex(cloud, edge, Twc[s], stamp=t)
            omaps[s].update()
            _, ref0 = flight.initial_state(1200 + s, prm)
            ref0[:, 0] += x[0]; ref0[:, 1] = x[1]
            refs[s] = ref0
            sq, p0 = flight.period_inputs(x[None], refs[s][None], prm, shift=False)
            sqs[s], px[s] = sq[0], p0[0]
        gmap.add_vertex(torch.from_numpy(clouds).to(dev), torch.from_numpy(edges).to(dev), torch.from_numpy(Twc).to(dev),
                        counts=torch.from_numpy(cn).to(dev), edge_counts=torch.from_numpy(en).to(dev))
        gmap.update()
        st = gmap.state()
        for s in range(S):
            nk, sizes = omaps[s].summary()
            assert st["n_keyframes"][s] == nk, (t, s, st["n_keyframes"][s], nk)
            assert st["n_query_frames"][s] == len(sizes), (t, s)
            assert list(st["frame_sizes"][s][:len(sizes)]) == sizes, (t, s, st["frame_sizes"][s], sizes)
            assert (st["frame_sizes"][s][len(sizes):] == -1).all()
            assert st["last_outliers"][s] == max(omaps[s].last_outliers, 0), (t, s)
            if omaps[s].last_outliers > 0:
                sweeps_rebuilt += int(omaps[s].last_outliers >= th_count); sweeps_kept += int(omaps[s].last_outliers < th_count)
            seen_multi = seen_multi or len(sizes) >= 3
        if t > 0:
            seen_pop = seen_pop or (st["n_keyframes"] < prev_nk).any() or (st["n_keyframes"] == prev_nk).any()
        prev_nk = st["n_keyframes"].copy()
        # the control step over the map, both sides from the same path
        dref = torch.from_numpy(refs.copy()).to(dev)
        out = gmap.step(gmpc, prm, torch.from_numpy(sqs).to(dev), torch.from_numpy(px).to(dev), dref, cam=gcam)
        torch.cuda.synchronize()
        gu, gf, gr = out["u"].cpu().numpy(), out["flags"].cpu().numpy(), dref.cpu().numpy()
        for s in range(S):
            r = omaps[s].step(ompc[s], prm, sqs[s], px[s], refs[s], cam)
            assert np.array_equal(gf[s][:3], r["flags"][:3]), (t, s, gf[s], r["flags"])
            if np.array_equal(gf[s], r["flags"]):
                assert np.abs(gu[s] - r["u"]).max() <= 1e-6 and np.abs(gr[s] - refs[s]).max() <= 1e-6, (t, s)
    assert seen_multi, "the scripted flight never had three query frames: the test did not exercise the merge"
    if th_count >= 100:   # the mixed case: both kinds of sweep happened (a keyframe that stays is swept in record order, a fresh one in grid order)
        assert sweeps_rebuilt > 0 and sweeps_kept > 0, (sweeps_rebuilt, sweeps_kept)
    print(f"sweeps that rebuilt their keyframe {sweeps_rebuilt}, that kept it {sweeps_kept}")
    nk_final = [len(m.kfs) for m in omaps]
    print(f"max_frame_count {max_frames}: keyframes at the end {nk_final}, query frames {[len(m.frames()) for m in omaps]}")
    gmap.close()
    capi.load().amk__frames_force_wide(0)
    capi.load().amk__sweep_set_target(1)
    capi.load().amk__sweep_set_order(1)


def test_argument_errors(torch_cuda):
    import ctypes as C
    from avoid_mpc_amd import capi
    lib = capi.load()
    h = C.c_void_p()
    Tbc = (C.c_double * 16)(*np.eye(4).reshape(-1))
    ok = capi.KfmapParams(3, 10, 0.1, 0.1, Tbc)
    assert lib.amk_kfmap_create(0, 100, 100, C.byref(ok), C.byref(h)) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_kfmap_create(4, 100, 100, C.byref(capi.KfmapParams(0, 10, 0.1, 0.1, Tbc)), C.byref(h)) == capi.AMK_ERR_UNSUPPORTED
    assert lib.amk_kfmap_create(4, 100, 100, C.byref(capi.KfmapParams(101, 10, 0.1, 0.1, Tbc)), C.byref(h)) == capi.AMK_ERR_UNSUPPORTED
    assert lib.amk_kfmap_create(4, 100, 100, C.byref(capi.KfmapParams(3, 0, 0.1, 0.1, Tbc)), C.byref(h)) == capi.AMK_ERR_INVALID_ARG
    # a map whose pools cannot fit is refused up front, with the figure: 4096 robots x 102 slots x 200 k points ~ 4 TB
    need = C.c_longlong()
    assert lib.amk_kfmap_pool_bytes(4096, 200000, 20000, 100, C.byref(need)) == 0 and need.value > 1e12
    assert lib.amk_kfmap_create(4096, 200000, 20000, C.byref(capi.KfmapParams(100, 10, 0.1, 0.1, Tbc)), C.byref(h)) == capi.AMK_ERR_UNSUPPORTED and not h.value
    assert lib.amk_kfmap_create(4, 100, 100, C.byref(ok), C.byref(h)) == 0 and h.value
    assert lib.amk_kfmap_frames(h) == 4 and lib.amk_kfmap_scenes(h) == 4
    assert lib.amk_kfmap_add_vertex(h, 2, 3, None, None, None, None, 3, None, None) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_kfmap_update(None, None) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_kfmap_destroy(h) == 0


@pytest.mark.parametrize("max_frames,gang,batch", [(3, 1, 16), (10, 2, 8)])
def test_flights_with_the_keyframe_map(max_frames, gang, batch):
    """>= 32 flights x 100 periods from rendered depth images with FrameKDMap's keyframe list in the loop: the pipeline's slot map
    (AddVertex -> KeyframeThreadWorker's body -> the step over mVecQueryVector, all on the device) against the oracle's list.
    Positions agree to 1e-6 m while the flags agree; the keyframe counts, query-vector lengths, sweep outliers and map sizes are
    EQUAL every period of every flight that has not separated."""
    seeds = list(range(2000, 2032))
    kw = dict(cyl_per_m=1.5, x_first=3.0, length=60.0)
    kf = dict(max_frame_count=max_frames, th_dist=0.1, th_count=10)
    P = 100
    o = _flight.oracle_depth_flights(seeds, "C1", P, world_kw=kw, keyframes=kf)
    g = _flight.gpu_depth_flights(seeds, "C1", P, world_kw=kw, gang=gang, batch=batch, keyframes=kf)
    cmp = _flight.compare(g, o, pos_tol=1e-6)
    sep = cmp["separation_period"]
    prm, _ = _flight.make_prm("C1")
    print(f"\\nkeyframe flights, max_frame_count {max_frames}, gang {gang}:",
          {k: v for k, v in cmp.items() if k not in ("separation_period", "dpos_final")}, _flight.flight_stats(g, prm),
          "keyframes per period (mean / max):", float(o["n_keyframes"].mean()), int(o["n_keyframes"].max()),
          "query frames (mean / max):", float(o["n_query_frames"].mean()), int(o["n_query_frames"].max()),
          "map points (mean):", float(o["map_points"].mean()))
    assert cmp["dpos_max_while_together"] <= 1e-6 and cmp["separated"] <= max(1, len(seeds) // 32)
    assert o["n_query_frames"].max() >= min(max_frames, 3), "the flights never had a multi-frame map"
    for f in range(len(seeds)):
        upto = P if sep[f] < 0 else sep[f]
        for key in ("n_keyframes", "n_query_frames", "outliers", "map_points", "n_cloud"):
            assert np.array_equal(g[key][f, :upto], o[key][f, :upto]), (f, key, np.nonzero(g[key][f, :upto] != o[key][f, :upto])[0][:5])


def test_a_flight_with_the_reference_yaml_configuration():
    """The reference's own default configuration end to end (mpc_parameters.yaml): 640 x 480 depth images down-scaled by 10
    (<= 3072-point frames), N = 30 (T = 1.0, dt = 0.033), nearest_point_num = 3, keyframe_th_dist 0.1, keyframe_th_count 10,
    max_frame_count = 100 -- the pool holds 102 index slots per robot and the step runs over up to 101 query frames.  4 flights x
    45 periods through the pipeline against the oracle's map."""
    seeds = list(range(2400, 2404))
    kw = dict(cyl_per_m=0.6, x_first=3.0, length=60.0)
    kf = dict(max_frame_count=100, th_dist=0.1, th_count=10)
    P = 45
    o = _flight.oracle_depth_flights(seeds, "YAML", P, world_kw=kw, keyframes=kf, cam=_flight.YAML_CAM, workers=1)
    g = _flight.gpu_depth_flights(seeds, "YAML", P, world_kw=kw, gang=2, batch=2, keyframes=kf, cam=_flight.YAML_CAM)
    cmp = _flight.compare(g, o, pos_tol=1e-6)
    print("\nyaml configuration, max_frame_count 100:", {k: v for k, v in cmp.items() if k not in ("separation_period", "dpos_final")},
          "keyframes (mean / max):", float(o["n_keyframes"].mean()), int(o["n_keyframes"].max()), "query frames (max):",
          int(o["n_query_frames"].max()), "points per frame:", float(o["n_cloud"].mean()))
    assert cmp["dpos_max_while_together"] <= 1e-6 and cmp["separated"] == 0
    assert o["n_query_frames"].max() >= 4 and o["n_cloud"].max() <= 3072
    for key in ("n_keyframes", "n_query_frames", "outliers", "map_points", "n_cloud"):
        assert np.array_equal(g[key], o[key]), key


def test_cloud_frame_flights_with_the_keyframe_map_and_the_cpp_fleet_host(tmp_path):
    """Frames handed over as clouds (+ mCurFrame.Twc = [I | odometry position] * the yaml's T_b_c, PtIsInFrame through the yaml's 64 x 48 camera), as
    bench.py --workload flight --keyframes does: the pipeline's slot map against the oracle's list, host-driven and TASK mode,
    gang 1 and 2; then a C++ host that only knows include/avoid_mpc_amd.h (tests/cpp/flight_driver.cpp with a third argument)
    flies the same flights bit for bit."""
    import os
    import struct
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seeds = list(range(2600, 2616))
    kw = dict(cyl_per_m=1.5, x_first=3.0)
    kf = dict(max_frame_count=4, th_dist=0.1, th_count=10)
    P, n = 50, 3072
    o = _flight.oracle_flights(seeds, "C1", P, n_points=n, world_kw=kw, keyframes=kf)
    assert o["n_query_frames"].max() >= 3
    runs = {}
    for mode, gang, batch in (("host", 1, 16), ("task", 2, 4)):
        g = _flight.gpu_flights(seeds, "C1", P, n_points=n, world_kw=kw, batch=batch, mode=mode, gang=gang, keyframes=kf)
        cmp = _flight.compare(g, o, pos_tol=1e-6)
        print(f"\ncloud frames + keyframes, {mode} mode, gang {gang}:", {k: v for k, v in cmp.items() if k not in ("separation_period", "dpos_final")},
              "keyframes mean / max:", float(o["n_keyframes"].mean()), int(o["n_keyframes"].max()))
        assert cmp["dpos_max_while_together"] <= 1e-6 and cmp["separated"] <= 1
        sep = cmp["separation_period"]
        for f in range(len(seeds)):
            upto = P if sep[f] < 0 else sep[f]
            assert np.array_equal(g["n_keyframes"][f, :upto], o["n_keyframes"][f, :upto]) and \
                np.array_equal(g["n_query_frames"][f, :upto], o["n_query_frames"][f, :upto]), f
        runs[mode] = g
    assert np.array_equal(runs["host"]["x"], runs["task"]["x"]) and np.array_equal(runs["host"]["flags"], runs["task"]["flags"])
    # ---- the C++ fleet host with the map in its slot
    exe = str(tmp_path / "flight_driver")
    libdir = os.path.join(ROOT, "avoid_mpc_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "flight_driver.cpp"), "-o", exe, "-L", libdir, "-lavoid_mpc_amd",
                           f"-Wl,-rpath,{libdir}"])
    prm, _ = _flight.make_prm("C1")
    S, Pc = 8, 30
    worlds = [flight.FlightWorld(s, prm, n, **kw) for s in seeds[:S]]
    st = [flight.initial_state(s, prm) for s in seeds[:S]]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("6i", S, Pc, n, n // 10, prm.K, prm.max_iter))
        f.write(np.array([prm.T, prm.dt, prm.speed, prm.safety_distance, prm.decay, prm.height, 500.0, 0.3, 0.3]).tobytes())
        f.write(np.array(prm.weights, np.float64).tobytes()); f.write(np.array(prm.tau, np.float64).tobytes())
        f.write(np.array(prm.gain, np.float64).tobytes())
        f.write(np.array([prm.a_min_z, prm.a_max_z, prm.a_max_xy, prm.a_max_yaw_dot, prm.radius]).tobytes())
        f.write(np.stack([a for a, _ in st]).tobytes()); f.write(np.stack([b for _, b in st]).tobytes())
        for t in range(Pc):
            fr = [w.frame(t) for w in worlds]
            f.write(np.stack([c for c, _ in fr]).tobytes()); f.write(np.stack([e for _, e in fr]).tobytes())
    subprocess.check_call([exe, fin, fout, str(kf["max_frame_count"])])
    want = runs["task"]
    buf = open(fout, "rb").read()
    off = 0
    for t in range(Pc):
        x = np.frombuffer(buf, np.float64, S * 10, off).reshape(S, 10); off += x.nbytes
        cmd = np.frombuffer(buf, np.float64, S * 3, off).reshape(S, 3); off += cmd.nbytes
        fl = np.frombuffer(buf, np.int32, S * 4, off).reshape(S, 4); off += fl.nbytes
        assert np.array_equal(fl, want["flags"][:S, t]) and np.array_equal(cmd, want["cmd"][:S, t]), t
        assert np.array_equal(x, want["x"][:S, t + 1]), t
    assert off == len(buf)
