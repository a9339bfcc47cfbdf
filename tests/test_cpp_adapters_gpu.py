"""GPU: the reference-shaped C++ adapters (include/avoid_mpc_amd/*.hpp: KDTreeTwo, FrameKDMap,
ObstacleAvoidanceMPC, AvoidanceTaskStep) compiled with g++ against the C ABI, versus the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import _oracle
from avoid_mpc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapters_match_oracle(tmp_path):
    exe = str(tmp_path / "adapter_demo")
    libdir = os.path.join(ROOT, "avoid_mpc_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_demo.cpp"), "-o", exe,
                           "-L", libdir, "-lavoid_mpc_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    prm = synth.MpcParams(T=0.33, K=3)
    n = 5000
    sc = synth.make_scene(n, 321, prm)
    sc["acc"] = np.array([0.2, -0.1, 0.05])
    N, K = prm.N, prm.K
    rng = np.random.default_rng(0)
    qs = np.stack([rng.uniform(0, 10, 16), rng.uniform(-2, 2, 16), rng.uniform(0.5, 2.5, 16)], 1)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("6i", n, n // 10, N, K, prm.max_iter, len(qs)))
        f.write(np.array([prm.T, prm.dt, prm.speed, prm.safety_distance, prm.decay, prm.height]).tobytes())
        f.write(np.array(prm.weights, np.float64).tobytes()); f.write(np.array(prm.tau, np.float64).tobytes())
        f.write(np.array(prm.gain, np.float64).tobytes())
        f.write(np.array([prm.a_min_z, prm.a_max_z, prm.a_max_xy, prm.a_max_yaw_dot, prm.radius]).tobytes())
        f.write(sc["cloud"].tobytes()); f.write(sc["edge"].tobytes())
        f.write(np.concatenate([sc["pos"], sc["vel"], sc["acc"], [sc["yaw"]]]).tobytes())
        f.write(sc["ref_path"].tobytes()); f.write(qs.tobytes())
    subprocess.check_call([exe, fin, fout])
    buf = open(fout, "rb").read()
    off = 0

    def take(dtype, cnt):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=cnt, offset=off)
        off += a.nbytes
        return a

    kd, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
    for q in qs:                                                     # 1. KDTreeTwo
        c = int(take(np.int32, 1)[0])
        ia, da, pa = kd.search(q, K)
        assert c == len(ia)
        assert np.array_equal(take(np.int32, c), ia) and np.array_equal(take(np.float64, c), da)
        assert np.array_equal(take(np.float32, 3 * c).reshape(-1, 3), pa)
    assert int(take(np.int32, 1)[0]) == 1                            # 1b. SearchForNearestBatch == one at a time
    lat = _oracle.kd_oracle((np.round(sc["cloud"] * 4) / 4).astype(np.float32))   # 1c. nanoflann tie order on a lattice
    tied = 0
    for q in qs:
        c = int(take(np.int32, 1)[0])
        ia, da, _ = lat.search(np.round(q * 8) / 8, K)
        assert c == len(ia)
        assert np.array_equal(take(np.int32, c), ia) and np.array_equal(take(np.float64, c), da)
        tied += len(np.unique(lat.search(np.round(q * 8) / 8, K + 1)[1])) < K + 1
    assert tied >= 4                                                 # these queries do have equidistant neighbours
    for q in qs:                                                     # 2. FrameKDMap
        c = int(take(np.int32, 1)[0]); d2 = take(np.float64, c); nd = take(np.float64, 1)[0]
        ce = int(take(np.int32, 1)[0]); ed2 = take(np.float64, ce)
        assert np.array_equal(d2, kd.search(q, K)[1])
        assert nd == np.sqrt(kd.search(q, 1)[1][0])
        assert np.array_equal(ed2, ke.search(q, 1)[1])
    m = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m.configure(prm)    # 3. TASK step x 2
    rp = sc["ref_path"].copy()
    sq = _oracle.scene_state_quads(sc, prm)
    for rep in range(2):
        r = _oracle.step_oracle(kd, ke, m, prm, sq, sc["pos"][0], rp)
        safe = int(take(np.int32, 1)[0]); flags = take(np.int32, 4); u = take(np.float64, 4)
        x0 = take(np.float64, 14 * N).reshape(N, 14); ref = take(np.float64, 10 * N).reshape(N, 10)
        assert safe == r["flags"][0] and np.array_equal(flags, r["flags"])
        assert np.abs(u - r["u"]).max() <= 1e-6 and np.abs(x0 - r["x0array"]).max() <= 1e-6
        assert np.abs(ref - rp).max() <= 1e-6
    nref = 20 + 10 * N + 3 * K * N                                   # 4. ObstacleAvoidanceMPC
    vec = take(np.float64, nref); u = take(np.float64, 4); info = take(np.int32, 4)
    m2 = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m2.configure(prm)
    uc, _, ic = m2.Solve(vec.copy(), True)
    assert np.abs(u - uc).max() <= 1e-6 and np.array_equal(info, ic)
    # 5. keyframes + multi-frame merge
    frames = [sc["cloud"][(sc["cloud"][:, 0] >= 1.5 * f) & (sc["cloud"][:, 0] < 1.5 * f + 12.0)] for f in range(3)]
    trees = [_oracle.kd_oracle(fr) for fr in frames]
    keyframes = []
    exp = []
    for f in range(3):
        cur = trees[f]
        if not keyframes:
            keyframes.append(cur); exp.append((1, 0)); continue
        last = keyframes[-1]
        r, n_out = last.keyframe_sweep(cur, 0.1, 10)
        if r:
            keyframes.append(cur)
        exp.append((len(keyframes), n_out))
    for f in range(3):
        kc, lo = take(np.int32, 2)
        assert (kc, lo) == exp[f], (f, kc, lo, exp[f])
    query_frames = [trees[2]] + keyframes[:-1]                    # UpdateQueryVector: cur + all key frames but the newest
    for q in qs:
        c = int(take(np.int32, 1)[0]); d2 = take(np.float64, c); nd = take(np.float64, 1)[0]
        alld = np.sort(np.concatenate([t.search(q, min(K, t.size()))[1] for t in query_frames]))[:K]
        assert np.array_equal(d2, alld)
        nn = min([t.search(q, 1)[1][0] for t in query_frames if t.size() > 1] or [np.finfo(np.float64).max])
        assert nd == np.sqrt(nn)
    # 6. ProcessDepth through the adapter == the depth oracle on the same synthetic frame
    r, c = np.meshgrid(np.arange(120), np.arange(160), indexing="ij")
    img = (1500 + (r * 37 + c * 91) % 4000).astype(np.uint16)
    img[(r * 7 + c * 13) % 11 == 0] = 0
    dprm = dict(pixel2meter=1e-3, depth_min=0.1, depth_max=100.0, resize_scale=4.0, fx=80.0, fy=80.0, cx=80.0, cy=60.0,
                Tbc=np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.0], [0, -1, 0, 0.05], [0, 0, 0, 1.0]]))
    Twb = np.array([[0.8, -0.6, 0, 2.0], [0.6, 0.8, 0, -1.0], [0, 0, 1, 1.5], [0, 0, 0, 1.0]])
    ref, _ = _oracle.depth_oracle(img, dprm, Twb)
    cnt = int(take(np.int32, 1)[0])
    pts = take(np.float32, 3 * cnt).reshape(-1, 3)
    assert cnt == len(ref) > 500 and np.array_equal(pts, ref)
    # 7. AddVertex(Twb, depth) twice: edge cloud of frame 2 uses frame 1's Twb * Tbc
    eref = _oracle.depth_edge_oracle(img, dprm, Twb @ dprm["Tbc"])[0]
    ecnt = int(take(np.int32, 1)[0])
    assert ecnt == len(eref) > 20
    ec = int(take(np.int32, 1)[0]); ed2 = take(np.float64, ec)
    et = _oracle.kd_oracle(eref)
    assert ec == 1 and ed2[0] == et.search(np.array([3.0, -1.0, 1.5]), 1)[1][0]
    # 8. PtIsInFrame: the reference's formula (general inverse of Twc, resized intrinsics, image bounds), FrameKDMap.cpp:215-231
    Twb2 = np.array([[1, 0, 0, 2.5], [0, 1, 0, -1.0], [0, 0, 1, 1.5], [0, 0, 0, 1.0]])
    Twc = Twb2 @ dprm["Tbc"]
    W, H, sc_ = 160 // 4, 120 // 4, dprm["resize_scale"]
    inside = []
    for q in qs:
        x, y, z, _ = np.linalg.inv(Twc) @ np.array([q[0], q[1], q[2], 1.0])
        ok = not (z > dprm["depth_max"] or z < 0)
        if ok:
            u = dprm["fx"] / sc_ * x / z + dprm["cx"] / sc_; v = dprm["fy"] / sc_ * y / z + dprm["cy"] / sc_
            ok = not (u < 0 or u >= W or v < 0 or v >= H)
        inside.append(int(ok))
    got = take(np.int32, len(qs))
    assert np.array_equal(got, inside) and 0 < sum(inside) < len(qs), (got, inside)
    assert off == len(buf)
