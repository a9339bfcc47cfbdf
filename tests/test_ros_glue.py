"""CPU: include/avoid_mpc_amd/ros_glue.hpp compiled with g++ against message-shaped structs
(tests/cpp/ros_glue_test.cpp; no ROS in the image) and checked against closed forms of the reference's code:
PubCmd / PubSlowDownCmd (AM/src/AvoidanceStateMachine.cpp:369-397), the callbacks (:118-164), the depth image formats
cv_bridge hands to ProcessDepth (AM/src/FrameKDMap.cpp:90-101), the yaml keys (AM/src/ParameterManager.cpp:59-104,
AM/config/mpc_parameters.yaml)."""
import os
import subprocess

import numpy as np
import pytest

from avoid_mpc_amd import build as amk_build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def out(tmp_path_factory):
    amk_build.build()
    exe = str(tmp_path_factory.mktemp("ros") / "ros_glue_test")
    libdir = os.path.join(ROOT, "avoid_mpc_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "ros_glue_test.cpp"), "-o", exe, "-L", libdir,
                           "-lavoid_mpc_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    kv = {"mpc_T": 0.66, "mpc_dt": 0.033, "mpc_max_iter": 3, "nearest_point_num": 8, "speed": 10.0, "drone_radius": 0.5,
          "safety_distance": 0.2, "a_min_z": 5.0, "a_max_z": 15.0, "a_max_xy": 10.0, "a_max_yaw_dot": 10.0,
          "slow_down_kp": 0.3, "slow_down_kd": 0.3, "fx": 320, "fy": 321, "cx": 322, "cy": 240, "resize_scale": 10,
          "pixel2meter": 0.001, "depth_max": 100, "depth_min": 0.1, "T_b_c/0/2": 1.0, "T_b_c/0/0": 0.0, "T_b_c/0/3": 0.05}
    names = ["goal_p_x", "goal_p_y", "goal_p_z", "goal_yaw", "goal_v_x", "goal_v_y", "goal_v_z", "goal_a_x", "goal_a_y",
             "goal_a_z", "path_p_x", "path_p_y", "path_p_z", "path_yaw", "path_v_x", "path_v_y", "path_v_z", "path_a_x",
             "path_a_y", "path_a_z", "u_a_x", "u_a_y", "u_a_z", "u_yaw_dot", "collide_lambda"]
    kv.update(dict(zip(names, synth.DEFAULT_WEIGHTS)))
    kv.update(dict(zip(["tau_a_x", "tau_a_y", "tau_a_z", "tau_yaw_dot"], synth.DEFAULT_TAU)))
    kv.update(dict(zip(["gain_a_x", "gain_a_y", "gain_a_z", "gain_yaw_dot"], synth.DEFAULT_GAIN)))
    args = []
    for k, v in kv.items():
        args += [k, repr(float(v))]
    txt = subprocess.run([exe] + args, capture_output=True, text=True, check=True).stdout
    return {l.split()[0]: l.split()[1:] for l in txt.splitlines()}


def f(v):
    return np.array([float(x) for x in v])


def test_yaml_keys_reach_the_setters(out):
    assert f(out["T"])[0] == 0.66 and f(out["dt"])[0] == 0.033 and out["maxIter"] == ["3"] and out["K"] == ["8"]
    assert np.array_equal(f(out["weights"]), synth.DEFAULT_WEIGHTS)          # weightsName order, ParameterManager.cpp:63-68
    assert np.array_equal(f(out["taus"]), synth.DEFAULT_TAU) and np.array_equal(f(out["gains"]), synth.DEFAULT_GAIN)
    assert np.array_equal(f(out["limits"]), [5.0, 15.0, 10.0, 10.0])
    assert np.array_equal(f(out["depth"]), [0.001, 0.1, 100.0, 10.0, 320.0, 321.0, 322.0, 240.0])
    T = np.eye(4); T[0, 0] = 0.0; T[0, 2] = 1.0; T[0, 3] = 0.05
    assert np.array_equal(f(out["Tbc"]).reshape(4, 4), T)
    assert np.array_equal(f(out["step"]), [10.0, 0.2, 3.0])


def test_command_mapping_and_slow_down_fallback(out):
    assert np.array_equal(f(out["cmd_ok"]), [1, 1, 1.25, -2.5, 9.81, 0.0])          # PubCmd: ACCELERATION_MODE, u[0:3], yaw 0
    vel, acc = np.array([9.0, -40.0, 0.5]), np.array([1.0, 2.0, -30.0])
    a = -vel * 0.3 - acc * 0.3 + np.array([0, 0, 9.8])                                # PubSlowDownCmd :380-382
    want = [np.clip(a[0], -10, 10), np.clip(a[1], -10, 10), np.clip(a[2], -15, 15)]   # :383-388 (z clamped to +-aMaxZ)
    assert a[1] > 10 and a[2] > 15                                                    # both clamps are exercised
    assert np.allclose(f(out["cmd_unsafe"]), [0, 1] + want + [0.0], rtol=0, atol=1e-15)
    assert f(out["cmd_cap_default"])[0] == 1 and f(out["cmd_cap_default"])[1] == 1.25  # the reference ignores the solver status
    assert f(out["cmd_cap_fallback"])[0] == 0 and np.isclose(f(out["cmd_cap_fallback"])[1], want[0])


def test_depth_image_formats(out):
    assert out["img16"] == ["0", "2", "3", "1", "100", "200", "300", "400", "500", "65535"]   # row padding removed
    assert out["img16_packed"] == ["1", "1"]                                                   # tightly packed: zero copy
    assert out["img32be"][0] == "1" and f(out["img32be"][1:]).tolist() == [1.5, -0.25]         # byte-swapped floats
    assert out["imgbad"] == ["rejected"]                                                       # FrameKDMap.cpp:99-101


def test_callbacks(out):
    o = f(out["odom"])
    assert np.array_equal(o[:6], [1, 2, 3, 4, 5, 6]) and np.isclose(o[6], np.pi / 4) and o[7] == 10.0
    # IMUCallback: dead reckoning over 0.1 s with the PREVIOUS acceleration (zero), then acc = R accb - g
    i = f(out["imu"])
    c = np.cos(np.pi / 4)
    assert np.isclose(i[0], 1 + 4 * 0.1) and i[1] == 4.0 and np.isclose(i[2], c) and np.isclose(i[3], c) and abs(i[4]) < 1e-12
    assert np.isclose(i[5], 10.1) and np.isclose(i[6], 3 + 6 * 0.1)
    T = f(out["Twb"]).reshape(4, 4)
    R = np.array([[c, -c, 0], [c, c, 0], [0, 0, 1]])
    assert np.allclose(T[:3, :3], R, atol=1e-15) and np.array_equal(T[3], [0, 0, 0, 1])
    dt = 10.3 - 10.1
    assert np.isclose(T[0, 3], 1.4 + 4.0 * dt + 0.5 * c * dt * dt)          # DepthCallback :155-158
