"""Input / output files of tests/cpp/sweep_driver.cpp (shared by the CPU stub test and the GPU test)."""
import struct

import numpy as np

from avoid_mpc_amd import fsm, synth


def write_input(path, scenes, prm, n, ne):
    """scenes: list of synth.make_scene dicts -> in.bin; returns the per-scene (state_quads, pos_x) used."""
    extra = []
    with open(path, "wb") as f:
        f.write(struct.pack("6i", len(scenes), n, ne, prm.N, prm.K, prm.max_iter))
        f.write(np.array([prm.T, prm.dt, prm.speed, prm.safety_distance]).tobytes())
        f.write(np.array(prm.weights, np.float64).tobytes()); f.write(np.array(prm.tau, np.float64).tobytes())
        f.write(np.array(prm.gain, np.float64).tobytes())
        f.write(np.array([prm.a_min_z, prm.a_max_z, prm.a_max_xy, prm.a_max_yaw_dot, prm.radius]).tobytes())
        for sc in scenes:
            sq = fsm.state_quads(sc["pos"], sc["vel"], sc["acc"], sc["yaw"], prm.decay, prm.max_iter)
            f.write(np.ascontiguousarray(sc["cloud"], np.float32).tobytes())
            f.write(np.ascontiguousarray(sc["edge"], np.float32).tobytes())
            f.write(np.ascontiguousarray(sq, np.float64).tobytes())
            f.write(np.array([sc["pos"][0]], np.float64).tobytes())
            f.write(np.ascontiguousarray(sc["ref_path"], np.float64).tobytes())
            extra.append((sq, float(sc["pos"][0])))
    return extra


def read_output(path, n_flags_scenes):
    buf = open(path, "rb").read()
    total = struct.unpack_from("i", buf, 0)[0]
    seconds = struct.unpack_from("d", buf, 4)[0]
    u = np.frombuffer(buf, np.float64, total * 4, 12).reshape(total, 4)
    flags = np.frombuffer(buf, np.int32, n_flags_scenes * 4, 12 + total * 32).reshape(-1, 4)
    return total, seconds, u, flags
