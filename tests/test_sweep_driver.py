"""CPU: the C++ multi-process sweep driver (tests/cpp/sweep_driver.cpp) at world size 2 and 3 over its host-only transport
(-DAMK_SWEEP_STUB: a shared file in place of RCCL, a closed form in place of the GPU step): block partition through
amk_shard_scene_range, equal-shard padding, the gather's layout, global scene order on rank 0, max-over-ranks timing.  The
GPU build of the same file runs in tests/test_sweep_driver_gpu.py (RCCL, world size 1 on a one-GPU box)."""
import os
import subprocess

import numpy as np
import pytest

from avoid_mpc_amd import build as amk_build, synth
from tests._sweep_io import read_output, write_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_driver(exe, stub):
    amk_build.build()
    libdir = os.path.join(ROOT, "avoid_mpc_amd")
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include")]
    if stub:
        cmd += ["-DAMK_SWEEP_STUB"]
    else:
        cmd += ["-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-Wno-unused-result"]
    cmd += [os.path.join(ROOT, "tests", "cpp", "sweep_driver.cpp"), "-o", exe, "-L", libdir, "-lavoid_mpc_amd",
            f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
    if not stub:
        cmd += ["-L", "/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)


@pytest.mark.parametrize("world,total", [(2, 7), (3, 8), (2, 6)])
def test_partition_and_exchange_on_the_stub_transport(tmp_path, world, total):
    exe = str(tmp_path / "sweep_stub")
    compile_driver(exe, stub=True)
    prm = synth.MpcParams(T=0.33, K=3)
    n, ne = 64, 8
    scenes = []
    for g in range(total):
        sc = synth.make_scene(640, 900 + g, prm)
        sc["cloud"], sc["edge"] = sc["cloud"][:n], sc["edge"][:ne]
        scenes.append(sc)
    fin, fout, rdv = str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "rdv")
    extra = write_input(fin, scenes, prm, n, ne)
    procs = [subprocess.Popen([exe, fin, fout, str(r), str(world), rdv]) for r in range(world)]
    assert [p.wait(timeout=120) for p in procs] == [0] * world
    first, count = (0, total // world + (1 if total % world else 0))
    tot, seconds, u, flags = read_output(fout, count)
    assert tot == total and seconds >= 0.0
    want = np.stack([sc["cloud"].reshape(-1)[:4].astype(np.float64) + 10.0 * px + sc["ref_path"].reshape(-1)[:4] + 100.0 * sq.reshape(-1)[:4]
                     for sc, (sq, px) in zip(scenes, extra)])
    assert np.array_equal(u, want)          # every scene exactly once, in global order, from the rank that owns it
