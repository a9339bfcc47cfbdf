"""GPU: the C++ sweep driver (tests/cpp/sweep_driver.cpp) on the real transport -- amk_pipeline_* for the steps in flight,
amk_shard_* (RCCL bound at run time: ncclCommInitRank, ncclAllGather, ncclAllReduce) for the exchange -- at the world size a
one-GPU box allows (1); once with one frame per launch, once with gangs of 3 (7 frames: two full gangs and a single).  Controls and flags against the CPU oracle's step on the same scenes."""
import json
import subprocess

import numpy as np
import pytest

from avoid_mpc_amd import synth
from tests import _oracle
from tests._sweep_io import read_output, write_input
from tests.test_sweep_driver import compile_driver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("slots,repeat,gang", [("3", "7", "0"), ("2", "7", "3")])
def test_sweep_driver_world_1_matches_the_oracle(tmp_path, slots, repeat, gang):
    exe = str(tmp_path / "sweep_gpu")
    compile_driver(exe, stub=False)
    prm = synth.MpcParams(T=0.33, K=3)
    n, total = 5000, 6
    scenes = [synth.make_scene(n, 1200 + g, prm) for g in range(total)]
    fin, fout, rdv = str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "rdv")
    extra = write_input(fin, scenes, prm, n, n // 10)
    r = subprocess.run([exe, fin, fout, "0", "1", rdv, slots, repeat, gang], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["world"] == 1 and line["scenes"] == total and line["repeat"] == 7 and line["scene_steps_per_s"] > 0
    tot, seconds, u, flags = read_output(fout, total)
    for s, (sc, (sq, px)) in enumerate(zip(scenes, extra)):
        ko, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
        m = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m.configure(prm)
        ro = _oracle.step_oracle(ko, ke, m, prm, sq, px, sc["ref_path"].copy())
        assert np.array_equal(flags[s], ro["flags"]), (s, flags[s], ro["flags"])
        assert np.abs(u[s] - ro["u"]).max() <= 1e-6, (s, u[s], ro["u"])
    print(line)
