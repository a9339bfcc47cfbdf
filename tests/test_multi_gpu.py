"""GPU, auto-arming: the multi-GPU path on REAL RCCL at world size 2 (BASELINE.json configs[3]).  A one-GPU
box -- every box these tests have run on so far -- skips them (`torch.cuda.device_count() < 2`); the first box with two devices
runs them unattended:
  (a) bench.py --gpus 2: one JSON line with n_gpus 2, one RCCL instance in every rank (the library binds the copy torch mapped),
      the per-rank files, and a whole-job value of at least 1.8 x the same box's N = 1 line of the same command;
  (b) the C++ sweep driver (tests/cpp/sweep_driver.cpp) at world 2 over amk_shard_* (ncclCommInitRank / ncclAllGather): the
      gathered controls in GLOBAL scene order equal the CPU oracle's step, scene by scene.
World-2 LOGIC (partition, padding, gather layout, max-over-ranks) is covered without a GPU on gloo / the stub transport:
tests/test_shard_gloo.py, tests/test_bench_launch.py, tests/test_sweep_driver.py."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs on the box (auto-arms on the first multi-GPU lease)")


def _bench(gpus, rank_dir, extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", AMK_BENCH_RANK_DIR=str(rank_dir))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "16", "--warmup", "4",
                        "--steady-steps", "0", "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@needs_two
def test_bench_two_ranks_on_rccl(tmp_path):
    one = _bench(1, tmp_path / "n1")
    two = _bench(2, tmp_path / "n2")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["steps"] == 16
    rc = two["config"]["rccl"]
    assert rc["single_rccl_instance"] is True, rc
    ranks = sorted(glob.glob(str(tmp_path / "n2" / "rank*.json")))
    assert len(ranks) >= 2, ranks                       # every rank leaves its own account (device, RCCL binding, local rate)
    seen = {json.load(open(f)).get("rank") for f in ranks}
    assert {0, 1} <= seen, seen
    assert two["parity"]["fixtures"]["ok"], two["parity"]
    assert two["value"] >= 1.8 * one["value"], (one["value"], two["value"])   # weak scaling: twice the scenes in about the same time
    print(json.dumps(dict(n1=one["value"], n2=two["value"], ratio=two["value"] / one["value"])))


@needs_two
def test_sweep_driver_world_2_on_rccl_matches_the_oracle(tmp_path):
    from avoid_mpc_amd import synth
    from tests import _oracle
    from tests._sweep_io import read_output, write_input
    from tests.test_sweep_driver import compile_driver
    exe = str(tmp_path / "sweep_gpu")
    compile_driver(exe, stub=False)
    prm = synth.MpcParams(T=0.33, K=3)
    n, total = 5000, 7                                   # uneven: ranks own 4 + 3 scenes, the gather pads to equal shards
    scenes = [synth.make_scene(n, 1300 + g, prm) for g in range(total)]
    fin, fout, rdv = str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "rdv")
    extra = write_input(fin, scenes, prm, n, n // 10)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, fin, fout, str(r), "2", rdv, "2", "3", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-3000:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["world"] == 2 and line["scenes"] == total
    tot, seconds, u, flags = read_output(fout, 4)        # rank 0 owns the first ceil(7 / 2) scenes: its flags travel with the file
    assert tot == total
    for s, (sc, (sq, px)) in enumerate(zip(scenes, extra)):
        ko, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
        m = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m.configure(prm)
        ro = _oracle.step_oracle(ko, ke, m, prm, sq, px, sc["ref_path"].copy())
        assert np.abs(u[s] - ro["u"]).max() <= 1e-6, (s, u[s], ro["u"])    # GLOBAL scene order, whichever rank computed it
        if s < 4:
            assert np.array_equal(flags[s], ro["flags"]), (s, flags[s], ro["flags"])


def test_the_multi_gpu_tests_are_armed():
    """Runs everywhere: says in the test log how many devices the box had, i.e. whether the two tests above ran or skipped."""
    n = _n_devices()
    print(f"multi-GPU tests: {n} device(s) visible -> {'ran' if n >= 2 else 'skipped (auto-arm at >= 2)'}")
    assert n >= 1
