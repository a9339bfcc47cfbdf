"""CPU: the interior-point study harness of round 6 (tools/experiments/ipm/, profiles/r06_ipm_iteration_study.txt) stays honest:
dump_problems.py replays a few cold bench steps of the step oracle Solve by Solve, proto.c -- which #includes oracle/mpc_oracle.c and
restates mpco_solve with switches -- returns mpco_solve's BITS with every switch off, and a switched-on variant still converges on
every problem (the study's numbers come from exactly this machinery on 128 scenes)."""
import importlib.util
import os

import numpy as np

from tests import _oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "experiments", "ipm", name + ".py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_proto_is_the_oracle_with_the_switches_off(tmp_path):
    _oracle.build_oracle()
    dump, run = _load("dump_problems"), _load("run")
    npz = str(tmp_path / "problems.npz")
    dump.main(S=3, n=5000, out=npz)                       # 3 scenes x <= 3 passes of the bench's scene model at a small cloud
    b = run.Bench(path=npz, out_dir=str(tmp_path))
    assert 3 <= len(b.P) <= 9 and set(b.pas.tolist()) <= {0, 1, 2}
    for i in range(len(b.P)):
        w, info, st = b.solve(i)
        wo, io, so = _oracle.mpco_solve(b.P[i], b.w0[i], b.lbu, b.ubu, b.N, b.K, b.prm.dt)
        assert np.array_equal(w.view(np.int64), wo.view(np.int64)) and info[1] == io[1] and info[2] == io[2], i
        assert info[0] == 0 and info[1] == int(b.D["iters"][i])          # converged, with the iteration count the step oracle logged
        w1, info1, _ = b.solve(i, variant=1, p=(0.02,))                    # the primal-dual barrier update of the study: another path, ...
        assert info1[0] == 0                                               # ... still a converged solve
