"""bench.py's launch path without a GPU (VERDICT r3 missing #2): `python bench.py --gpus 2` started PLAINLY must spawn its own
ranks (torch.distributed.run, 127.0.0.1), and exactly one JSON line must come back from rank 0.  --dry-run keeps the ranks on
the CPU (gloo) and exercises rendezvous, the library's scene partition and a gather shaped like the sweep's."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_plain_invocation_spawns_its_ranks():
    out = _run(["--gpus", "2", "--scenes", "5"])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["gather_ok"] and out["dry_run"]
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scenes_per_rank"] == 5


def test_single_rank_needs_no_launcher():
    out = _run(["--gpus", "1"])
    assert out["n_gpus"] == 1 and out["gather_ok"]


def test_under_a_launcher_the_world_size_wins():
    """The driver's own command line (`python -m torch.distributed.run ... bench.py --gpus N`): bench.py must not spawn again."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run", "--steps", "2",
                        "--scenes", "7"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["gather_ok"] and out["padded_scenes_per_rank"] == 7
