"""GPU: bench.py as the driver runs it (`--gpus 1 --steps 20 --warmup 5`: a warm-up that does NOT fill a gang, 20 timed steps)
and with an odd pipeline shape -- one JSON line with the contract's fields, the roofline and cpu_baseline blocks, and both
parity checks green (the fixture gate and the timed workload's own scenes against the CPU oracle)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--steps", "20", "--warmup", "5"],
                                   ["--steps", "9", "--warmup", "2", "--streams", "3", "--gang", "3", "--scenes", "64", "--steady-steps", "0"],
                                   # inputs in pinned host memory, copied every step (the PCIe-inclusive rate): same controls
                                   ["--steps", "12", "--warmup", "2", "--streams", "2", "--gang", "2", "--scenes", "32", "--steady-steps", "0",
                                    "--inputs", "host"]])
def test_bench_line(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == int(extra[1]) and d["value"] > 0 and d["dtype"] == "f64"
    assert d["roofline"]["bound"] and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["parity"]["fixtures"]["ok"], d["parity"]["fixtures"]
    assert ("NOT_THE_HEADLINE_pcie_inclusive" in d) == ("host" in extra)
    chk = d["parity"]["timed_workload_vs_cpu_oracle"]
    assert chk["ok"] and chk["scenes"] == d["config"]["scenes_per_gpu"], chk


def test_bench_flight_line():
    """bench.py --workload flight (closed-loop flights in the pipeline's TASK mode), small shape: one JSON
    line, warm-started steps, parity block green (the timed run's own flights on the CPU oracle)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "flight", "--streams", "2", "--gang", "2",
                        "--scenes", "32", "--periods", "12", "--points", "20000", "--warmup", "2"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 4 * 12 and "flight" in d["config"]["workload"]
    f = d["flight"]
    assert f["flights"] == 4 * 32 and f["solves_per_step"] >= 1.0 and f["ipm_iters_per_step"] < f["ipm_iters_first_period"] * 2
    p = d["parity"]["flights_vs_cpu_oracle"]
    assert p["ok"] and p["dpos_max_while_flags_agree_m"] <= 1e-6, p


def test_bench_flight_line_with_the_keyframe_map():
    """bench.py --workload flight --config yaml --keyframes 100, small shape: the reference's own configuration (3072-point frames of a
    forward-looking sensor, N = 30, K = 3, max_frame_count 100) with the keyframe map in every slot; the map must actually hold
    keyframes (frames that reach behind the vehicle would empty it every other period: DESIGN.md section 15, row 6), and the
    timed run's own flights agree with the CPU oracle's map and step."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "flight", "--config", "yaml", "--keyframes", "100",
                        "--streams", "2", "--gang", "2", "--scenes", "16", "--periods", "30", "--warmup", "2"], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["config"]["points"] == 3072 and d["config"]["horizon"] == 30 and d["config"]["K"] == 3
    assert d["flight"]["flights"] == 4 * 16 and d["steps"] == 4 * 30
    assert d["flight"]["keyframes_at_end_mean_max"][0] >= 2.0 and d["flight"]["outliers_of_the_last_sweep_mean"] > 0, d["flight"]
    p = d["parity"]["flights_vs_cpu_oracle"]
    assert p["ok"] and p["dpos_max_while_flags_agree_m"] <= 1e-6, p
