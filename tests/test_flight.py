"""Closed-loop flight harness on the CPU: the pieces around the step (avoid_mpc_amd/flight.py) against the oracle's own
restatements, and the oracle / IPOPT-emulation drivers on small flights.  The GPU-vs-oracle trajectory test is
tests/test_flight_gpu.py.  Reference: AM/src/AvoidanceStateMachine.cpp:24-54,183-203,322-355,369-397."""
import numpy as np
import pytest

from avoid_mpc_amd import flight, fsm, synth
from tests import _flight, _oracle


def test_plant_is_the_oracles_model():
    """plant_step / affine_plant restate mpc_obstacle_casadi.py:106-122,338-357 exactly as oracle/mpc_oracle.c does."""
    lib = _oracle.load_oracle()
    prm = synth.MpcParams()
    tau = np.ascontiguousarray(prm.tau, np.float64)
    rng = np.random.default_rng(5)
    for _ in range(20):
        x, u = rng.normal(size=10) * 3, rng.normal(size=4) * 5
        xn = np.zeros(10)
        lib.mpco_rk4_step(np.ascontiguousarray(x), np.ascontiguousarray(u), tau, prm.dt, xn)
        assert np.abs(flight.plant_step(x, u, prm.tau, prm.dt) - xn).max() <= 1e-13
    A, B, c = flight.affine_plant(prm.tau, prm.dt)
    Ao, Bo, co = np.zeros(100), np.zeros(40), np.zeros(10)
    lib.mpco_affine(tau, prm.dt, Ao, Bo, co)
    assert np.abs(A - Ao.reshape(10, 10)).max() <= 1e-15 and np.abs(B - Bo.reshape(10, 4)).max() <= 1e-15
    assert np.abs(c - co).max() <= 1e-15
    x, u = rng.normal(size=(7, 10)), rng.normal(size=(7, 4))   # batched, and affine
    assert np.abs(flight.plant_step(x, u, prm.tau, prm.dt) - (x @ A.T + u @ B.T + c)).max() <= 1e-13


def test_get_init_path_twins_agree():
    lib = _oracle.load_oracle()
    prm = synth.MpcParams()
    rng = np.random.default_rng(2)
    ref = rng.normal(size=(prm.N, 10))
    a = fsm.get_init_path(ref.copy(), prm.speed, prm.T, 3.25, 500.0, prm.height)
    b = ref.copy()
    lib.stepo_get_init_path(b.reshape(-1), prm.N, prm.speed, prm.T, 3.25, 500.0, prm.height)
    assert np.array_equal(a, b.reshape(prm.N, 10))
    assert a[-1, 0] == prm.speed * prm.T + 3.25 and np.array_equal(a[0, [0, 1]], ref[1, [0, 1]]) and a[0, 2] == prm.height


def test_world_frames_are_a_function_of_seed_and_period():
    prm = synth.MpcParams()
    w1, w2 = flight.FlightWorld(11, prm, 5000, cyl_per_m=1.5), flight.FlightWorld(11, prm, 5000, cyl_per_m=1.5)
    c1, e1 = w1.frame(7)
    c2, e2 = w2.frame(7)
    assert c1.dtype == np.float32 and c1.shape == (5000, 3) and e1.shape == (500, 3)
    assert np.array_equal(c1, c2) and np.array_equal(e1, e2)
    c3, _ = w1.frame(8)
    assert not np.array_equal(c1, c3)                       # resampled every period
    xn = w1.x_nom(7)
    assert c1[:, 0].min() >= xn - w1.back - 0.6 and c1[:, 0].max() <= xn + w1.ahead + 0.6
    # every non-ground point lies on a cylinder surface: clearance ~0 (negative where two cylinders overlap)
    cyl = c1[c1[:, 2] > 0]
    cl = w1.clearance(cyl)
    assert cl.max() < 1e-5 and np.median(np.abs(cl)) < 1e-6
    assert w1.clearance(np.array([w1.cx[3], w1.cy[3], 1.0])) == pytest.approx(-w1.cr[3])


def test_command_is_pubcmd_or_slow_down():
    prm = synth.MpcParams()
    u = np.array([[1.0, 2.0, 9.0, 0.3], [1.0, 2.0, 9.0, 0.3]])
    x = np.zeros((2, 10)); x[:, 4:7] = [10.0, -50.0, 1.0]; x[:, 7:10] = [1.0, 1.0, -30.0]
    a = flight.command(u, np.array([[1, 1, 0, 5], [0, 0, -1, 0]]), x, prm)
    assert np.array_equal(a[0], u[0, :3])                                        # PubCmd :369-378
    assert np.allclose(a[1], [-3.3, 10.0, 15.0])   # -kp v - kd a + (0,0,9.8): (-3.3, 14.7 -> clamp 10, 18.5 -> clamp 15)  :379-397


def test_oracle_flights_small():
    """C1-sized flights on the CPU oracle: deterministic, warm-started (few iterations per period), they move forward."""
    seeds = [300, 301, 302]
    kw = dict(cyl_per_m=1.5)
    a = _flight.oracle_flights(seeds, "C1", 25, world_kw=kw, workers=3)
    b = _flight.oracle_flights(seeds[:1], "C1", 25, world_kw=kw, workers=1)
    assert np.array_equal(a["x"][0], b["x"][0]) and np.array_equal(a["flags"][0], b["flags"][0])
    prm, _ = _flight.make_prm("C1")
    st = _flight.flight_stats(a, prm)
    assert st["capped_periods"] == 0 and st["iters_per_period"] < 15 and st["x_final_mean"] > 6.0
    assert a["flags"][:, 0, 3].mean() > a["flags"][:, 5:, 3].mean()      # the cold first period costs more than the warm ones
    cmp = _flight.compare(a, a)
    assert cmp["separated"] == 0 and cmp["dpos_max_while_together"] == 0.0


def test_ipopt_emulation_flights_small():
    """The reference's solver regime (IPOPT stopped at 10 iterations, primal warm start), emulated, in the same loop: it runs,
    and on these small flights it stays near the converged-optimum flights."""
    seeds = [300, 301, 302, 303]
    kw = dict(cyl_per_m=2.0, x_first=3.0)       # obstacles from the first metres on
    a = _flight.oracle_flights(seeds, "C1", 30, world_kw=kw, workers=4)
    b = _flight.ipopt_flights(seeds, "C1", 30, world_kw=kw, workers=4)
    assert b["flags"][:, :, 1].min() >= 1 and b["ipopt_iters"].max() <= 30
    d = np.abs(a["x"][:, :, 0:3] - b["x"][:, :, 0:3]).max(axis=(1, 2))
    du = np.abs(a["u"] - b["u"]).max(axis=(1, 2))
    print("oracle vs IPOPT-10 emulation, C1, 30 periods: max |dpos| per flight", np.round(d, 4), "max |du|", np.round(du, 3))
    assert d.max() < 0.5 and du.max() > 1e-6      # different iterates, nearby trajectories


def test_flight_fixture_c2():
    """tests/golden/flight_golden.npz (make_flight_golden.py): 16 flights x 100 periods at BASELINE configs[1] size flown by the
    oracle and by the IPOPT-10 emulation.  (1) the oracle driver reproduces the committed oracle trajectories (first flights,
    first periods: a regression pin of the whole loop -- world, frames, shift, warm start, step, vehicle); (2) the statistics
    DESIGN.md quotes follow from the fixture."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "flight_golden.npz"))
    seeds, P = G["seeds"], 30
    a = _flight.oracle_flights(seeds[:4], "C2", P, world_kw=dict(cyl_per_m=float(G["cyl_per_m"])), workers=4)
    assert np.abs(a["x"][:, :, 0:3] - G["oracle.pos"][:4, :P + 1]).max() < 1e-5          # float32 fixture
    assert np.array_equal(a["flags"], G["oracle.flags"][:4, :P])
    d = np.abs(G["oracle.pos"].astype(np.float64) - G["ipopt10.pos"]).max(axis=(1, 2))
    hits_o, hits_i = int((G["oracle.clearance_min"] < 0).sum()), int((G["ipopt10.clearance_min"] < 0).sum())
    print("\noracle vs IPOPT-10 emulation over 100 periods, max |dpos| per flight: median %.2f m, p90 %.2f, max %.2f; "
          "flights through a cylinder: %d vs %d of %d; IPOPT stopped by max_iter in %.0f %% of its solves"
          % (np.median(d), np.quantile(d, 0.9), d.max(), hits_o, hits_i, len(d),
             100.0 * (G["ipopt10.status"] == 1).sum() / max(1, (G["ipopt10.status"] >= 0).sum())))
    assert np.median(d) < 1.0 and hits_o <= hits_i + 2
