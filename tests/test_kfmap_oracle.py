"""CPU: the restatement of FrameKDMap's keyframe list (tests/_kfmap.py; AM/src/FrameKDMap.cpp:29-74,233-252,428-488) on
hand-checkable inputs -- it is the checker of the device map (tests/test_kfmap_gpu.py), so its own statements are pinned here."""
import numpy as np

from tests import _kfmap


def _wall(x, n=400, seed=0, spread=2.0):
    """n points on the plane x = const (a wall ahead of a drone flying along +x), y in [-spread, spread], z in [0, 2]"""
    rng = np.random.default_rng(seed)
    return np.stack([np.full(n, x), rng.uniform(-spread, spread, n), rng.uniform(0, 2, n)], 1).astype(np.float32)


def _lattice_wall(x, ny=20, nz=20, pitch=0.2):
    """ny x nz points on the plane x = const, `pitch` apart (> the sweep's 0.1 m: no point has a chance neighbour)"""
    y, z = np.meshgrid(np.arange(ny) * pitch - 2.0, np.arange(nz) * pitch, indexing="ij")
    return np.stack([np.full(y.size, x), y.reshape(-1), z.reshape(-1)], 1).astype(np.float32)


def _twc(x):
    T = np.eye(4); T[0, 3] = x
    return T


def test_first_frame_becomes_the_only_keyframe_and_is_not_queried_twice():
    m = _kfmap.MapOracle(3, 0.1, 10, 0.1, np.eye(4))
    assert m.frames() == [] and m.summary() == (0, [])
    m.update()                                            # no frame yet: nothing happens
    c = _wall(5.0)
    m.add_vertex(c, c[:50], _twc(0.0))
    m.update()                                            # :446-449 InsertKeyFrame
    assert len(m.kfs) == 1 and m.kfs[0] is m.cur
    assert len(m.frames()) == 1                           # UpdateQueryVector leaves the newest keyframe out (:64-74)
    m.update()                                            # flag cleared: a second pass without a new frame does nothing
    assert len(m.kfs) == 1 and m.last_outliers == -1


def test_sweep_rebuilds_the_newest_keyframe_from_its_outliers_and_inserts():
    m = _kfmap.MapOracle(5, 0.1, 10, 0.1, np.eye(4))
    a = _lattice_wall(5.0)
    m.add_vertex(a, a[:20], _twc(0.0)); m.update()
    b = np.concatenate([a[:300] + np.float32(0.01), _wall(9.0, 100, seed=2)])   # 300 points moved by 1.7 cm, the other 100 of `a` gone
    m.add_vertex(b, b[:20], _twc(0.3)); m.update()
    # outliers of `a` against b: exactly the 100 points that disappeared (the lattice pitch is 0.2 m, the threshold 0.1 m)
    assert m.last_outliers == 100
    assert len(m.kfs) == 2 and m.kfs[0].kd.size() == m.last_outliers and m.kfs[1] is m.cur
    assert [f.kd.size() for f in m.frames()] == [400, m.last_outliers]
    # fewer than th_count outliers: nothing is rebuilt, nothing inserted (:477-479)
    c = b.copy(); c[:5, 0] += np.float32(1.0)
    m.add_vertex(c, c[:20], _twc(0.6)); m.update()
    assert m.last_outliers == 5 and len(m.kfs) == 2 and m.kfs[1].kd.size() == 400 and m.kfs[1] is not m.cur


def test_keyframes_behind_the_drone_and_beyond_the_limit_are_popped():
    m = _kfmap.MapOracle(2, 0.1, 10, 0.1, np.eye(4))
    for i, x in enumerate([2.0, 4.0, 6.0, 8.0]):          # four disjoint walls: every sweep finds 400 outliers and inserts
        w = _wall(x, seed=10 + i)
        m.add_vertex(w, w[:20], _twc(0.0)); m.update()
        assert len(m.kfs) <= 2 + 1                        # the pop runs before the insertion: at most max_frame_count + 1 (:450-459,486)
    assert [float(f.kd.search(np.zeros(3), 1)[2][0][0]) for f in m.kfs] == [4.0, 6.0, 8.0]   # the wall at 2 m went first
    # the drone moves past the wall at 4 m: DroneBehindPts fails for the oldest keyframe (:233-252)
    w = _wall(10.0, seed=20)
    m.add_vertex(w, w[:20], _twc(4.5)); m.update()
    assert [float(f.kd.search(np.zeros(3), 1)[2][0][0]) for f in m.kfs] == [6.0, 8.0, 10.0]
    # a keyframe of exactly 10 points: SearchForNearest(10) on 10 points returns nothing (kd_tree_two.h:119-124) -> "still ahead"
    m2 = _kfmap.MapOracle(5, 0.1, 1, 0.1, np.eye(4))
    tiny = _wall(1.0, 10, seed=3)
    m2.add_vertex(tiny, tiny[:5], _twc(0.0)); m2.update()
    far = _wall(30.0, 50, seed=4)
    m2.add_vertex(far, far[:5], _twc(20.0)); m2.update()  # the drone is 19 m past the tiny keyframe, which stays
    assert m2.kfs[0].kd.size() == 10 and len(m2.kfs) == 2


def test_rigid_inverse_and_drone_pose():
    Tbc = np.array([[0.0, 0.0, 1.0, 0.05], [-1.0, 0.0, 0.0, 0.0], [0.0, -1.0, 0.0, 0.01], [0.0, 0.0, 0.0, 1.0]])   # mpc_parameters.yaml:67-70
    inv = _kfmap.rigid_inverse(Tbc)
    assert np.abs(inv @ Tbc - np.eye(4)).max() < 1e-15
    Twb = np.eye(4); Twb[:3, 3] = [3.0, -1.0, 1.5]
    twb, bx = _kfmap.drone_pose(Twb @ Tbc, inv)
    assert np.abs(twb - [3.0, -1.0, 1.5]).max() < 1e-15 and np.abs(bx - [1.0, 0.0, 0.0]).max() < 1e-15
