"""Pins the C restatement of the KD path (oracle/kd_oracle.c) against the reference's own
nanoflann header compiled in place (oracle/_ref) and against the committed golden vectors."""
import numpy as np
import pytest

from tests import _oracle
from avoid_mpc_amd import synth


def _clouds():
    rng = np.random.default_rng(1)
    out = {}
    out["uniform5k"] = rng.uniform(-10, 10, (5000, 3)).astype(np.float32)
    out["corridor20k"] = synth.make_cloud(20000, 3)[0]
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(10), indexing="ij"), -1)
    out["grid_ties"] = g.reshape(-1, 3).astype(np.float32)              # massive exact ties
    out["dupes"] = np.repeat(rng.uniform(-1, 1, (300, 3)).astype(np.float32), 4, axis=0)
    out["planar"] = np.concatenate([rng.uniform(-5, 5, (3000, 2)), np.zeros((3000, 1))], 1).astype(np.float32)
    out["tiny7"] = rng.uniform(-1, 1, (7, 3)).astype(np.float32)
    out["one"] = np.array([[1.0, 2.0, 3.0]], np.float32)
    return out


@pytest.mark.parametrize("name", list(_clouds().keys()))
@pytest.mark.parametrize("k", [1, 3, 8])
def test_restatement_equals_reference(name, k, oracle, ref_kd):
    cloud = _clouds()[name]
    a = _oracle.kd_oracle(cloud)
    b = _oracle.kd_ref(cloud)
    rng = np.random.default_rng(7)
    lo, hi = cloud.min(0) - 1.0, cloud.max(0) + 1.0
    qs = rng.uniform(lo, hi, (200, 3))
    qs[:20] = cloud[rng.integers(0, len(cloud), 20)]                     # queries on data points
    if name == "grid_ties":
        qs[20:40] = rng.integers(0, 9, (20, 3)) + 0.5                    # equidistant to 8 corners
    for q in qs:
        ia, da, pa = a.search(q, k)
        ib, db, pb = b.search(q, k)
        assert np.array_equal(ia, ib), (name, k, q)
        assert np.array_equal(da.view(np.int64), db.view(np.int64))
        assert np.array_equal(pa, pb)
        ra, rb = a.search_raw(q, k), b.search_raw(q, k)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])


def test_nan_x_filter_and_count_quirk(oracle, ref_kd):
    rng = np.random.default_rng(2)
    cloud = rng.uniform(-1, 1, (50, 3)).astype(np.float32)
    cloud[[3, 17, 40], 0] = np.nan                                       # dropped (kd_tree_two.h:99)
    a, b = _oracle.kd_oracle(cloud), _oracle.kd_ref(cloud)
    assert a.size() == b.size() == 47
    q = np.array([0.1, 0.2, 0.3])
    for n in (1, 8, 46, 47, 48, 60):
        ia, da, _ = a.search(q, n)
        ib, db, _ = b.search(q, n)
        assert np.array_equal(ia, ib) and np.array_equal(da.view(np.int64), db.view(np.int64))
        if n == 47:
            assert len(ia) == 0                                          # size == n quirk (:119-124)
        elif n > 47:
            assert len(ia) == 47
        else:
            assert len(ia) == n


def test_nan_y_is_kept(oracle, ref_kd):
    rng = np.random.default_rng(3)
    cloud = rng.uniform(-1, 1, (50, 3)).astype(np.float32)
    cloud[5, 1] = np.nan                                                 # kept: only x is tested (:99)
    a, b = _oracle.kd_oracle(cloud), _oracle.kd_ref(cloud)
    assert a.size() == b.size() == 50
    for q in rng.uniform(-1, 1, (20, 3)):
        ia, da, _ = a.search(q, 8)
        ib, db, _ = b.search(q, 8)
        assert np.array_equal(ia, ib) and np.array_equal(da.view(np.int64), db.view(np.int64))
        assert 5 not in ia                                               # NaN distance never inserted


def test_empty_cloud(oracle, ref_kd):
    cloud = np.zeros((0, 3), np.float32)
    a, b = _oracle.kd_oracle(cloud), _oracle.kd_ref(cloud)
    assert a.size() == b.size() == 0
    assert len(a.search(np.zeros(3), 3)[0]) == 0 and len(b.search(np.zeros(3), 3)[0]) == 0


def test_bruteforce_agrees_when_no_ties(oracle):
    cloud = synth.make_cloud(5000, 11)[0]
    a = _oracle.kd_oracle(cloud)
    rng = np.random.default_rng(5)
    for q in rng.uniform([0, -8, 0], [30, 8, 4], (100, 3)):
        i1, d1 = a.search_raw(q, 8)
        i2, d2 = a.bruteforce(q, 8)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)


def test_fma_build_of_reference_same_indices(oracle):
    """The reference's own flags (-O3 -march=native) contract the distance into FMAs: index lists
    must not change (SURVEY.md appendix C finding 3)."""
    fast = _oracle.load_ref(strict=False)
    if fast is None:
        pytest.skip("oracle/_ref not built")
    cloud = synth.make_cloud(20000, 5)[0]
    a = _oracle.kd_oracle(cloud)
    b = _oracle.kd_ref(cloud, strict=False)
    rng = np.random.default_rng(9)
    for q in rng.uniform([0, -8, 0], [30, 8, 4], (300, 3)):
        assert np.array_equal(a.search(q, 8)[0], b.search(q, 8)[0])
