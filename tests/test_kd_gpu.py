"""GPU: the HIP KD path (through the C ABI) against the oracle and the golden vectors.
Indices bit-exact, squared distances bit-exact (IEEE, no FMA), counts per kd_tree_two.h:119-124."""
import os

import numpy as np
import pytest

from tests import _oracle
from avoid_mpc_amd import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "kd_golden.npz"))
FMAX = np.finfo(np.float64).max


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test needs a GPU"
    return torch


MODES = {"grid": 0, "scan": 1}      # bucketed index (default product path) / streaming scan (cross-check)
MODE = ["grid"]


@pytest.fixture(autouse=True, params=["grid", "scan"])
def search_mode(request):
    MODE[0] = request.param
    yield
    MODE[0] = "grid"


def _gpu_search(torch, clouds, queries, k, counts=None, stride=3):
    """clouds: list of [n_i,3] float32 -> batched device search; returns host arrays."""
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import KdBatch
    S = len(clouds)
    nmax = max(max(len(c) for c in clouds), 1)
    buf = np.full((S, nmax, stride), 7.0, np.float32)
    cnt = np.zeros(S, np.int32)
    for s, c in enumerate(clouds):
        buf[s, :len(c), :3] = c
        cnt[s] = len(c)
    kd = KdBatch(S, nmax)
    capi.load().amk__kd_set_mode(kd.h, MODES[MODE[0]])
    kd.build(torch.from_numpy(buf).cuda(), torch.from_numpy(cnt).cuda())
    out = kd.search(torch.from_numpy(np.ascontiguousarray(queries, np.float64)).cuda(), k)
    torch.cuda.synchronize()
    res = {n: (v.cpu().numpy() if v is not None else None) for n, v in out.items()}
    res["sizes"] = kd.sizes()
    kd.close()
    return res


@pytest.mark.parametrize("name", ["uniform2k", "corridor3k", "tiny5", "nan_x500"])
def test_golden_small(name, torch_cuda):
    cloud, qs = G[f"{name}.cloud"], G[f"{name}.queries"]
    ks = (1, 3, 8) + ((5, 7) if name == "tiny5" else ())
    for k in ks:
        r = _gpu_search(torch_cuda, [cloud], qs[None], k)
        cnt = G[f"{name}.k{k}.counts"]
        assert np.array_equal(r["counts"][0], cnt)
        for i in range(len(qs)):
            c = cnt[i]
            assert np.array_equal(r["indices"][0, i, :c], G[f"{name}.k{k}.indices"][i, :c])
            assert np.array_equal(r["sqdist"][0, i, :c].view(np.int64), G[f"{name}.k{k}.sqdist"][i, :c].view(np.int64))
            assert (r["indices"][0, i, c:] == -1).all() and (r["sqdist"][0, i, c:] == FMAX).all()


@pytest.mark.parametrize("tag", ["c1_5k", "c2_50k", "c5_200k"])
def test_golden_baseline_sizes(tag, torch_cuda):
    n, seed = (int(v) for v in G[f"{tag}.seed"])
    cloud, edge = synth.make_cloud(n, seed)
    qs = G[f"{tag}.queries"]
    r = _gpu_search(torch_cuda, [cloud], qs[None], 8)
    assert np.array_equal(r["indices"][0], G[f"{tag}.k8.indices"])
    assert np.array_equal(r["sqdist"][0].view(np.int64), G[f"{tag}.k8.sqdist"].view(np.int64))
    assert np.array_equal(r["pts"][0], cloud[G[f"{tag}.k8.indices"]])
    r = _gpu_search(torch_cuda, [edge], qs[None], 1)
    assert np.array_equal(r["indices"][0], G[f"{tag}.edge.k1.indices"])
    assert np.array_equal(r["sqdist"][0].view(np.int64), G[f"{tag}.edge.k1.sqdist"].view(np.int64))


def test_batch_ragged_vs_oracle(torch_cuda, oracle):
    """Many scenes of different sizes (incl. empty, 1 point, NaN-x points), pcl 16-byte stride."""
    rng = np.random.default_rng(4)
    sizes = [0, 1, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1000, 4097, 20000, 333, 2, 5000, 12, 100, 1024]
    clouds = []
    for i, n in enumerate(sizes):
        c = synth.make_cloud(max(n, 10), 100 + i)[0][:n].copy()
        if n > 20:
            c[rng.choice(n, n // 10, replace=False), 0] = np.nan
        clouds.append(c)
    Q, k = 22, 8
    qs = np.stack([rng.uniform(0, 15, (len(sizes), Q)), rng.uniform(-3, 3, (len(sizes), Q)),
                   rng.uniform(0, 3, (len(sizes), Q))], -1)
    r = _gpu_search(torch_cuda, clouds, qs, k, stride=4)
    for s, c in enumerate(clouds):
        t = _oracle.kd_oracle(c)
        assert r["sizes"][s] == t.size()
        for q in range(Q):
            ia, da, pa = t.search(qs[s, q], k)
            cnt = r["counts"][s, q]
            assert cnt == len(ia)
            assert np.array_equal(r["indices"][s, q, :cnt], ia)
            assert np.array_equal(r["sqdist"][s, q, :cnt].view(np.int64), da.view(np.int64))
            assert np.array_equal(r["pts"][s, q, :cnt], pa)


def test_ties_follow_lowest_index_policy(torch_cuda, oracle):
    """Exact distance ties: nanoflann resolves them by traversal order (unspecified by any contract);
    the HIP path resolves them by lowest index == the ordered brute force of the oracle.  The
    distance lists must still equal the reference's."""
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(10), indexing="ij"), -1)
    cloud = g.reshape(-1, 3).astype(np.float32)
    dup = np.repeat(np.random.default_rng(1).uniform(-1, 1, (200, 3)).astype(np.float32), 3, axis=0)
    rng = np.random.default_rng(3)
    for c, qs in ((cloud, rng.integers(0, 9, (32, 3)) + 0.5), (dup, rng.uniform(-1, 1, (32, 3)))):
        t = _oracle.kd_oracle(c)
        for k in (1, 4, 8):
            r = _gpu_search(torch_cuda, [c], qs[None], k)
            for i, q in enumerate(qs):
                ib, db = t.bruteforce(q, k)
                assert np.array_equal(r["indices"][0, i], ib)
                assert np.array_equal(r["sqdist"][0, i], db)
                assert np.array_equal(r["sqdist"][0, i], t.search_raw(q, k)[1])   # same distances as nanoflann


def test_fp32_prefilter_is_conservative(torch_cuda, oracle):
    """The scan rejects most points with an fp32 distance test; it must never reject a true neighbour.
    Stress it where fp32 is weakest: coordinates of magnitude 1e3..1e6 (fp32 ulp 6e-5..6e-2) with
    neighbours much closer to each other than an ulp of the query, queries far outside the cloud,
    and a cloud so small that fp32 squares go denormal."""
    rng = np.random.default_rng(12)
    cases = []
    for mag in (1e3, 1e5, 1e6):
        c = (mag + rng.uniform(-1.0, 1.0, (4000, 3))).astype(np.float32)
        q = mag + rng.uniform(-1.5, 1.5, (22, 3))
        cases.append((c, q))
    c = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    cases.append((c, rng.uniform(-1, 1, (22, 3)) + np.array([1e7, 0, 0])))         # far query
    cases.append((c * 1e-20, rng.uniform(-1, 1, (22, 3)) * 1e-20))                  # tiny scale (fp32 denormal squares)
    for cloud, qs in cases:
        t = _oracle.kd_oracle(cloud)
        for k in (1, 8):
            r = _gpu_search(torch_cuda, [cloud], qs[None], k)
            for i, q in enumerate(qs):
                ib, db = t.bruteforce(q, k)                                         # ties possible at 1e6: index order
                assert np.array_equal(r["indices"][0, i, :len(ib)], ib)
                assert np.array_equal(r["sqdist"][0, i, :len(ib)].view(np.int64), db.view(np.int64))


def test_single_query_and_host_api(torch_cuda, oracle):
    from avoid_mpc_amd.host import KdBatch
    cloud = synth.make_cloud(5000, 77)[0]
    t = _oracle.kd_oracle(cloud)
    kd = KdBatch(1, 5000)
    from avoid_mpc_amd import capi
    capi.load().amk__kd_set_mode(kd.h, MODES[MODE[0]])
    kd.build_host(cloud[None])
    rng = np.random.default_rng(8)
    for Q, k in ((1, 1), (1, 8), (3, 3), (30, 10), (64, 64)):
        qs = rng.uniform([0, -4, 0], [20, 4, 3], (1, Q, 3))
        r = kd.search_host(qs, k)
        for q in range(Q):
            ia, da, pa = t.search(qs[0, q], k)
            assert np.array_equal(r["indices"][0, q], ia) and np.array_equal(r["sqdist"][0, q], da)
            assert np.array_equal(r["pts"][0, q], pa)
    kd.close()


def test_argument_errors(torch_cuda):
    import ctypes as C
    from avoid_mpc_amd import capi
    lib = capi.load()
    h = C.c_void_p()
    assert lib.amk_kd_create(0, 10, C.byref(h)) == 1
    assert lib.amk_kd_create(1, 10, C.byref(h)) == 0
    assert lib.amk_kd_search(h, None, 1, 1, None, None, None, None, None) == 1
    q = torch_cuda.zeros((1, 1, 3), dtype=torch_cuda.float64, device="cuda")
    assert lib.amk_kd_search(h, C.c_void_p(q.data_ptr()), 1, 65, None, None, None, None, None) == 4
    assert lib.amk_kd_destroy(h) == 0


def test_points_outside_the_sampled_box_and_nan_runs(torch_cuda, oracle):
    """amk_kd_build takes its grid box from a sample of a large cloud (runs of 64 points, one run in 16): far outliers at unsampled positions are
    clamped into boundary cells and must still be found (or correctly ignored); runs of NaN-x points shift the cloud
    indices of everything behind them."""
    rng = np.random.default_rng(21)
    n = 40000
    pts = rng.uniform([-5, -5, 0], [5, 5, 3], (n, 3)).astype(np.float32)
    out = np.arange(n)[np.arange(n) % 1024 >= 64][::97][:300]        # never a sampled position
    pts[out] = rng.uniform(-60, 60, (len(out), 3)).astype(np.float32)
    q = np.concatenate([rng.uniform([-6, -6, -1], [6, 6, 4], (40, 3)), pts[out[:12]].astype(np.float64) + 0.25,
                        rng.uniform(-70, 70, (12, 3))])                # (a NaN query is undefined in the reference too)
    pts[1000:1200, 0] = np.nan                                        # three full 64-point groups and two ragged ones
    pts[rng.choice(n, 500, replace=False), 0] = np.nan
    res = _gpu_search(torch_cuda, [pts], q[None], 8)
    tree = _oracle.kd_oracle(pts)
    for j, qq in enumerate(q):
        ri, rd, _ = tree.search(qq, 8)
        assert np.array_equal(res["indices"][0, j], ri) and np.array_equal(res["sqdist"][0, j], rd), j


def test_degenerate_grids_and_far_queries(torch_cuda, oracle):
    """Clouds whose bounding box is (nearly) a line or a plane give grids of 1024 x 1 x 1 or 32 x 32 x 1 cells; a query far
    from the cloud walks many rings before its k-th distance reaches the next face.  The ring walk enumerates only the rows
    inside the grid (and decodes them through float reciprocals): results must still be the oracle's."""
    rng = np.random.default_rng(311)
    n = 20000
    t = rng.uniform(0, 1, n).astype(np.float32)
    line = np.stack([1000.0 * t, 0.001 * rng.standard_normal(n), 0.001 * rng.standard_normal(n)], 1).astype(np.float32)
    plane = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), 0.001 * rng.standard_normal(n)], 1).astype(np.float32)
    diag = np.stack([300 * t, 300 * t + 0.01 * rng.standard_normal(n), 0.5 * rng.standard_normal(n)], 1).astype(np.float32)
    q = np.concatenate([rng.uniform([-100, -5, -5], [1100, 5, 5], (16, 3)), rng.uniform(-400, 400, (16, 3)),
                        np.array([[500.0, 300.0, 0.0], [-2000.0, 0.0, 0.0], [0.0, 0.0, 900.0]])])
    clouds = [line, plane, diag]
    qs = np.broadcast_to(q, (len(clouds),) + q.shape)
    for k in (1, 8):
        res = _gpu_search(torch_cuda, clouds, qs, k)
        for s, c in enumerate(clouds):
            tree = _oracle.kd_oracle(c)
            for j in range(len(q)):
                ri, rd, _ = tree.search(q[j], k)
                assert np.array_equal(res["indices"][s, j, :len(ri)], ri) and np.array_equal(res["sqdist"][s, j, :len(ri)], rd), (s, j, k)


def test_candidate_windows_across_item_boundaries(torch_cuda, oracle):
    """grid_knn flattens the bucket ranges of a shell's (row, tile) items into one candidate index space and reads it in
    windows of 64; a candidate finds its item through the slots the items announce plus a running maximum, the item that
    holds the window's first candidate through a ballot (csrc/kd_grid.h: fetch).  Clouds that stress that mapping: one cell
    far denser than a window (an item spans many windows: carry only), runs of exactly 64 / 128 points per bucket (items
    begin ON the window boundaries), single points scattered over many buckets and tiles (dozens of one-point items per
    window, most lanes' items empty), and the mix of all three; k up to 16."""
    rng = np.random.default_rng(77)
    blob = (np.array([5.0, 0.0, 1.5]) + 1e-3 * rng.standard_normal((2500, 3))).astype(np.float32)
    sparse = rng.uniform([0, -8, 0], [30, 8, 4], (500, 3)).astype(np.float32)
    dense_cell = np.concatenate([sparse, blob])[rng.permutation(3000)]
    # 48 tight clusters of exactly 64 points (+ 8 of 128) on a lattice: every bucket run is a whole number of windows
    centres = np.stack(np.meshgrid(np.arange(4) * 7.0, np.arange(-2, 2) * 4.0, np.arange(3) * 1.3, indexing="ij"), -1).reshape(-1, 3)
    lattice = np.concatenate([c + 1e-2 * rng.standard_normal((128 if i < 8 else 64, 3)) for i, c in enumerate(centres)]).astype(np.float32)
    # 9000 points = three tiles of the one-pass build, uniformly thin: ~3 points per bucket and tile
    thin = rng.uniform([0, -8, 0], [30, 8, 4], (9000, 3)).astype(np.float32)
    mix = np.concatenate([thin[:5000], blob[:700], lattice[:1024]])[rng.permutation(6724)]
    clouds = [dense_cell, lattice, thin, mix]
    q = np.concatenate([blob[:6].astype(np.float64) + 0.02, centres[:10] + np.array([0.31, 0.17, 0.23]), rng.uniform([0, -8, 0], [30, 8, 4], (16, 3)),
                        np.array([[5.0, 0.0, 1.5], [40.0, 0.0, 2.0], [-3.0, 9.0, 5.0]])])
    qs = np.broadcast_to(q, (len(clouds),) + q.shape)
    for k in (1, 8, 16):
        res = _gpu_search(torch_cuda, clouds, qs, k)
        for s, c in enumerate(clouds):
            tree = _oracle.kd_oracle(c)
            for j in range(len(q)):
                ri, rd, _ = tree.search(q[j], k)
                assert np.array_equal(res["indices"][s, j, :len(ri)], ri), (s, j, k)
                assert np.array_equal(res["sqdist"][s, j, :len(ri)].view(np.int64), rd.view(np.int64)), (s, j, k)


def test_tile_boundaries_of_the_one_pass_build(torch_cuda, oracle):
    """The index build cuts a cloud into tiles of 4096 points (kd_grid.h grid_build_tiles_scene): ragged batch whose sizes
    sit on, before and behind tile and 512-point round boundaries, NaN-x runs across a tile boundary (the cloud indices of
    everything behind them shift), a tile that is all NaN, pcl::PointXYZ stride -- indices, distances and
    sizes must equal the oracle's on the NaN-x-filtered cloud."""
    rng = np.random.default_rng(77)
    sizes = [4096, 4097, 4095, 8192, 8193, 512, 513, 1, 12289, 20000, 0, 3]
    clouds = []
    for i, n in enumerate(sizes):
        c = rng.uniform([-4, -4, 0], [4, 4, 3], (n, 3)).astype(np.float32)
        if n > 4200:
            c[4090:4110, 0] = np.nan                      # a run across the first tile boundary
        if n >= 12289:
            c[8192:12288, 0] = np.nan                     # the third tile holds no point at all
        if n > 600:
            c[rng.choice(n, 37, replace=False), 0] = np.nan
        clouds.append(c)
    q = rng.uniform([-5, -5, -1], [5, 5, 4], (24, 3))
    qs = np.broadcast_to(q, (len(sizes),) + q.shape)
    for stride in (3, 4):
        for k in (1, 8):
            res = _gpu_search(torch_cuda, clouds, qs, k, stride=stride)
            for s, c in enumerate(clouds):
                tree = _oracle.kd_oracle(c)
                assert res["sizes"][s] == tree.size(), (s, stride)
                for j in range(len(q)):
                    ri, rd, _ = tree.search(q[j], k)
                    cnt = len(ri)
                    assert res["counts"][s, j] == cnt, (s, j, k)
                    assert np.array_equal(res["indices"][s, j, :cnt], ri[:cnt]) and np.array_equal(res["sqdist"][s, j, :cnt], rd[:cnt]), (s, j, k)


def test_tie_flags_mark_every_query_whose_indices_may_differ_from_nanoflann(torch_cuda, oracle):
    """amk_kd_tie_flags.  (a) tie-free random cloud: no flag, indices == the reference-pinned oracle (traversal order).
    (b) a QUANTISED cloud as the edge pipeline makes them (8-bit depth on a pixel grid, FrameKDMap.cpp:180-200: exact
    ties are common): wherever the index list differs from nanoflann's the flag is set, wherever the flag is clear the
    index list is identical; the distance lists are identical everywhere.  Reports how often the SETS differ."""
    torch = torch_cuda
    from avoid_mpc_amd.host import KdBatch, kd_tie_flags
    rng = np.random.default_rng(5)
    clouds = {"random": rng.uniform(-5, 5, (4000, 3)).astype(np.float32)}
    u, v = np.meshgrid(np.arange(64), np.arange(48))
    # byte * (max - min) / 200 with walls at two quantised depths: the points of a wall form a regular lattice
    depth = np.where(((u // 8 + v // 8) % 2) == 0, 40, 60) * (100.0 / 200.0)
    clouds["quantised"] = np.stack([(u - 32.0) * depth / 32.0, (v - 24.0) * depth / 32.0, depth], -1).reshape(-1, 3).astype(np.float32)
    for name, c in clouds.items():
        t = _oracle.kd_oracle(c)
        qs = c[rng.integers(0, len(c), 48)].astype(np.float64) + (0.0 if name == "quantised" else 0.01)
        for k in (1, 3, 8):
            kd = KdBatch(1, len(c)); kd.build(torch.from_numpy(c[None]).cuda())
            qd = torch.from_numpy(qs[None].copy()).cuda()
            r = kd.search(qd, k)
            fl = kd_tie_flags(kd, qd, k).cpu().numpy()[0]
            torch.cuda.synchronize()
            idx, d2 = r["indices"].cpu().numpy()[0], r["sqdist"].cpu().numpy()[0]
            differ = sets_differ = 0
            for i, q in enumerate(qs):
                ia, da = t.search_raw(q, k)                      # nanoflann's order (oracle pinned to the reference)
                assert np.array_equal(d2[i][:len(da)], da)
                if not np.array_equal(idx[i][:len(ia)], ia):
                    differ += 1
                    sets_differ += set(idx[i][:len(ia)]) != set(ia)
                    assert fl[i] == 1, (name, k, i)
                # the flag itself: an exact tie among the k + 1 nearest
                db = t.bruteforce(q, k + 1)[1]
                assert fl[i] == int(np.any(np.diff(db) == 0)), (name, k, i, db)
            print(f"{name} k={k}: flagged {int(fl.sum())}/48, index lists differing from nanoflann {differ}, sets differing {sets_differ}")
            if name == "random":
                assert fl.sum() == 0 and differ == 0
            elif k > 1:
                assert fl.sum() > 0          # the lattice really produces exact ties
            kd.close()


def test_nanoflann_tie_order_mode(torch_cuda, oracle):
    """amk_kd_set_tie_order(AMK_TIES_NANOFLANN): the handle builds the reference's own tree on the device and answers by
    nanoflann's own traversal -- index lists (not only distances) identical to the reference on tie-heavy clouds:
    integer lattice, triplicated points, a quantised two-wall depth lattice, a planar cloud, plus random / tiny / NaN-x
    clouds; node counts equal the oracle's tree (the oracle is pinned to the reference header, tests/test_kd_oracle.py,
    and the reference itself is consulted directly when oracle/_ref is present)."""
    import ctypes as C
    torch = torch_cuda
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import KdBatch
    lib = capi.load()
    rng = np.random.default_rng(8)
    g = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(10), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    u, v = np.meshgrid(np.arange(64), np.arange(48))
    depth = np.where(((u // 8 + v // 8) % 2) == 0, 40, 60) * (100.0 / 200.0)
    lattice = np.stack([(u - 32.0) * depth / 32.0, (v - 24.0) * depth / 32.0, depth], -1).reshape(-1, 3).astype(np.float32)
    dup = np.repeat(rng.uniform(-1, 1, (300, 3)).astype(np.float32), 3, axis=0)
    planar = rng.uniform(-3, 3, (2000, 3)).astype(np.float32); planar[:, 2] = 1.0
    nanx = rng.uniform(-2, 2, (700, 3)).astype(np.float32); nanx[::7, 0] = np.nan
    # geometric coordinates: every box midpoint cuts off ONE point, the tree is a chain (the register-resident subtrees must
    # not queue the one-point children: a 40-point chain is 29 levels deep)
    geo = rng.uniform(-0.1, 0.1, (40, 3)).astype(np.float32); geo[:, 0] = (3.0 ** np.arange(40)).astype(np.float32)
    geo_neg = geo.copy(); geo_neg[:, 0] = -geo_neg[:, 0]
    geo_mix = rng.uniform(-5, 5, (3000, 3)).astype(np.float32); geo_mix[:36, 1] = (2.5 ** np.arange(3, 39)).astype(np.float32)
    clouds = dict(grid=g, lattice=lattice, dup=dup, planar=planar, random=rng.uniform(-5, 5, (5000, 3)).astype(np.float32),
                  tiny=rng.uniform(-1, 1, (7, 3)).astype(np.float32), nanx=nanx, eleven=rng.uniform(-1, 1, (11, 3)).astype(np.float32),
                  geo=geo, geo_neg=geo_neg, geo_mix=geo_mix)
    differ_default = 0
    for name, c in clouds.items():
        t = _oracle.kd_oracle(c)
        tr = _oracle.kd_ref(c)
        pts = c[~np.isnan(c[:, 0])]
        qs = np.concatenate([pts[rng.integers(0, len(pts), 24)].astype(np.float64),
                             pts[rng.integers(0, len(pts), 24)].astype(np.float64) + 0.5,
                             rng.uniform(-6, 6, (16, 3))])
        if name.startswith("geo"):   # (the shifted queries above are meaningless at 3^39: query at the points and between them)
            qs = np.concatenate([pts.astype(np.float64)[:40], 0.5 * (pts[:-1] + pts[1:]).astype(np.float64)[:24]])
        kd = KdBatch(1, len(c))
        assert lib.amk_kd_set_tie_order(kd.h, 1) == 0 and lib.amk_kd_set_tie_order(kd.h, 7) == capi.AMK_ERR_UNSUPPORTED
        kd.build(torch.from_numpy(c[None].copy()).cuda())
        nn = np.zeros(1, np.int32)
        assert lib.amk__kd_exact_nodes(kd.h, nn.ctypes.data_as(C.c_void_p)) == 0
        lib_o = _oracle.load_oracle(); lib_o.kdo_num_nodes.restype = C.c_int; lib_o.kdo_num_nodes.argtypes = [C.c_void_p]
        assert nn[0] == lib_o.kdo_num_nodes(t.h), (name, nn[0], lib_o.kdo_num_nodes(t.h))
        for k in (1, 3, 8, 10):
            r = kd.search(torch.from_numpy(qs[None].copy()).cuda(), k)
            torch.cuda.synchronize()
            idx, d2, cnt = (r[n].cpu().numpy()[0] for n in ("indices", "sqdist", "counts"))
            pt = r["pts"].cpu().numpy()[0]
            for i, q in enumerate(qs):
                ia, da, pa = t.search(q, k)                       # KDTreeTwo::SearchForNearest semantics, traversal tie order
                assert cnt[i] == len(ia), (name, k, i)
                assert np.array_equal(idx[i][:cnt[i]], ia), (name, k, i, idx[i], ia)
                assert np.array_equal(d2[i][:cnt[i]].view(np.int64), da.view(np.int64))
                assert np.array_equal(pt[i][:cnt[i]], pa)
                if tr is not None:
                    ir = tr.search(q, k)[0]
                    assert np.array_equal(idx[i][:cnt[i]], ir), (name, k, i)
                differ_default += not np.array_equal(t.bruteforce(q, k)[0][:len(ia)], ia)
        kd.close()
    assert differ_default > 50     # the default (lowest-index) policy does differ on these clouds: the mode matters


def test_nanoflann_tie_order_large_and_quantised(torch_cuda, oracle):
    """The device-built reference tree at BASELINE sizes -- every path of the build (block-wide top levels, the work queue, LDS
    windows, register-resident subtrees): 200 k continuous points, 50 k points on a 5 cm lattice (exact ties and duplicates), a
    20 k-point line; three orders of the same cloud per batch.  Node counts and traversal results equal the oracle's."""
    import ctypes as C
    torch = torch_cuda
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import KdBatch
    lib = capi.load()
    lib_o = _oracle.load_oracle(); lib_o.kdo_num_nodes.restype = C.c_int; lib_o.kdo_num_nodes.argtypes = [C.c_void_p]
    rng = np.random.default_rng(3)
    for n, kind in ((200000, "continuous"), (50000, "lattice"), (20000, "line")):
        c = synth.make_cloud(n, 11)[0]
        if kind == "lattice":
            c = (np.round(c * 20) / 20).astype(np.float32)
        if kind == "line":
            c[:, 1] = 0.5; c[:, 2] = 1.0
        cl = np.stack([c, c[::-1].copy(), c[rng.permutation(n)]])
        S = len(cl)
        kd = KdBatch(S, n); kd.set_tie_order(1)
        kd.build(torch.from_numpy(cl).cuda())
        nn = np.zeros(S, np.int32)
        assert lib.amk__kd_exact_nodes(kd.h, nn.ctypes.data_as(C.c_void_p)) == 0
        qs = np.concatenate([c[rng.integers(0, n, 24)].astype(np.float64), rng.uniform(-5, 25, (24, 3))])
        r = kd.search(torch.from_numpy(np.stack([qs] * S)).cuda(), 8)
        torch.cuda.synchronize()
        idx, d2 = r["indices"].cpu().numpy(), r["sqdist"].cpu().numpy()
        for s in range(S):
            t = _oracle.kd_oracle(cl[s])
            assert nn[s] == lib_o.kdo_num_nodes(t.h), (n, kind, s, nn[s])
            for i, q in enumerate(qs):
                ia, da, _ = t.search(q, 8)
                assert np.array_equal(idx[s, i][:len(ia)], ia), (n, kind, s, i)
                assert np.array_equal(d2[s, i][:len(ia)].view(np.int64), da.view(np.int64))
        kd.close()


def test_pair_build_equals_two_builds(torch_cuda, oracle):
    """amk_kd_build_pair (FrameKDMap::AddVertex's two InitializeNew calls as one launch, grid.y = tree): the obstacle and
    the edge index answer exactly like two separately built ones -- ragged counts, NaN-x points, 16-byte stride."""
    torch = torch_cuda
    from avoid_mpc_amd.host import KdBatch, kd_build_pair
    rng = np.random.default_rng(21)
    S, n, ne = 6, 3000, 400
    cl = np.zeros((S, n, 4), np.float32); ed = np.zeros((S, ne, 4), np.float32)
    cn = np.array([3000, 2999, 17, 0, 1024, 2500], np.int32); en = np.array([400, 0, 9, 1, 399, 64], np.int32)
    for s in range(S):
        c, e = synth.make_cloud(n, 500 + s)
        cl[s, :, :3] = c; ed[s, :len(e[:ne]), :3] = e[:ne]
    cl[0, ::13, 0] = np.nan
    kd_o, kd_e, ref_o, ref_e = KdBatch(S, n), KdBatch(S, ne), KdBatch(S, n), KdBatch(S, ne)
    d_cl, d_ed = torch.from_numpy(cl).cuda(), torch.from_numpy(ed).cuda()
    d_cn, d_en = torch.from_numpy(cn).cuda(), torch.from_numpy(en).cuda()
    kd_build_pair(kd_o, d_cl, kd_e, d_ed, d_cn, d_en)
    ref_o.build(d_cl, d_cn); ref_e.build(d_ed, d_en)
    qs = torch.from_numpy(np.stack([rng.uniform(0, 20, (S, 16)), rng.uniform(-4, 4, (S, 16)), rng.uniform(0, 3, (S, 16))], -1)).cuda()
    for a, b, k in ((kd_o, ref_o, 8), (kd_e, ref_e, 1), (kd_e, ref_e, 3)):
        ra, rb = a.search(qs, k), b.search(qs, k)
        torch.cuda.synchronize()
        assert np.array_equal(a.sizes(), b.sizes())
        for key in ("indices", "sqdist", "pts", "counts"):
            assert torch.equal(ra[key], rb[key]), key
    # and against the oracle for one scene of each tree
    t = _oracle.kd_oracle(cl[0, :cn[0], :3])
    r = kd_o.search(qs, 8)
    for q in range(16):
        ia, da, _ = t.search(qs[0, q].cpu().numpy(), 8)
        assert np.array_equal(r["indices"][0, q, :len(ia)].cpu().numpy(), ia)
        assert np.array_equal(r["sqdist"][0, q, :len(ia)].cpu().numpy(), da)


def test_exact_tree_is_never_used_stale(torch_cuda, oracle):
    """NANOFLANN mode, build; back to LOWEST_INDEX, rebuild with ANOTHER cloud; NANOFLANN again WITHOUT a build: the
    tree of the first cloud must not be walked over the second cloud's points -- until the next build the bucketed index
    answers (lowest-index order), afterwards nanoflann's order again."""
    torch = torch_cuda
    from avoid_mpc_amd.host import KdBatch
    g1 = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    g2 = (g1[::-1] * 0.5 + 3.0).astype(np.float32).copy()
    kd = KdBatch(1, len(g1))
    q = np.array([[[3.5, 3.5, 3.5], [4.25, 4.25, 4.25], [0.1, 7.0, 2.0]]])
    dq = torch.from_numpy(q).cuda()
    kd.set_tie_order(1); kd.build(torch.from_numpy(g1[None].copy()).cuda())
    kd.set_tie_order(0); kd.build(torch.from_numpy(g2[None].copy()).cuda())
    kd.set_tie_order(1)
    r = kd.search(dq, 4); torch.cuda.synchronize()
    t2 = _oracle.kd_oracle(g2)
    for i in range(3):
        ib, db = t2.bruteforce(q[0, i], 4)                      # lowest index among equal distances
        assert np.array_equal(r["sqdist"][0, i].cpu().numpy(), db)
        assert np.array_equal(r["indices"][0, i].cpu().numpy(), ib)
    kd.build(torch.from_numpy(g2[None].copy()).cuda())          # now the tree belongs to g2
    r = kd.search(dq, 4); torch.cuda.synchronize()
    for i in range(3):
        ia, da, _ = t2.search(q[0, i], 4)
        assert np.array_equal(r["indices"][0, i].cpu().numpy(), ia) and np.array_equal(r["sqdist"][0, i].cpu().numpy(), da)


def test_exact_build_gives_up_cleanly_when_the_ring_of_open_nodes_is_full(torch_cuda, oracle):
    """the ring of open nodes of exact_build_rest used to advance q_tail BEFORE it checked the capacity; on overflow two
    positions stayed unwritten and the consumer that claimed one spun for ever (reachable beyond ~430 k points per scene).  With
    the ring shrunk to 2 / 4 / 16 entries (amk__exact_set_queue_cap, tests only) a 50 k-point build must END: every scene either
    has the reference's tree (same node count, nanoflann's order) or reports it unavailable (-1) and is answered by the bucketed
    index (lowest-index order, same distances)."""
    import ctypes as C
    torch = torch_cuda
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import KdBatch
    lib = capi.load()
    lib_o = _oracle.load_oracle(); lib_o.kdo_num_nodes.restype = C.c_int; lib_o.kdo_num_nodes.argtypes = [C.c_void_p]
    n = 50000
    cl = np.stack([synth.make_cloud(n, 31 + s)[0] for s in range(4)])
    rng = np.random.default_rng(5)
    qs = np.concatenate([cl[0][rng.integers(0, n, 8)].astype(np.float64), rng.uniform(-5, 25, (8, 3))])
    assert lib.amk__exact_set_queue_cap(3) == capi.AMK_ERR_INVALID_ARG and lib.amk__exact_set_queue_cap(4096) == capi.AMK_ERR_INVALID_ARG
    gave_up = 0
    try:
        for cap in (2, 4, 16, 0):
            assert lib.amk__exact_set_queue_cap(cap) == 0
            kd = KdBatch(len(cl), n); kd.set_tie_order(1)
            kd.build(torch.from_numpy(cl).cuda())
            nn = np.zeros(len(cl), np.int32)
            assert lib.amk__kd_exact_nodes(kd.h, nn.ctypes.data_as(C.c_void_p)) == 0     # synchronises: the build ended
            status = kd.exact_status().cpu().numpy()                                      # the public account of the same fact
            assert np.array_equal(status, np.where(nn < 0, capi.AMK_EXACT_GAVE_UP, capi.AMK_EXACT_IN_USE)), (cap, nn, status)
            r = kd.search(torch.from_numpy(np.stack([qs] * len(cl))).cuda(), 8)
            torch.cuda.synchronize()
            idx, d2 = r["indices"].cpu().numpy(), r["sqdist"].cpu().numpy()
            for s in range(len(cl)):
                t = _oracle.kd_oracle(cl[s])
                assert nn[s] == -1 or nn[s] == lib_o.kdo_num_nodes(t.h), (cap, s, nn[s])
                gave_up += int(nn[s] == -1)
                for i, q in enumerate(qs):
                    ia, da, _ = t.search(q, 8) if nn[s] >= 0 else (*t.bruteforce(q, 8), None)
                    assert np.array_equal(d2[s, i].view(np.int64), da.view(np.int64)), (cap, s, i)
                    assert np.array_equal(idx[s, i], ia), (cap, s, i)
            if cap == 0:
                assert (nn >= 0).all()                                                    # the shipped capacity holds a 50 k-point tree
            kd.close()
    finally:
        lib.amk__exact_set_queue_cap(0)
    assert gave_up > 0, "a 2-entry ring must overflow on 50 k points: the give-up path was not exercised"


def test_exact_status_tells_when_the_reference_tie_order_does_not_hold(torch_cuda, oracle):
    """amk_kd_exact_status: AMK_TIES_NANOFLANN used to degrade silently.  Off / before a build: -1; an ordinary
    cloud: 0; a geometric point sequence in front of a tight cluster (x_i = 2^-i, i < 100, then 1000 points within 2^-110 of the
    origin: planeSplit's mid-range cut peels a point or two per level while the cluster stays one big node, so the tree is > 48
    levels deep above an ordinary subtree): 2; a bare geometric sequence (the deep part lies inside one of the build's subtree
    windows, whose own stack it exceeds): 1.  In every case the answers carry the right distances (bucketed fallback)."""
    torch = torch_cuda
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import KdBatch
    n = 1100
    rng = np.random.default_rng(3)
    deep = np.zeros((n, 3), np.float32)
    deep[:100, 0] = 2.0 ** -np.arange(100, dtype=np.float64)
    deep[100:] = (rng.uniform(0.0, 1.0, (n - 100, 3)) * 2.0 ** -110).astype(np.float32)
    ordinary = synth.make_cloud(n, 5)[0]
    geo = np.full((n, 3), np.nan, np.float32); geo[:120] = 0.0; geo[:120, 0] = 2.0 ** -np.arange(120, dtype=np.float64)   # 120 points, NaN-x padding
    cl = np.stack([deep, ordinary, geo])
    kd = KdBatch(3, n)
    assert (kd.exact_status().cpu().numpy() == capi.AMK_EXACT_OFF).all()
    kd.set_tie_order(capi.AMK_TIES_NANOFLANN)
    assert (kd.exact_status().cpu().numpy() == capi.AMK_EXACT_OFF).all()          # no build since the switch: the bucketed index answers
    kd.build(torch.from_numpy(cl).cuda())
    st = kd.exact_status().cpu().numpy()
    print("exact status of (deep top, ordinary, bare geometric):", st.tolist())
    assert st[1] == capi.AMK_EXACT_IN_USE and st[0] in (capi.AMK_EXACT_TOO_DEEP, capi.AMK_EXACT_GAVE_UP) and st[2] in (capi.AMK_EXACT_TOO_DEEP, capi.AMK_EXACT_GAVE_UP), st
    assert capi.AMK_EXACT_TOO_DEEP in st.tolist(), st                              # the depth account is exercised, not only the give-up flag
    qs = np.array([[0.0, 0.0, 0.0], [1e-30, 0.0, 0.0], [0.3, 0.1, 0.0], [2.0, 0.0, 0.0]])
    r = kd.search(torch.from_numpy(np.stack([qs] * 3)).cuda(), 5)
    torch.cuda.synchronize()
    for s_ in range(3):
        c = cl[s_][~np.isnan(cl[s_][:, 0])]
        t = _oracle.kd_oracle(c)
        for i, q in enumerate(qs):
            ia, da = t.bruteforce(q, 5)
            assert np.array_equal(r["sqdist"][s_, i].cpu().numpy().view(np.int64), da.view(np.int64)), (s_, i)
    h = np.zeros(3, np.int32)
    assert capi.load().amk_kd_exact_status_host(kd.h, h.ctypes.data) == 0 and h.tolist() == st.tolist()
    kd.close()
