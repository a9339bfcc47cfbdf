"""GPU: amk_depth_to_cloud (FrameKDMap::ProcessDepth) against oracle/depth_oracle.c:
identical float32 bits, identical order and count, for both pixel types; and the produced cloud handed straight to
the KD build answers queries like the oracle tree built from the oracle cloud."""
import numpy as np
import pytest

from tests import _oracle
from tests.test_depth_oracle import YAML, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _poses(rng, S):
    out = np.zeros((S, 4, 4))
    for s in range(S):
        th = rng.uniform(-np.pi, np.pi)
        out[s] = [[np.cos(th), -np.sin(th), 0, rng.uniform(-5, 5)], [np.sin(th), np.cos(th), 0, rng.uniform(-5, 5)],
                  [0, 0, 1, rng.uniform(0.5, 3)], [0, 0, 0, 1]]
    return out


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
@pytest.mark.parametrize("shape,scale,stride", [((480, 640), 10.0, 3), ((97, 131), 4.0, 4), ((48, 64), 1.0, 3),
                                                ((60, 80), 2.5, 3)])
def test_bit_exact_vs_oracle(dtype, shape, scale, stride, torch_cuda):
    torch = torch_cuda
    from avoid_mpc_amd.host import depth_params, depth_to_cloud
    rng = np.random.default_rng(11)
    S = 5
    imgs, p2m = zip(*[scene(rng, *shape, dtype) for _ in range(S)])
    imgs = np.stack(imgs)
    imgs[S - 1] = 0                                   # an empty frame
    Tbc = np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.0], [0, -1, 0, 0.05], [0, 0, 0, 1.0]])
    prm = dict(YAML, pixel2meter=p2m[0], resize_scale=scale, Tbc=Tbc)
    Twb = _poses(rng, S)
    dev = torch.from_numpy(imgs.view(np.int16) if dtype == np.uint16 else imgs).cuda()
    cloud, counts = depth_to_cloud(dev, depth_params(**prm), torch.from_numpy(Twb).cuda(), point_stride=stride)
    torch.cuda.synchronize()
    cloud, counts = cloud.cpu().numpy(), counts.cpu().numpy()
    for s in range(S):
        ref, _ = _oracle.depth_oracle(imgs[s], prm, Twb[s])
        assert counts[s] == len(ref), (s, counts[s], len(ref))
        assert np.array_equal(cloud[s, :counts[s], :3].view(np.uint32), ref.view(np.uint32))
    assert counts[S - 1] == 0 and counts[:S - 1].min() > 0


def test_cloud_feeds_the_kd_build(torch_cuda):
    torch = torch_cuda
    from avoid_mpc_amd.host import KdBatch, depth_params, depth_to_cloud
    rng = np.random.default_rng(5)
    S = 3
    imgs = np.stack([scene(rng, 480, 640, np.float32)[0] for _ in range(S)])
    Twb = _poses(rng, S)
    cloud, counts = depth_to_cloud(torch.from_numpy(imgs).cuda(), depth_params(**YAML), torch.from_numpy(Twb).cuda())
    kd = KdBatch(S, cloud.shape[1])
    kd.build(cloud, counts)
    q = rng.uniform(-5, 5, (S, 4, 3))
    out = kd.search(torch.from_numpy(q).cuda(), 8)
    torch.cuda.synchronize()
    idx, d2 = out["indices"].cpu().numpy(), out["sqdist"].cpu().numpy()
    for s in range(S):
        ref, _ = _oracle.depth_oracle(imgs[s], YAML, Twb[s])
        tree = _oracle.kd_oracle(ref)
        for j in range(4):
            ri, rd, _ = tree.search(q[s, j], 8)
            assert len(ri) == 8 and np.array_equal(idx[s, j], ri) and np.array_equal(d2[s, j], rd)


def test_host_api_and_errors(torch_cuda):
    import ctypes as C
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import depth_params
    lib = capi.load()
    rng = np.random.default_rng(2)
    img, _ = scene(rng, 60, 80, np.uint16)
    prm = dict(YAML, pixel2meter=1e-3, resize_scale=2.0, fx=40.0, fy=40.0, cx=40.0, cy=30.0)
    p = depth_params(**prm)
    w, h = C.c_int(), C.c_int()
    assert lib.amk_depth_out_size(60, 80, 2.0, C.byref(w), C.byref(h)) == capi.AMK_OK and (w.value, h.value) == (40, 30)
    cloud = np.zeros((1200, 3), np.float32); cnt = np.zeros(1, np.int32); Twb = np.eye(4)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    st = lib.amk_depth_to_cloud_host(vp(img), capi.AMK_DEPTH_U16, 60, 80, 4800, 1, C.byref(p), vp(Twb), vp(cloud), 3, 3600, vp(cnt))
    assert st == capi.AMK_OK
    ref, _ = _oracle.depth_oracle(img, prm, Twb)
    assert cnt[0] == len(ref) and np.array_equal(cloud[:cnt[0]], ref)
    assert lib.amk_depth_to_cloud_host(vp(img), 7, 60, 80, 4800, 1, C.byref(p), vp(Twb), vp(cloud), 3, 3600, vp(cnt)) == capi.AMK_ERR_UNSUPPORTED
    assert lib.amk_depth_to_cloud_host(None, 0, 60, 80, 4800, 1, C.byref(p), vp(Twb), vp(cloud), 3, 3600, vp(cnt)) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_depth_out_size(60, 80, 0.0, C.byref(w), C.byref(h)) == capi.AMK_ERR_INVALID_ARG


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
@pytest.mark.parametrize("shape,scale,stride", [((480, 640), 10.0, 3), ((97, 131), 4.0, 4), ((60, 80), 2.5, 3)])
def test_edge_cloud_bit_exact_vs_oracle(dtype, shape, scale, stride, torch_cuda):
    """amk_depth_to_edge_cloud (FrameKDMap::BuildEdgeCloud, row f3) against the oracle: same points, same order."""
    torch = torch_cuda
    from avoid_mpc_amd.host import depth_params, depth_to_cloud
    rng = np.random.default_rng(4)
    S = 5
    rows, cols = shape
    imgs = []
    for s in range(S):
        d = np.full(shape, 20.0) + rng.normal(0, 0.02, shape)
        r0, c0 = rng.integers(2, rows // 2), rng.integers(2, cols // 2)
        d[r0:r0 + rows // 3, c0:c0 + cols // 3] = rng.uniform(2, 8)
        d[rng.random(shape) < 0.02] = 0.0
        imgs.append(np.round(d * 1000).astype(np.uint16) if dtype == np.uint16 else d.astype(np.float32))
    imgs = np.stack(imgs)
    imgs[S - 1] = 0                                   # no obstacle point -> no edge cloud either
    Tbc = np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.0], [0, -1, 0, 0.05], [0, 0, 0, 1.0]])
    prm = dict(YAML, pixel2meter=1e-3 if dtype == np.uint16 else 1.0, resize_scale=scale, Tbc=Tbc)
    Twc = _poses(rng, S)
    dev = torch.from_numpy(imgs.view(np.int16) if dtype == np.uint16 else imgs).cuda()
    cloud, counts = depth_to_cloud(dev, depth_params(**prm), torch.from_numpy(Twc).cuda(), point_stride=stride, edge=True)
    torch.cuda.synchronize()
    cloud, counts = cloud.cpu().numpy(), counts.cpu().numpy()
    for s in range(S):
        ref = _oracle.depth_edge_oracle(imgs[s], prm, Twc[s])[0]
        assert counts[s] == len(ref), (s, counts[s], len(ref))
        assert np.array_equal(cloud[s, :counts[s], :3].view(np.uint32), ref.view(np.uint32))
    assert counts[S - 1] == 0 and counts[:S - 1].min() > 10


def test_edge_cloud_size_limit(torch_cuda):
    import ctypes as C
    from avoid_mpc_amd import capi
    from avoid_mpc_amd.host import depth_params
    lib = capi.load()
    img = np.ones((480, 640), np.float32)
    p = depth_params(**dict(YAML, resize_scale=2.0))           # 320 x 240 = 76800 > AMK_EDGE_MAX_PIXELS
    cloud = np.zeros((76800, 3), np.float32); cnt = np.zeros(1, np.int32); T = np.eye(4)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    st = lib.amk_depth_to_edge_cloud_host(vp(img), capi.AMK_DEPTH_F32, 480, 640, 480 * 640, 1, C.byref(p), vp(T), vp(cloud), 3,
                                          76800 * 3, vp(cnt))
    assert st == capi.AMK_ERR_UNSUPPORTED
