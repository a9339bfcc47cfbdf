"""CPU: oracle/depth_oracle.c (restatement of FrameKDMap::ProcessDepth, AM/src/FrameKDMap.cpp:75-138) against an
independent numpy restatement and against hand-checkable cases.  Parity with OpenCV's own resize / Eigen's products is
unpinned (neither is in the image; see the oracle's header) -- what is pinned here is the published INTER_LINEAR
coordinate rule, the range gates and the back-projection."""
import numpy as np
import pytest

from tests import _oracle

YAML = dict(pixel2meter=1.0, depth_min=0.1, depth_max=100.0, resize_scale=10.0, fx=320.0, fy=320.0, cx=320.0, cy=240.0)


def np_process(depth, prm, Twb):
    """numpy twin (float32 where the reference holds float, float64 elsewhere)."""
    rows, cols = depth.shape
    s = prm["resize_scale"]
    W, H = int(cols / s), int(rows / s)
    d = (depth.astype(np.float32).astype(np.float64) * prm["pixel2meter"]).astype(np.float32)
    bad = (d.astype(np.float64) < prm["depth_min"]) | (d.astype(np.float64) > prm["depth_max"])
    with np.errstate(divide="ignore"):
        inv = np.where(bad, np.float32(0), (1.0 / d.astype(np.float64)).astype(np.float32)).astype(np.float32)

    def taps(n_dst, n_src):
        scale = 1.0 / (n_dst / n_src)
        f = (((np.arange(n_dst) + 0.5) * scale) - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        w = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0; w[lo] = 0; i0[lo] = 0
        hi = i0 >= n_src - 1; w[hi] = 0; i0[hi] = n_src - 1
        return i0, np.minimum(i0 + 1, n_src - 1), w
    x0, x1, ax = taps(W, cols)
    y0, y1, ay = taps(H, rows)
    a0, b0 = (np.float32(1) - ax), (np.float32(1) - ay)
    t0 = (inv[y0][:, x0] * a0[None, :] + inv[y0][:, x1] * ax[None, :]).astype(np.float32)
    t1 = (inv[y1][:, x0] * a0[None, :] + inv[y1][:, x1] * ax[None, :]).astype(np.float32)
    small = (t0 * b0[:, None] + t1 * ay[:, None]).astype(np.float32)
    M = np.asarray(Twb, np.float64).reshape(4, 4) @ np.asarray(prm.get("Tbc", np.eye(4)), np.float64).reshape(4, 4)
    pts = []
    for row in range(H):
        for col in range(W):
            invd = float(small[row, col])
            if invd < 1e-2:
                continue
            dd = 1.0 / invd
            if not (prm["depth_min"] < dd < prm["depth_max"]):
                continue
            pc = np.array([(col - prm["cx"] / s) * dd / (prm["fx"] / s), (row - prm["cy"] / s) * dd / (prm["fy"] / s), dd, 1.0])
            pts.append((M @ pc)[:3])
    return np.array(pts, np.float64).reshape(-1, 3), small


def scene(rng, rows, cols, dtype):
    d = rng.uniform(0.5, 30.0, (rows, cols))
    d[rng.random((rows, cols)) < 0.1] = 0.0          # holes
    d[rng.random((rows, cols)) < 0.05] = 150.0       # beyond depth_max
    if dtype == np.uint16:
        return np.round(d * 1000).astype(np.uint16), 1e-3
    return d.astype(np.float32), 1.0


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
@pytest.mark.parametrize("shape,scale", [((480, 640), 10.0), ((97, 131), 4.0), ((48, 64), 1.0), ((60, 80), 2.5)])
def test_c_equals_numpy(dtype, shape, scale):
    rng = np.random.default_rng(hash((shape, scale)) % 2**32)
    img, p2m = scene(rng, *shape, dtype)
    prm = dict(YAML, pixel2meter=p2m, resize_scale=scale,
               Tbc=np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.0], [0, -1, 0, 0.05], [0, 0, 0, 1.0]]))
    th = 0.3
    Twb = np.array([[np.cos(th), -np.sin(th), 0, 3.0], [np.sin(th), np.cos(th), 0, -1.0], [0, 0, 1, 1.5], [0, 0, 0, 1]])
    cloud, inv = _oracle.depth_oracle(img, prm, Twb)
    pts, small = np_process(img, prm, Twb)
    assert inv.shape == small.shape and np.array_equal(inv, small)          # float taps: same bits
    assert len(cloud) == len(pts) > 0
    # numpy's @ uses BLAS accumulation (possibly FMA): compare as float64 values, then one float32 ulp
    assert np.allclose(cloud, pts, rtol=0, atol=np.abs(pts).max() * 2e-7)


def test_constant_plane_back_projects_to_the_pinhole_model():
    """Depth 2 m everywhere, identity poses: x = (u - cx') * 2 / fx', y likewise, z = 2; W*H points in row-major order."""
    img = np.full((480, 640), 2.0, np.float32)
    cloud, inv = _oracle.depth_oracle(img, YAML, np.eye(4))
    assert inv.shape == (48, 64) and np.all(inv == np.float32(0.5)) and len(cloud) == 48 * 64
    u, v = np.meshgrid(np.arange(64), np.arange(48))
    exp = np.stack([(u - 32.0) * 2.0 / 32.0, (v - 24.0) * 2.0 / 32.0, np.full(u.shape, 2.0)], -1).reshape(-1, 3)
    assert np.array_equal(cloud, exp.astype(np.float32))


def test_range_gates_and_interpolation_with_holes():
    """A hole (depth 0 -> inverse depth 0) next to a valid pixel pulls the interpolated inverse depth down, i.e. the
    point farther away, exactly as cv::resize on the inverse-depth image does (FrameKDMap.cpp:99-109)."""
    img = np.full((20, 20), 1.0, np.float32)
    img[:, 10:] = 0.0                              # right half invalid
    prm = dict(YAML, resize_scale=2.0, fx=20.0, fy=20.0, cx=10.0, cy=10.0)
    cloud, inv = _oracle.depth_oracle(img, prm, np.eye(4))
    # INTER_LINEAR at scale 2 samples source x = 2*dx + 0.5: taps (2dx, 2dx+1) with weights (.5, .5)
    assert np.all(inv[:, :5] == 1.0) and np.all(inv[:, 5:] == 0.0)
    assert len(cloud) == 10 * 5 and np.all(cloud[:, 2] == 1.0)
    img[:, 9] = 0.0                                # now the tap pair (8, 9) straddles the hole: inv = 0.5 -> depth 2
    cloud, inv = _oracle.depth_oracle(img, prm, np.eye(4))
    assert np.all(inv[:, 4] == 0.5) and np.all(cloud.reshape(10, 5, 3)[:, 4, 2] == 2.0)
    # inverse depth below 1e-2 (depth > 100 m) is dropped even though the raw pixel passed the gate
    far = np.full((20, 20), 100.0, np.float32)
    cloud, _ = _oracle.depth_oracle(far, prm, np.eye(4))
    assert len(cloud) == 0                         # 1/100 -> float 0.01 < 1e-2 in double? (float(0.01) = 0.00999999977)
    empty, _ = _oracle.depth_oracle(np.zeros((20, 20), np.uint16), prm, np.eye(4))
    assert len(empty) == 0


def test_downscaled_inverse_depth_equals_torch_bilinear_interpolate():
    """Independent pin of the cv::resize restatement:
    torch.nn.functional.interpolate(mode="bilinear", align_corners=False) implements the same half-pixel rule as
    cv::resize's INTER_LINEAR (source coordinate (d + 0.5) * scale - 0.5, clamped at the borders).  The two differ
    only in where they round (torch forms the four-tap sum in a different order): <= 2 ulp of the float32 inverse depth
    when the size ratio is an integer (the reference's scale 10, mpc_parameters.yaml:63).  For a non-integer ratio
    torch evaluates the source coordinate in float32 where OpenCV (and the oracle) use double before the cast: the tap
    weights then differ by ~1e-5 and so does the result (bounded here at 1e-4 relative)."""
    import torch
    rng = np.random.default_rng(12)
    for (rows, cols, s, dtype) in ((480, 640, 10.0, np.float32), (120, 160, 4.0, np.uint16), (96, 130, 3.0, np.float32)):
        depth, p2m = scene(rng, rows, cols, dtype)
        prm = dict(YAML, resize_scale=s, pixel2meter=p2m)
        _, small = _oracle.depth_oracle(depth, prm, np.eye(4))
        d = (depth.astype(np.float32).astype(np.float64) * p2m).astype(np.float32)
        bad = (d.astype(np.float64) < prm["depth_min"]) | (d.astype(np.float64) > prm["depth_max"])
        with np.errstate(divide="ignore"):
            inv = np.where(bad, np.float32(0), (1.0 / d.astype(np.float64)).astype(np.float32)).astype(np.float32)
        H, W = int(rows / s), int(cols / s)
        t = torch.nn.functional.interpolate(torch.from_numpy(inv)[None, None], size=(H, W), mode="bilinear",
                                            align_corners=False)[0, 0].numpy()
        assert small.shape == t.shape
        ulp = np.spacing(np.maximum(np.abs(small), np.abs(t)).astype(np.float32))
        diff = np.abs(small.astype(np.float64) - t.astype(np.float64))
        if cols / s == int(cols / s) and rows / s == int(rows / s):
            assert np.all(diff <= 2.0 * ulp + 1e-12), float((diff / np.maximum(ulp, 1e-30)).max())
        else:
            assert np.all(diff <= 1e-4 * np.maximum(np.abs(t), 1e-3))
