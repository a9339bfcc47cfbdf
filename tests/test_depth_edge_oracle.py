"""CPU: the BuildEdgeCloud restatement in oracle/depth_oracle.c (FrameKDMap.cpp:176-214: 8-bit quantisation, 3x3 erode,
Canny(0.1, 0.3), back-projection) against scipy.ndimage for the erosion and the Sobel gradients, an independent numpy
non-maximum suppression, and hand-checkable frames.  OpenCV itself is not in the image: parity with cv::erode / cv::Canny
is unpinned; what is pinned is their published algorithm."""
import numpy as np
import pytest
from scipy import ndimage

from tests import _oracle
from tests.test_depth_oracle import YAML, np_process, scene


def np_edges(eroded):
    e = eroded.astype(np.int64)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]); ky = kx.T
    dx = ndimage.correlate(e, kx, mode="nearest"); dy = ndimage.correlate(e, ky, mode="nearest")   # BORDER_REPLICATE
    mag = np.abs(dx) + np.abs(dy)
    H, W = e.shape
    mp = np.zeros((H + 2, W + 2), np.int64); mp[1:-1, 1:-1] = mag
    out = np.zeros((H, W), np.uint8)
    for r in range(H):
        for c in range(W):
            m = mag[r, c]
            if m <= 0:
                continue
            x, y = abs(int(dx[r, c])), abs(int(dy[r, c])) << 15
            t22 = x * 13573
            R, Cc = r + 1, c + 1
            if y < t22:
                ok = m > mp[R, Cc - 1] and m >= mp[R, Cc + 1]
            elif y > t22 + (x << 16):
                ok = m > mp[R - 1, Cc] and m >= mp[R + 1, Cc]
            else:
                s = -1 if (int(dx[r, c]) ^ int(dy[r, c])) < 0 else 1
                ok = m > mp[R - 1, Cc - s] and m > mp[R + 1, Cc + s]
            out[r, c] = 255 if ok else 0
    return out


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
@pytest.mark.parametrize("shape,scale", [((480, 640), 10.0), ((97, 131), 4.0), ((60, 80), 2.5)])
def test_pipeline_against_scipy_and_numpy(dtype, shape, scale):
    rng = np.random.default_rng(3)
    # piecewise-smooth depth: a far wall with two near boxes and holes -> real depth edges
    rows, cols = shape
    d = np.full(shape, 20.0) + rng.normal(0, 0.02, shape)
    d[rows // 4: rows // 2, cols // 5: cols // 2] = 4.0
    d[rows // 2: rows - 5, cols // 2: cols - 9] = 9.0
    d[rng.random(shape) < 0.02] = 0.0
    img = np.round(d * 1000).astype(np.uint16) if dtype == np.uint16 else d.astype(np.float32)
    prm = dict(YAML, pixel2meter=1e-3 if dtype == np.uint16 else 1.0, resize_scale=scale,
               Tbc=np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.0], [0, -1, 0, 0.05], [0, 0, 0, 1.0]]))
    Twc = np.array([[0.8, -0.6, 0, 2.0], [0.6, 0.8, 0, -1.0], [0, 0, 1, 1.5], [0, 0, 0, 1.0]])
    cloud, quant, eroded, edges = _oracle.depth_edge_oracle(img, prm, Twc)
    _, inv = np_process(img, prm, np.eye(4))
    rng_d = prm["depth_max"] - prm["depth_min"]
    with np.errstate(divide="ignore"):
        q = np.where(inv.astype(np.float64) > 1e-2,
                     ((np.float32(1) / inv).astype(np.float64) / rng_d * 200.0).astype(np.int64), 255).astype(np.uint8)
    assert np.array_equal(quant, q)
    assert np.array_equal(eroded, ndimage.minimum_filter(q, size=3, mode="constant", cval=255))
    assert np.array_equal(edges, np_edges(eroded)) and 0 < (edges > 0).sum() < edges.size // 2
    # back-projection of the edge pixels at the quantised depth through Twc * Tbc
    M = Twc @ np.asarray(prm["Tbc"])
    s = scale
    pts = []
    for r, c in zip(*np.nonzero(edges)):
        dd = float(np.float32(eroded[r, c])) * rng_d / 200.0
        if dd > prm["depth_max"] or dd < prm["depth_min"]:
            continue
        pts.append((M @ np.array([(c - prm["cx"] / s) * dd / (prm["fx"] / s), (r - prm["cy"] / s) * dd / (prm["fy"] / s), dd, 1.0]))[:3])
    pts = np.array(pts)
    assert len(cloud) == len(pts) > 0 and np.allclose(cloud, pts, rtol=0, atol=np.abs(pts).max() * 2e-7)


def test_a_step_edge_is_one_pixel_wide_and_flat_frames_have_none():
    img = np.full((40, 60), 10.0, np.float32)
    img[:, 30:] = 20.0
    prm = dict(YAML, resize_scale=1.0, fx=30.0, fy=30.0, cx=30.0, cy=20.0)
    cloud, quant, eroded, edges = _oracle.depth_edge_oracle(img, prm, np.eye(4))
    assert quant[0, 0] == int(10.0 / 99.9 * 200) and quant[0, 59] == int(20.0 / 99.9 * 200)
    cols = np.nonzero(edges.any(axis=0))[0]
    assert len(cols) == 1 and np.all(edges[:, cols[0]] == 255)         # one column of edge pixels, every row
    assert np.all(np.abs(cloud[:, 2] - eroded[0, cols[0]] * 99.9 / 200.0) < 1e-5)   # quantised depth, camera = world
    flat, _, _, e2 = _oracle.depth_edge_oracle(np.full((40, 60), 10.0, np.float32), prm, np.eye(4))
    assert len(flat) == 0 and not e2.any()
    none, _, _, _ = _oracle.depth_edge_oracle(np.zeros((40, 60), np.float32), prm, np.eye(4))   # empty obstacle cloud
    assert len(none) == 0
