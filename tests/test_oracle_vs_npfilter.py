"""CPU: the C oracle's prefilter against the independent numpy statement (tests/npfilter.py)."""
import numpy as np
import pytest

import npfilter


@pytest.mark.parametrize("scene,n,W,H,pose", [("room_shell", 60_000, 320, 240, 11), ("uniform_box", 90_000, 160, 128, 500),
                                              ("room_shell", 200_000, 640, 360, 777),  # H % 16 != 0: tail rows
                                              ("uniform_box", 3_000, 64, 48, 3)])       # sparse: mostly empty frame
def test_filter_matches_numpy_model(orc, pkg, scene, n, W, H, pose):
    xyzw, rgba = orc.generate(scene, 0xC0FFEE02, 0, n, n)
    r = orc.project(xyzw, rgba, pkg.orbit_projection(pose, W, H), W, H)
    a = orc.filter(r["depth_bits"], r["img"])
    b = npfilter.apply_filter(r["depth_bits"], r["img"])
    assert np.array_equal(a["mask"], b["mask"])
    assert np.array_equal(a["minmax"], b["minmax"])
    assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
    assert np.array_equal(a["img"], b["img"])
    assert np.array_equal(a["tensor"], b["tensor"])
    assert 0 < (a["mask"] > 0).sum() < W * H


def test_filter_left_top_border_weights(orc):
    """A11 at x = 0 / y = 0 (project_cloud.cu:149-155): x0 is clamped BEFORE wx = inX - x0, so
    the weight is -0.25 with x0 == x1.  A frame whose left and top borders are empty forces the
    in-place fill there; both statements must agree bit for bit (including non-finite fills)."""
    rng = np.random.default_rng(4)
    d = rng.uniform(1.0, 3.0, size=(64, 64)).astype(np.float32).view(np.uint32).copy()
    d[:, :12] = 0x7F7FFFFF
    d[:10, :] = 0x7F7FFFFF
    d[40:44, 30:50] = np.float32(9.0).view(np.uint32)
    img = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    a = orc.filter(d, img)
    b = npfilter.apply_filter(d, img)
    for k in ("mask", "img", "tensor", "minmax"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["depth"].view(np.uint32), b["depth"].view(np.uint32))
