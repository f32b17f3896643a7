"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py): the oracle
must keep reproducing them (CPU), and the HIP path must match them (GPU) -- the fixtures
travel to the GPU box, the reference does not."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _load(path):
    z = np.load(path)  # allow_pickle=False (default): plain arrays only
    return {k: z[k] for k in z.files}


def test_fixtures_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(orc, path):
    g = _load(path)
    W, H = int(g["W"]), int(g["H"])
    r = orc.project(g["xyz"], g["rgb"], g["P"], W, H)
    assert np.array_equal(r["depth_bits"], g["depth_bits"]) and np.array_equal(r["acc"], g["acc"])
    assert np.array_equal(r["img"], g["img"])
    f = orc.filter(r["depth_bits"], r["img"])
    assert np.array_equal(f["depth"].view(np.uint32), g["f_depth_bits"]) and np.array_equal(f["img"], g["f_img"])
    assert np.array_equal(f["mask"], g["f_mask"]) and np.array_equal(f["tensor"], g["f_tensor"])
    assert np.array_equal(f["minmax"], g["f_minmax"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(pkg, projector, path, mode):
    g = _load(path)
    W, H = int(g["W"]), int(g["H"])
    projector.set_option("mode", mode)
    try:
        projector.upload_points(g["xyz"], g["rgb"])  # tight xyz / rgb strides (12 B / 3 B)
        projector.set_resolution(W, H)
        img, depth = projector.project(g["P"])
        assert np.array_equal(depth.view(np.uint32), g["depth_bits"]) and np.array_equal(img, g["img"])
        img_f, depth_f = projector.project(g["P"], filtered=True)
        assert np.array_equal(depth_f.view(np.uint32), g["f_depth_bits"]) and np.array_equal(img_f, g["f_img"])
        assert np.array_equal(projector.download(pkg._lib.BUF_MASK), g["f_mask"])
        assert np.array_equal(projector.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), g["f_tensor"])
        assert np.array_equal(projector.download(pkg._lib.BUF_MINMAX), g["f_minmax"])
    finally:
        projector.set_option("mode", 1)
