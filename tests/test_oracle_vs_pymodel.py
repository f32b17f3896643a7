"""CPU: the C oracle against the exact-rational Python model (tests/pymodel.py)."""
import numpy as np

import pymodel
from helpers import bits, kat_P


def _check(orc, P, pts, W, H):
    n_in = 0
    for x, y, z in pts:
        pix, b = orc.project_point(P, float(x), float(y), float(z), W, H)
        mp, md = pymodel.project_point(P, x, y, z, W, H)
        assert pix == mp, (x, y, z, pix, mp)
        if pix >= 0:
            assert b == bits(md)
            n_in += 1
    return n_in


def test_projection_matches_exact_model_random(orc, pkg):
    rng = np.random.default_rng(11)
    W, H = 1920, 1080
    total_in = 0
    for k in (0, 333, 771):
        P = pkg.orbit_projection(k, W, H)
        pts = rng.uniform((-4, -1.5, -4), (4, 1.5, 4), size=(700, 3)).astype(np.float32)
        total_in += _check(orc, P, pts, W, H)
    assert total_in > 50


def test_projection_matches_exact_model_near_half_pixel_boundaries(orc, pkg):
    """Points constructed to sit within a few ulp of a half-pixel boundary, where one
    rounding more or less flips the pixel (hard part 1)."""
    W, H = 640, 480
    P = pkg.orbit_projection(40, W, H)
    E = pkg.orbit_pose(40)
    K = pkg.benchmark_calibration(W, H).getIntrinsicsMatrix()
    Rinv, t = E[:3, :3].T, E[:3, 3]
    rng = np.random.default_rng(3)
    pts = []
    for _ in range(400):
        u = rng.integers(0, W) + 0.5
        v = rng.integers(0, H) + rng.uniform(-0.4, 0.4)
        z = rng.uniform(0.5, 6.0)
        cam = np.array([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z])
        w = Rinv @ (cam - t)
        base = w.astype(np.float32)
        for d in (-2, -1, 0, 1, 2):  # step x by a few ulps across the boundary
            p = base.copy()
            p[0] = np.float32(p[0]) + d * np.spacing(np.float32(p[0]))
            pts.append(p)
    assert _check(orc, P, pts, W, H) > 500


def test_projection_matches_exact_model_specials(orc):
    P = kat_P(orc)
    pts = [(0, 0, 2), (0, 0, 1e-30), (1e-38, 0, 1e-38), (-7.875, 0, 25), (7.875, 0, 25), (0, 0, 3e38),
           (1e30, 0, 1e30), (0.3, -0.2, 1e-45), (-0.32, 0.0, 1.0), (0, 0, 1.17549435e-38)]
    _check(orc, P, pts, 64, 48)
