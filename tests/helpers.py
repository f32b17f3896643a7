"""Shared test helpers (inputs only; no reference or product logic)."""
import numpy as np


def kat_P(orc):
    """SURVEY.md 8c KAT camera: P = K * I, fx = fy = 100, cx = 32, cy = 24 (64x48)."""
    K = np.array([[100.0, 0, 32.0], [0, 100.0, 24.0], [0, 0, 1.0]])
    return orc.compose_projection(K, np.eye(4))


def cloud(points, colors):
    xyz = np.asarray(points, dtype=np.float32).reshape(-1, 3)
    rgb = np.asarray(colors, dtype=np.uint8).reshape(-1, 3)
    xyzw = np.concatenate([xyz, np.ones((len(xyz), 1), np.float32)], axis=1)
    rgba = np.concatenate([rgb, np.full((len(rgb), 1), 255, np.uint8)], axis=1)
    return np.ascontiguousarray(xyzw), np.ascontiguousarray(rgba)


def random_cloud(n, seed, lo=(-4, -1.5, -4), hi=(4, 1.5, 4)):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    return cloud(xyz, rgb)


def f32(x):
    return np.float32(x)


def bits(x):
    return int(np.float32(x).view(np.uint32))
