"""CPU: hypothesis-driven properties of the oracle and the host logic (small cases, adversarial
floats): what the domain guarantees independently of any particular cloud."""
import numpy as np
from hypothesis import given, settings, strategies as st

import pymodel
from helpers import cloud

f32 = st.floats(width=32, allow_nan=False, allow_infinity=False, min_value=-1e6, max_value=1e6)
f32_wild = st.floats(width=32, allow_nan=True, allow_infinity=True)
SET = dict(max_examples=60, deadline=None)


@settings(**SET)
@given(pts=st.lists(st.tuples(f32_wild, f32_wild, f32_wild), min_size=1, max_size=40), pose=st.integers(0, 999))
def test_projection_agrees_with_exact_model_on_any_float(orc, pkg, pts, pose):
    """Every float32 triple -- huge, tiny, denormal, inf, nan -- projects to the same pixel and
    depth bits in the C oracle and in the exact-rational model."""
    W, H = 320, 200
    P = pkg.orbit_projection(pose, W, H)
    for x, y, z in pts:
        pix, b = orc.project_point(P, float(np.float32(x)), float(np.float32(y)), float(np.float32(z)), W, H)
        mp, md = pymodel.project_point(P, x, y, z, W, H)
        assert pix == mp
        if pix >= 0:
            assert b == int(np.float32(md).view(np.uint32))


@settings(**SET)
@given(data=st.data())
def test_any_split_merges_to_the_same_frame(orc, pkg, data):
    """k-way point split + element-wise MIN (depth) / SUM (accumulators) == the unsplit frame,
    for arbitrary cut positions: the specification of the multi-GPU exchange (render.cu:81,125-128)."""
    n = data.draw(st.integers(1, 300))
    seed = data.draw(st.integers(0, 2 ** 31))
    rng = np.random.default_rng(seed)
    xyz = rng.uniform((-1.5, -1.0, 0.5), (1.5, 1.0, 4.0), size=(n, 3)).astype(np.float32)
    xyz[rng.integers(0, n, n // 3)] = xyz[0]  # pile points up on shared pixels
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz, rgb)
    W, H = 48, 32
    K = np.array([[40.0, 0, 24], [0, 40.0, 16], [0, 0, 1]])
    P = orc.compose_projection(K, np.eye(4))
    ref = orc.project(xyzw, rgba, P, W, H)
    cuts = sorted(data.draw(st.lists(st.integers(0, n), min_size=0, max_size=5)))
    bounds = [0] + cuts + [n]
    depth = np.full(W * H, orc.EMPTY_DEPTH, np.uint32)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        d, _ = orc.clear(W, H)
        depth = np.minimum(depth, orc.min_depth_pass(xyzw[lo:hi], P, W, H, d)) if hi > lo else depth
    acc = np.zeros(W * H * 4, np.uint32)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        if hi > lo:
            _, a = orc.clear(W, H)
            acc += orc.accumulate_pass(xyzw[lo:hi], rgba[lo:hi], P, W, H, depth, a)
    assert np.array_equal(depth.reshape(H, W), ref["depth_bits"])
    assert np.array_equal(acc.reshape(H, W, 4), ref["acc"])
    assert np.array_equal(orc.resolve(acc, W, H), ref["img"])


@settings(**SET)
@given(q=st.tuples(f32, f32, f32, f32).filter(lambda t: sum(abs(v) for v in t) > 1e-3),
       t=st.tuples(f32, f32, f32))
def test_trajectory_text_formats_roundtrip_any_pose(pkg, tmp_path_factory, q, t):
    F = pkg.formats
    d = tmp_path_factory.mktemp("traj")
    E = np.eye(4)
    E[:3, :3] = F.quat_to_rot(*q)
    E[:3, 3] = t
    F.write_images_txt(d / "images.txt", [E])
    (back, _), = F.read_trajectory_colmap(d / "images.txt")
    assert np.allclose(back, E, rtol=1e-9, atol=1e-9 * (1 + np.abs(E).max()))
    F.write_trajectory_tum(d / "t.txt", [E])
    back, = F.read_trajectory_tum(d / "t.txt")
    assert np.allclose(back, E, rtol=1e-7, atol=1e-7 * (1 + np.abs(E).max()))


@settings(**SET)
@given(v=st.floats(width=32, allow_nan=False, allow_infinity=True))
def test_f16_conversion_any_float(orc, v):
    with np.errstate(over="ignore"):
        assert orc.f32_to_f16(float(np.float32(v))) == int(np.float32(v).astype(np.float16).view(np.uint16))
