"""CPU: known-answer tests pinning the oracle (hand-derived from the reference lines;
SURVEY.md 8c items 1-9).  The reference ships no fixtures of its own: PARITY UNPINNED
beyond these.  Camera: P = K * I, fx = fy = 100, cx = 32, cy = 24, 64 x 48."""
import numpy as np
import pytest

from helpers import bits, cloud, kat_P

W, H = 64, 48
EMPTY = 0x7F7FFFFF


def test_contract_selftest(orc):
    assert orc.lib().orc_selftest() == 1


def test_kat1_single_point(orc):
    # render.cu:62-81: r = (32*2, 24*2, 2) -> u = 32, v = 24, depth bits of 2.0f
    P = kat_P(orc)
    xyzw, rgba = cloud([(0, 0, 2)], [(11, 22, 33)])
    r = orc.project(xyzw, rgba, P, W, H)
    assert orc.project_point(P, 0, 0, 2, W, H) == (24 * W + 32, 0x40000000)
    d = r["depth_bits"]
    assert d[24, 32] == 0x40000000 and (d != EMPTY).sum() == 1
    assert tuple(r["img"][24, 32]) == (11, 22, 33)
    assert r["img"].sum() == 66 and tuple(r["acc"][24, 32]) == (11, 22, 33, 1)


def test_kat2_z_cull(orc):
    P = kat_P(orc)
    assert orc.project_point(P, 0, 0, 0.0, W, H)[0] == -1      # render.cu:63  r.z <= 0
    assert orc.project_point(P, 0, 0, -1.0, W, H)[0] == -1
    assert orc.project_point(P, 0, 0, float("nan"), W, H)[0] == -1
    pix, b = orc.project_point(P, 0, 0, 1e-30, W, H)           # tiny positive z is kept
    assert pix == 24 * W + 32 and b == bits(1e-30)


@pytest.mark.parametrize("x,u", [(-7.875, 0), (-7.625, 2), (-7.375, 2), (-8.125, 0), (7.875, None)])
def test_kat3_ties_round_half_even(orc, x, u):
    # z = 25: r.x = 100 x + 800 is exact (12.5, 37.5, 62.5, -12.5, 1587.5); the quotient
    # lands on k + 0.5 and rintf (render.cu:65) rounds half to even: 0.5->0, 1.5->2,
    # 2.5->2, -0.5->-0 (u = 0 accepted), 63.5->64 (u >= W culled, render.cu:68)
    P = kat_P(orc)
    pix, _ = orc.project_point(P, x, 0.0, 25.0, W, H)
    if u is None:
        assert pix == -1
    else:
        assert pix == 24 * W + u


def test_kat4_blend_within_window(orc):
    P = kat_P(orc)
    xyzw, rgba = cloud([(0, 0, 1.00), (0, 0, 1.01)], [(10, 100, 201), (21, 101, 100)])
    r = orc.project(xyzw, rgba, P, W, H)
    assert r["depth_bits"][24, 32] == bits(1.0)
    assert tuple(r["img"][24, 32]) == (15, 100, 150)  # floor((c1 + c2) / 2), render.cu:160-162
    xyzw, rgba = cloud([(0, 0, 1.00), (0, 0, 1.03)], [(10, 100, 201), (21, 101, 100)])
    r = orc.project(xyzw, rgba, P, W, H)
    assert tuple(r["img"][24, 32]) == (10, 100, 201) and r["acc"][24, 32, 3] == 1


def test_kat5_window_boundary(orc):
    # render.cu:106: reject iff depth > min + 0.02f (fp32 add, strict compare)
    P = kat_P(orc)
    edge = np.float32(1.0) + np.float32(0.02)
    above = np.nextafter(edge, np.float32(np.inf), dtype=np.float32)
    xyzw, rgba = cloud([(0, 0, 1.0), (0, 0, edge)], [(0, 0, 0), (200, 200, 200)])
    assert orc.project(xyzw, rgba, P, W, H)["acc"][24, 32, 3] == 2
    xyzw, rgba = cloud([(0, 0, 1.0), (0, 0, above)], [(0, 0, 0), (200, 200, 200)])
    assert orc.project(xyzw, rgba, P, W, H)["acc"][24, 32, 3] == 1


def test_kat6_many_points_one_pixel(orc):
    P = kat_P(orc)
    xyzw, rgba = cloud([(0, 0, 2)] * 300, [(255, 255, 255)] * 300)
    r = orc.project(xyzw, rgba, P, W, H)
    assert tuple(r["acc"][24, 32]) == (76500, 76500, 76500, 300)
    assert tuple(r["img"][24, 32]) == (255, 255, 255)


def test_kat7_shuffle_and_shard_merge(orc, pkg):
    n = 10_000
    xyzw, rgba = orc.generate("uniform_box", 99, 0, n, n)
    P = pkg.orbit_projection(17, W, H)
    r = orc.project(xyzw, rgba, P, W, H)
    perm = np.random.default_rng(1).permutation(n)
    r2 = orc.project(xyzw[perm], rgba[perm], P, W, H)
    assert all(np.array_equal(r[k], r2[k]) for k in r)
    # k-way shard + element-wise min / sum merge == 1-way: the specification of the RCCL path
    for k in (2, 3, 8):
        depth, _ = orc.clear(W, H)
        parts = [(n * i // k, n * (i + 1) // k) for i in range(k)]
        ds = []
        for lo, hi in parts:
            d, _ = orc.clear(W, H)
            ds.append(orc.min_depth_pass(xyzw[lo:hi], P, W, H, d))
        depth = np.minimum.reduce(ds)
        acc = np.zeros(W * H * 4, np.uint32)
        for lo, hi in parts:
            _, a = orc.clear(W, H)
            acc += orc.accumulate_pass(xyzw[lo:hi], rgba[lo:hi], P, W, H, depth, a)
        assert np.array_equal(depth.reshape(H, W), r["depth_bits"])
        assert np.array_equal(acc.reshape(H, W, 4), r["acc"])
        assert np.array_equal(orc.resolve(acc, W, H), r["img"])


def _plane(w, h, depth=2.0, colour=(128, 64, 255)):
    d = np.full((h, w), np.float32(depth), np.float32).view(np.uint32).copy()
    img = np.empty((h, w, 3), np.uint8)
    img[:] = colour
    return d, img


def test_kat8_filter_leak_and_empty_quad(orc):
    # 32x32 flat plane at 2 m, one background pixel leaking through at 3 m and one empty
    # 2x2 quad.  Every pooled level is 2 m, the Laplacian is 0, so the test at each level is
    # d <= 2 * 1.025 (project_cloud.cu:119): plane kept, 3 m leak masked; empty pixels are
    # masked by project_cloud.cu:97.  Masked -> depth -1, colour 0 (project_cloud.cu:168-179).
    d, img = _plane(32, 32)
    d[12, 10] = bits(3.0)
    d[20:22, 20:22] = EMPTY
    img[20:22, 20:22] = 0
    f = orc.filter(d, img)
    exp = np.full((32, 32), 255, np.uint8)
    exp[12, 10] = 0
    exp[20:22, 20:22] = 0
    assert np.array_equal(f["mask"], exp)
    assert f["depth"][12, 10] == -1.0 and (f["depth"][20:22, 20:22] == -1.0).all()
    assert (f["depth"][exp == 255] == 2.0).all()
    assert (f["img"][exp == 0] == 0).all() and (f["img"][exp == 255] == (128, 64, 255)).all()
    # min / max are taken BEFORE masking (project_cloud.cu:375-380): the leak widens the range
    assert list(f["minmax"]) == [bits(2.0), bits(3.0)]


def test_kat9_tensor_encoding(orc):
    # project_cloud.cu:181-185: ch_k = half(float(half(u8)) / 255), ch3 = 1, ch4 = half(half(d - min) / range)
    d, img = _plane(32, 32, 2.0, (128, 0, 255))
    d[5, 5] = bits(4.0)       # a kept far pixel needs a close coarse parent: make a 4 m patch
    d[0:16, 16:32] = bits(4.0)
    f = orc.filter(d, img)
    t = f["tensor"]
    assert t.shape == (5, 32, 32)
    assert t[0, 0, 0] == 0x3804  # half(128/255)
    assert t[1, 0, 0] == 0x0000 and t[2, 0, 0] == 0x3C00 and t[3, 0, 0] == 0x3C00
    assert f["mask"][0, 0] == 255 and t[4, 0, 0] == 0x0000      # at min depth -> 0
    assert f["mask"][3, 20] == 255 and t[4, 3, 20] == 0x3C00    # at max depth -> 1
    assert f["mask"][5, 5] == 0 and t[4, 5, 5] == 0xBC00        # masked -> -1
    assert t[0, 5, 5] == 0 and t[3, 5, 5] == 0


def test_f16_conversion_matches_numpy(orc):
    rng = np.random.default_rng(0)
    xs = (rng.standard_normal(5000) * 10.0 ** rng.integers(-9, 6, 5000)).astype(np.float32)
    xs = np.concatenate([xs, np.array([0, -0.0, 65504, 65519.99, 65520, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25,
                                       1.5 * 2.0 ** -25, np.inf, -np.inf, 6.1e-5, 5.96e-8], np.float32)])
    with np.errstate(over="ignore"):
        ref = xs.astype(np.float16).view(np.uint16)
    got = np.array([orc.f32_to_f16(float(v)) for v in xs], np.uint16)
    assert np.array_equal(ref, got)
    assert orc.f32_to_f16(float("nan")) == 0x7E00
    for h in (0x0001, 0x03FF, 0x0400, 0x3C00, 0x7BFF, 0xBC00, 0x8001):
        assert orc.f16_to_f32(h) == float(np.array([h], np.uint16).view(np.float16)[0])


def test_filter_rejects_bad_width(orc):
    d, img = _plane(40, 32)  # 40 % 16 != 0: the reference's pyramid strides break (quirk Q3)
    with pytest.raises(ValueError):
        orc.filter(d, img)


def test_filter_tail_rows(orc):
    # H = 40 -> H_eff = 32: rows 32..39 skip the pyramid test, mask = non-empty
    d, img = _plane(32, 40)
    d[35, 3] = EMPTY
    d[36, 4] = bits(9.0)  # would be masked inside the filter domain; kept in the tail
    f = orc.filter(d, img)
    assert f["mask"][35, 3] == 0 and f["depth"][35, 3] == -1.0
    assert f["mask"][36, 4] == 255 and f["depth"][36, 4] == 9.0
    assert list(f["minmax"]) == [bits(2.0), bits(2.0)]  # tail rows do not enter min / max
