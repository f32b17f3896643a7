"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Bars (BASELINE.json north_star): pixel indices / depth bits / colour bytes / mask / fp16
tensor bits identical (this implies the <= 1e-5 float-depth criterion).
"""
import numpy as np
import pytest

from helpers import cloud, kat_P, random_cloud

pytestmark = pytest.mark.gpu


def _check_frame(pkg, orc, projector, xyzw, rgba, P, W, H, filtered=True):
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    img, depth = projector.project(P)
    ref = orc.project(xyzw, rgba, P, W, H)
    assert np.array_equal(depth.view(np.uint32), ref["depth_bits"])
    assert np.array_equal(img, ref["img"])
    projector.set_option("keep_accum", 1)
    projector.render(P)
    assert np.array_equal(projector.download(pkg._lib.BUF_ACCUM), ref["acc"])
    projector.set_option("keep_accum", 0)
    assert np.max(np.abs(depth[ref["depth_bits"] != orc.EMPTY_DEPTH] -
                         ref["depth_bits"].view(np.float32)[ref["depth_bits"] != orc.EMPTY_DEPTH]), initial=0) <= 1e-5
    if filtered:
        img_f, depth_f = projector.project(P, filtered=True)
        rf = orc.filter(ref["depth_bits"], ref["img"])
        assert np.array_equal(projector.download(pkg._lib.BUF_MASK), rf["mask"])
        assert np.array_equal(depth_f.view(np.uint32), rf["depth"].view(np.uint32))
        assert np.array_equal(img_f, rf["img"])
        assert np.array_equal(projector.download(pkg._lib.BUF_MINMAX), rf["minmax"])
        assert np.array_equal(projector.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
    return ref


@pytest.fixture(params=[1, 2, 3, 0], ids=["tile", "tile-split", "tile-packed", "two-pass"])
def mode(request, projector):
    """tile-split: the binned form with the split threshold lowered to 64 entries, so that every
    non-trivial tile of an ordinary test frame takes the several-workgroups-per-tile path.
    tile-packed: option pack = 2, i.e. the point kernel reads the packed coordinates whatever the
    cloud (the default only packs clouds that shrink by 1/8), and every upload is decoded and
    compared with the SoA arrays on the device."""
    projector.set_option("mode", 1 if request.param else 0)
    if request.param == 2:
        projector.set_option("split_threshold", 64)
        projector.set_option("split_slice", 48)
    projector.set_option("pack", 2 if request.param == 3 else 1)
    projector.set_option("keep_accum", 1)
    yield request.param
    projector.set_option("pack", 1)
    projector.set_option("mode", 1)
    projector.set_option("split_threshold", 32768)
    projector.set_option("split_slice", 16384)
    projector.set_option("keep_accum", 0)


@pytest.mark.parametrize("scene", ["uniform_box", "room_shell"])
@pytest.mark.parametrize("W,H", [(640, 480), (1920, 1080), (64, 48)])
def test_scene_frame_parity(pkg, orc, projector, mode, scene, W, H):
    n = 200_000
    xyzw, rgba = orc.generate(scene, 0xC0FFEE01, 0, n, n)
    for k in (0, 137, 500):
        ref = _check_frame(pkg, orc, projector, xyzw, rgba, pkg.orbit_projection(k, W, H), W, H)
    assert (ref["depth_bits"] != orc.EMPTY_DEPTH).sum() > 0


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 255, 257, 1023, 4099])
def test_ragged_point_counts(pkg, orc, projector, mode, n):
    xyzw, rgba = random_cloud(n, seed=n + 1)
    _check_frame(pkg, orc, projector, xyzw, rgba, pkg.orbit_projection(10, 64, 48), 64, 48)


def test_kat_points(pkg, orc, projector, mode):
    P = kat_P(orc)
    pts = [(0, 0, 2), (0, 0, 0), (0, 0, -1), (0, 0, 1e-30), (-0.32, 0, 1.0), (-0.325, 0, 1.0), (0.315, 0, 1.0),
           (0.0, 0.0, 1.0), (0.0, 0.0, 1.01), (0.0, 0.0, 1.03)]
    cols = [(10 * i, 20, 255 - i) for i in range(len(pts))]
    xyzw, rgba = cloud(pts, cols)
    _check_frame(pkg, orc, projector, xyzw, rgba, P, 64, 48)


def test_collisions_one_pixel(pkg, orc, projector, mode):
    """300 identical points in one pixel plus a far crowd behind them (atomic contention)."""
    P = kat_P(orc)
    pts = [(0, 0, 2.0)] * 300 + [(0, 0, 2.015)] * 200 + [(0, 0, 2.5)] * 100
    cols = [(255, 255, 255)] * 300 + [(1, 2, 3)] * 200 + [(9, 9, 9)] * 100
    xyzw, rgba = cloud(pts, cols)
    ref = _check_frame(pkg, orc, projector, xyzw, rgba, P, 64, 48)
    assert ref["acc"][24, 32, 3] == 500


def test_shuffle_and_shard_invariance(pkg, orc, projector, mode):
    n, W, H = 100_000, 640, 480
    xyzw, rgba = orc.generate("room_shell", 7, 0, n, n)
    P = pkg.orbit_projection(42, W, H)
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    img0, d0 = projector.project(P)
    perm = np.random.default_rng(0).permutation(n)
    projector.upload_points(xyzw[perm], rgba[perm])
    img1, d1 = projector.project(P)
    assert np.array_equal(img0, img1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))


def test_generator_matches_oracle(pkg, orc, projector):
    for scene in ("uniform_box", "room_shell"):
        total = 1_000_003
        for first, count in ((0, 5000), (500_000, 4097), (total - 333, 333)):
            projector.generate_synthetic(scene, 0xC0FFEE03, first, count, total)
            xyzw, rgba = projector.download_points()
            rx, rc = orc.generate(scene, 0xC0FFEE03, first, count, total)
            assert np.array_equal(xyzw.view(np.uint32), rx.view(np.uint32)), scene
            assert np.array_equal(rgba, rc), scene


def test_phase_calls_equal_whole_frame(pkg, orc, projector, mode):
    n, W, H = 50_000, 320, 240
    xyzw, rgba = orc.generate("uniform_box", 11, 0, n, n)
    P = pkg.orbit_projection(5, W, H)
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    img0, d0 = projector.project(P)
    projector.clear()
    projector.min_depth_pass(P)
    projector.accumulate_pass(P)
    projector.resolve()
    projector.synchronize()
    assert np.array_equal(projector.download(pkg._lib.BUF_IMAGE), img0)
    assert np.array_equal(projector.download(pkg._lib.BUF_DEPTH), d0.view(np.uint32))


def test_project_cloud_mirror(pkg, orc):
    n, W, H = 20_000, 160, 128
    xyzw, rgba = orc.generate("room_shell", 3, 0, n, n)
    pc = pkg.ProjectCloud(xyzw, rgba)
    cal = pkg.benchmark_calibration(W, H)
    E = pkg.orbit_pose(77)
    color = np.empty((H, W, 3), np.uint8)
    depth = np.empty((H, W), np.float32)
    assert pc.computeRGBD(cal, E, None, None) == -1
    assert pc.computeRGBD(cal, E, color, depth) == 1
    ref = orc.project(xyzw, rgba, orc.compose_projection(cal.getIntrinsicsMatrix(), E), W, H)
    assert np.array_equal(color, ref["img"]) and np.array_equal(depth.view(np.uint32), ref["depth_bits"])
    d2 = np.empty((H, W), np.float32)
    assert pc.computeRGBD(cal, E, None, d2) == 1  # depth-only use (cloudreader.cpp:246)
    assert np.array_equal(d2, depth)
    assert pc.computeFilteredRGBD(cal, E, color, depth) == 1
    rf = orc.filter(ref["depth_bits"], ref["img"])
    assert np.array_equal(color, rf["img"]) and np.array_equal(depth.view(np.uint32), rf["depth"].view(np.uint32))
    pc.projector.close()


def test_errors(pkg, projector):
    projector.set_resolution(100, 50)  # W % 16 != 0
    P = pkg.orbit_projection(0, 100, 50)
    with pytest.raises(pkg.RtrError) as e:
        projector.project(P, filtered=True)
    assert e.value.code == pkg._lib.RTR_ERR_UNSUPPORTED
    with pytest.raises(pkg.RtrError) as e:
        projector.project(P, want_img=False, want_depth=False)
    assert e.value.code == pkg._lib.RTR_ERR_NO_OUTPUT


def test_two_shards_with_external_min_sum(pkg, orc):
    """The multi-GPU frame sequence on ONE GPU: two contexts own the two halves of the cloud,
    the exchange steps are done on zero-copy torch views of their device buffers (the same
    views bench.py hands to RCCL).  Result must equal the single-context frame."""
    import torch
    n, W, H = 120_000, 640, 480
    xyzw, rgba = orc.generate("room_shell", 21, 0, n, n)
    P = pkg.orbit_projection(300, W, H)
    ref = orc.project(xyzw, rgba, P, W, H)
    reff = orc.filter(ref["depth_bits"], ref["img"])
    for mode in (1, 0):
        locs = []
        for r in range(2):
            lo, hi = pkg.shard_range(n, r, 2)
            p = pkg.Projector(0)
            p.set_option("mode", mode)
            p.upload_points(xyzw[lo:hi], rgba[lo:hi])
            p.set_resolution(W, H)
            loc = pkg.sharded.HipLocal(p)
            loc.bind_stream()
            locs.append(loc)
        for loc in locs:
            loc.clear()
            loc.min_depth_pass(P)
        d = torch.minimum(locs[0].depth_tensor(), locs[1].depth_tensor())
        for loc in locs:
            loc.depth_tensor().copy_(d)
            loc.accumulate_pass(P)
        a = locs[0].accum_tensor() + locs[1].accum_tensor()
        for loc in locs:
            loc.accum_tensor().copy_(a)
            loc.resolve()
            loc.filter()
        torch.cuda.synchronize()
        for loc in locs:
            assert np.array_equal(loc.p.download(pkg._lib.BUF_ACCUM), ref["acc"])
            assert np.array_equal(loc.p.download(pkg._lib.BUF_IMAGE), reff["img"])
            assert np.array_equal(loc.p.download(pkg._lib.BUF_DEPTH), reff["depth"].view(np.uint32))
            assert np.array_equal(loc.p.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), reff["tensor"])
            loc.p.close()


def test_4k_frame_uses_wide_tiles(pkg, orc, projector, mode):
    """3840x2160 (config C5 resolution): 64x32 tiles in the binned mode, H % 16 == 0."""
    n, W, H = 300_000, 3840, 2160
    xyzw, rgba = orc.generate("room_shell", 0xC0FFEE05, 0, n, n)
    _check_frame(pkg, orc, projector, xyzw, rgba, pkg.orbit_projection(7, W, H), W, H)


def test_odd_resolution_projection_only(pkg, orc, projector, mode):
    """W, H not multiples of the tile or of 4: ragged tiles and unaligned image rows."""
    n, W, H = 80_000, 333, 131
    xyzw, rgba = orc.generate("uniform_box", 4, 0, n, n)
    _check_frame(pkg, orc, projector, xyzw, rgba, pkg.orbit_projection(11, W, H), W, H, filtered=False)


def test_hot_pixel_and_hot_tile(pkg, orc, projector, mode):
    """Half a million points in ONE pixel plus a dense 20x20 pixel patch: one workgroup owns
    the whole load in the binned mode, the atomics pile up on one address in mode 0."""
    rng = np.random.default_rng(9)
    n_hot, n_patch = 500_000, 300_000
    hot = np.tile(np.array([[0.0, 0.0, 2.0]], np.float32), (n_hot, 1))
    hot[:, 2] += rng.uniform(0, 0.05, n_hot).astype(np.float32)  # straddles the 2 cm window
    patch = np.stack([rng.uniform(0.1, 0.3, n_patch), rng.uniform(0.1, 0.3, n_patch),
                      rng.uniform(1.9, 2.1, n_patch)], axis=1).astype(np.float32)
    xyz = np.concatenate([hot, patch])
    rgb = rng.integers(0, 256, size=(len(xyz), 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz, rgb)
    ref = _check_frame(pkg, orc, projector, xyzw, rgba, kat_P(orc), 64, 48)
    assert ref["acc"][24, 32, 3] > 100_000


def test_reorder_and_chunk_culling_keep_the_frame(pkg, orc, projector):
    """rtr_reorder_points (Morton sort) and option "cull" (skip 256-point chunks whose box is
    outside the frustum) are pure speed-ups: frames stay bit-identical, in every pose."""
    n, W, H = 400_000, 1920, 1080
    for scene in ("room_shell", "uniform_box"):
        xyzw, rgba = orc.generate(scene, 77, 0, n, n)
        projector.upload_points(xyzw, rgba)
        projector.set_resolution(W, H)
        refs = {}
        for k in (0, 130, 610, 875):
            P = pkg.orbit_projection(k, W, H)
            ref = orc.project(xyzw, rgba, P, W, H)
            refs[k] = (P, ref, orc.filter(ref["depth_bits"], ref["img"]))
        try:
            for step in ("cull", "reorder+cull", "reorder"):
                if step == "reorder+cull":
                    projector.reorder_points()
                    back, cols = projector.download_points()
                    key = lambda a, b: np.lexsort(np.concatenate([a.view(np.uint32), b.view(np.uint8)], axis=1).T)  # noqa: E731
                    assert np.array_equal(back[key(back, cols)], xyzw[key(xyzw, rgba)])  # a permutation
                projector.set_option("cull", 0 if step == "reorder" else 1)
                for k, (P, ref, rf) in refs.items():
                    img, depth = projector.project(P)
                    assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]), (scene, step, k)
                    assert np.array_equal(img, ref["img"]), (scene, step, k)
                    img_f, depth_f = projector.project(P, filtered=True)
                    assert np.array_equal(depth_f.view(np.uint32), rf["depth"].view(np.uint32))
                    assert np.array_equal(projector.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
        finally:
            projector.set_option("cull", 0)


def test_culling_with_extreme_cameras(pkg, orc, projector):
    """Cameras inside / far outside the cloud, near-degenerate depths: the conservative box test
    must never drop a point the exact arithmetic keeps."""
    n, W, H = 200_000, 640, 480
    xyzw, rgba = orc.generate("room_shell", 5, 0, n, n)
    projector.upload_points(xyzw, rgba)
    projector.reorder_points()
    projector.set_resolution(W, H)
    projector.set_option("cull", 1)
    try:
        cal = pkg.benchmark_calibration(W, H)
        rng = np.random.default_rng(2)
        for trial in range(12):
            E = pkg.orbit_pose(int(rng.integers(0, 1000)))
            E[:3, 3] += rng.normal(scale=[0.01, 3.0, 30.0][trial % 3], size=3)
            if trial % 4 == 3:  # camera exactly on a wall: many points with r.z ~ 0
                E[:3, 3] = -E[:3, :3] @ np.array([-4.0, 0.2, 0.3])
            P = pkg.compose_projection(cal.getIntrinsicsMatrix(), E)
            ref = orc.project(xyzw, rgba, P, W, H)
            img, depth = projector.project(P)
            assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]), trial
            assert np.array_equal(img, ref["img"]), trial
    finally:
        projector.set_option("cull", 0)


@pytest.mark.parametrize("n_same", [255, 256, 257, 258, 259, 600, 5000])
def test_packed_accumulator_boundary(pkg, orc, projector, mode, n_same):
    """The tile kernel's packed 16-bit accumulator fields hold exactly 257 points of value 255
    (65535); one more must fall back to the wide layout.  All-255 colours are the worst case."""
    P = kat_P(orc)
    rng = np.random.default_rng(n_same)
    pts = [(0.0, 0.0, 2.0)] * n_same + [(float(x), float(y), 2.0) for x, y in rng.uniform(-0.3, 0.3, size=(400, 2))]
    cols = [(255, 255, 255)] * n_same + [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(400)]
    xyzw, rgba = cloud(pts, cols)
    ref = _check_frame(pkg, orc, projector, xyzw, rgba, P, 64, 48)
    assert ref["acc"][24, 32, 3] >= n_same and tuple(ref["img"][24, 32]) != (0, 0, 0)


@pytest.mark.parametrize("n_hot,threshold", [(300, 32768), (70_000, 0)])
def test_wide_accumulators_in_both_halves_of_a_tile(pkg, orc, projector, n_hot, threshold):
    """The whole-frame tile kernel keeps only the packed accumulators in LDS; a tile that needs the wide ones is redone
    in two halves of 16 rows.  Pixels that blend more than 257 points in BOTH halves of one 32x32 tile, ordinary pixels
    around them; with the split threshold off and more than 60000 entries the tile goes straight to the wide halves."""
    W, H = 64, 48
    rng = np.random.default_rng(n_hot)
    K = np.array([[100.0, 0, 32.0], [0, 100.0, 24.0], [0, 0, 1.0]])
    P = orc.compose_projection(K, np.eye(4))
    def at(u, v, k):  # k points that round to pixel (u, v), depths inside the blending window of the nearest
        z = rng.uniform(2.0, 2.0004, k)
        return np.stack([(u - 32.0) * z / 100.0, (v - 24.0) * z / 100.0, z], axis=1)
    spread = np.stack([rng.uniform(-0.6, 0.6, 3000), rng.uniform(-0.45, 0.45, 3000), rng.uniform(2.05, 2.3, 3000)], axis=1)  # (never in front of the hot pixels)
    xyz = np.concatenate([at(40, 5, n_hot), at(45, 20, n_hot // 2 + 300), spread]).astype(np.float32)  # tile (1, 0): rows 5 and 20
    rgb = np.concatenate([np.full((len(xyz) - 3000, 3), 255, np.uint8), rng.integers(0, 256, size=(3000, 3), dtype=np.uint8)])
    xyzw, rgba = cloud(xyz, rgb)
    projector.set_option("split_threshold", threshold)
    try:
        ref = _check_frame(pkg, orc, projector, xyzw, rgba, P, W, H)
        assert projector.frame_stats()["split_tiles"] == 0
    finally:
        projector.set_option("split_threshold", 32768)
    assert ref["acc"][5, 40, 3] > 257 and ref["acc"][20, 45, 3] > 257


def test_frames_larger_than_4k_fall_back(pkg, orc, projector):
    """More than 4096 screen tiles (beyond 3840x2160): the library silently uses the atomic
    form; the frame is still exact."""
    n, W, H = 150_000, 5120, 2880
    xyzw, rgba = orc.generate("room_shell", 8, 0, n, n)
    _check_frame(pkg, orc, projector, xyzw, rgba, pkg.orbit_projection(50, W, H), W, H)


def test_point_grid_is_per_context(pkg, orc, projector):
    """The tuning knob "point_grid" belongs to a context; changing it on one context must not
    disturb another (it used to be process-wide)."""
    n, W, H = 300_000, 640, 480
    xyzw, rgba = orc.generate("room_shell", 12, 0, n, n)
    P = pkg.orbit_projection(9, W, H)
    ref = orc.project(xyzw, rgba, P, W, H)
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    img, _ = projector.project(P)          # lists allocated for the default grid
    other = pkg.Projector(0)
    try:
        other.set_option("point_grid", 37)
        other.upload_points(xyzw, rgba)
        other.set_resolution(W, H)
        img_o, depth_o = other.project(P)
        img2, depth2 = projector.project(P)  # still the default grid here
        for i, d in ((img_o, depth_o), (img2, depth2)):
            assert np.array_equal(i, ref["img"]) and np.array_equal(d.view(np.uint32), ref["depth_bits"])
        assert np.array_equal(img, ref["img"])
    finally:
        other.close()


def test_adversarial_floats(pkg, orc, projector, mode):
    """Coordinates drawn from raw bit patterns (inf, nan, denormals, 1e38 ...) mixed with an
    ordinary cloud: every culling path (r.z sign, conservative frustum test, chunk boxes, exact
    arithmetic with its correctly rounded reciprocal) must agree with the oracle bit for bit."""
    rng = np.random.default_rng(123)
    n, W, H = 60_000, 640, 480
    wild = rng.integers(0, 2 ** 32, size=(n, 3), dtype=np.uint64).astype(np.uint32).view(np.float32)
    tame, _ = orc.generate("room_shell", 3, 0, n, n)
    xyz = np.where(rng.random((n, 1)) < 0.5, wild, tame[:, :3]).astype(np.float32)
    xyz[::7, 2] = np.float32(1e-41)   # denormal depths right at the camera plane
    xyz[::11, 0] = np.float32(3e38)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz, rgb)
    projector.set_option("cull", 1)
    try:
        for k in (0, 444):
            _check_frame(pkg, orc, projector, xyzw, rgba, pkg.orbit_projection(k, W, H), W, H)
        # identity camera: r.z = z, so the denormal / huge depths reach the division unchanged
        K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
        _check_frame(pkg, orc, projector, xyzw, rgba, orc.compose_projection(K, np.eye(4)), W, H)
    finally:
        projector.set_option("cull", 0)


def test_pack_option(pkg, orc):
    """Option "pack": the default packs a spatially ordered cloud (6-9 B/pt) and leaves a hash-ordered
    one alone; 0 never packs, 2 always does (and verifies the decode on the device); switching it
    re-packs the resident cloud at once.  Frames and rtr_download_points never change."""
    n, W, H = 300_000, 640, 480
    p = pkg.Projector(0)
    try:
        p.set_resolution(W, H)
        p.set_option("auto_reorder", 0)
        for scene, expect in (("room_shell", 1), ("uniform_box", 0)):
            xyzw, rgba = orc.generate(scene, 21, 0, n, n)
            P = pkg.orbit_projection(40, W, H)
            ref = orc.project(xyzw, rgba, P, W, H)
            for pack in (1, 0, 2, 1):
                p.set_option("pack", pack)
                if pack == 1:
                    p.upload_points(xyzw, rgba)       # (0 and 2 re-pack the resident cloud)
                want = {0: 0, 1: expect, 2: 1}[pack]
                assert p.get_option("packed") == want, (scene, pack)
                mb = p.get_option("packed_millibytes_per_point")
                assert (mb < 10500) if (want and pack == 1) else (mb <= 12200), (scene, pack, mb)
                assert (mb == 12000) == (want == 0)
                img, depth = p.project(P)
                assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]), (scene, pack)
                assert np.array_equal(img, ref["img"]), (scene, pack)
                back, cols = p.download_points()
                assert np.array_equal(back.view(np.uint32), xyzw.view(np.uint32)) and np.array_equal(cols, rgba)
        # every width in one cloud: constant axes, 1 .. 25 differing low bits, the ranges that fall back to 32 bits,
        # sign changes, NaN / inf / -0 (axis a of chunk c spans 2^((c + 7 a) % 33) bit patterns)
        rng = np.random.default_rng(8)
        m = 256 * 70 + 77
        xyz = np.empty((m, 3), np.float32)
        base = int(np.float32(2.0).view(np.uint32))
        for c in range(0, m, 256):
            k = min(256, m - c)
            for a in range(3):
                span = 1 << ((c // 256 + 7 * a) % 33)
                v = (base + rng.integers(0, span, size=k, dtype=np.uint64)) & 0xFFFFFFFF
                v[0], v[-1] = base & 0xFFFFFFFF, (base + span - 1) & 0xFFFFFFFF  # (the whole range is in use)
                xyz[c:c + k, a] = v.astype(np.uint32).view(np.float32)
        xyz[5, 0], xyz[6, 1], xyz[7, 2], xyz[300, 0] = np.nan, np.inf, -0.0, -1.5
        xyzw, rgba = cloud(xyz, rng.integers(0, 256, size=(m, 3), dtype=np.uint8))
        p.set_option("pack", 2)
        p.upload_points(xyzw, rgba)
        assert p.get_option("packed") == 1
        K = np.array([[100.0, 0, 320], [0, 100.0, 240], [0, 0, 1]])
        P = orc.compose_projection(K, np.eye(4))
        ref = orc.project(xyzw, rgba, P, W, H)
        img, depth = p.project(P)
        assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"])
        assert (ref["depth_bits"] != orc.EMPTY_DEPTH).sum() > 100
    finally:
        p.close()


def test_auto_reorder_option(pkg, orc):
    """Option "auto_reorder": 1 always sorts; 2 (the default) sorts a hash-ordered cloud but leaves a
    spatially ordered one and tiny clouds alone; 0 never sorts.  Frames are the same in every case."""
    n, W, H = 100_000, 320, 240
    xyzw, rgba = orc.generate("uniform_box", 6, 0, n, n)
    P = pkg.orbit_projection(64, W, H)
    ref = orc.project(xyzw, rgba, P, W, H)
    p = pkg.Projector(0)
    try:
        assert p.get_option("auto_reorder") == 2  # the default
        for policy, expect in ((1, True), (2, True), (0, False)):
            p.set_option("auto_reorder", policy)
            p.set_option("cull", 1)
            p.upload_points(xyzw, rgba)
            back, _ = p.download_points()
            assert bool(p.get_option("reordered")) == expect, policy
            assert np.array_equal(back, xyzw) == (not expect)  # the resident order is the Morton order when sorted
            p.set_resolution(W, H)
            img, depth = p.project(P)
            assert np.array_equal(img, ref["img"]) and np.array_equal(depth.view(np.uint32), ref["depth_bits"])
        p.set_option("auto_reorder", 2)
        # a Morton-ordered surface cloud and a tiny hash-ordered one stay as they are
        xs, cs = orc.generate("room_shell", 6, 0, n, n)
        p.upload_points(xs, cs)
        assert p.get_option("reordered") == 0 and p.get_option("order_ratio_ppm") > 0
        p.upload_points(xyzw[:5000], rgba[:5000])
        assert p.get_option("reordered") == 0
        # the reference loader's order: 0.25 m blocks, unordered inside (cloudreader.cpp:8-82), at a size where
        # that is visibly looser than an ideal order
        big, cb = orc.generate("room_shell", 7, 0, 4_000_000, 4_000_000)
        cell = np.floor(big[:, :3] / 0.25).astype(np.int64)
        key = (cell[:, 0] * 73856093) ^ (cell[:, 1] * 19349663) ^ (cell[:, 2] * 83492791)
        order = np.lexsort((np.random.default_rng(3).permutation(len(big)), key))
        p.upload_points(big[order], cb[order])
        ratio_blocks = p.get_option("order_ratio_ppm")
        p.upload_points(big, cb)
        assert p.get_option("order_ratio_ppm") < ratio_blocks
    finally:
        p.close()


def test_hot_tile_is_split_in_the_binned_form(pkg, orc, projector):
    """A run of frames whose points all fall into one tile: the binned form stays in use (no
    fallback to the atomic form), the tile is split over several workgroups, every frame stays
    exact, and so does the return to an ordinary view."""
    rng = np.random.default_rng(31)
    n = 400_000
    xyz = np.stack([rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n), rng.uniform(1.9, 2.1, n)], axis=1)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz.astype(np.float32), rgb)
    P_hot = kat_P(orc)                    # everything lands in a 10x10 pixel patch
    K = np.array([[4000.0, 0, 32], [0, 4000.0, 24], [0, 0, 1]])
    P_wide = orc.compose_projection(K, np.eye(4))  # zoomed in: spread over the whole 64x48 frame
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(64, 48)
    refs = {id(P): orc.project(xyzw, rgba, P, 64, 48) for P in (P_hot, P_wide)}
    for k in range(12):
        P = P_hot if k < 8 else P_wide
        filtered = bool(k & 1)
        img, depth = projector.project(P, filtered=filtered)
        st = projector.frame_stats()
        assert st["errors"] == 0
        if P is P_hot:
            assert st["entries"] == n and st["heaviest_tile"] > n // 4 and st["split_tiles"] >= 1 \
                and st["split_items"] >= 12, st
        ref = refs[id(P)]
        rd, ri = ref["depth_bits"], ref["img"]
        if filtered:
            rf = orc.filter(rd, ri)
            rd, ri = rf["depth"].view(np.uint32), rf["img"]
            assert np.array_equal(projector.download(pkg._lib.BUF_TENSOR).reshape(5, 48, 64), rf["tensor"]), k
        assert np.array_equal(depth.view(np.uint32), rd), k
        assert np.array_equal(img, ri), k


def test_async_host_outputs_match_the_synchronous_calls(pkg, orc, projector):
    """rtr_project_async / rtr_wait (rtr.h 4b): frames queued back to back into the two pinned output slots -- frame k's
    device-to-host copies run beside frame k + 1's kernels -- equal the frames of the synchronous calls and the
    oracle, filtered or not, also when a slot is reused without an explicit wait in between."""
    W, H, n = 640, 480, 300_000
    xyzw, rgba = orc.generate("room_shell", 91, 0, n, n)
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    poses = [pkg.orbit_projection(k, W, H) for k in (3, 140, 277, 410, 520, 731)]
    bufs = [projector.host_output_buffers(s) for s in range(2)]
    for filtered in (False, True):
        refs = []
        for P in poses:
            ref = orc.project(xyzw, rgba, P, W, H)
            rd, ri = ref["depth_bits"], ref["img"]
            if filtered:
                rf = orc.filter(rd, ri)
                rd, ri = rf["depth"].view(np.uint32), rf["img"]
            refs.append((rd, ri))
        # two frames in flight, then collect: slots alternate
        for k0 in range(0, len(poses), 2):
            projector.project_async(poses[k0], 0, filtered)
            projector.project_async(poses[k0 + 1], 1, filtered)
            for s in range(2):
                projector.wait_outputs(s)
                img, depth = bufs[s]
                assert np.array_equal(depth.view(np.uint32), refs[k0 + s][0]), (filtered, k0 + s)
                assert np.array_equal(img, refs[k0 + s][1]), (filtered, k0 + s)
        # a slot reused while its previous frame may still be on its way: the library orders the two itself
        for k in range(len(poses)):
            projector.project_async(poses[k], k & 1, filtered)
        projector.wait_outputs()
        for s, k in ((0, len(poses) - 2), (1, len(poses) - 1)):
            assert np.array_equal(bufs[s][1].view(np.uint32), refs[k][0]) and np.array_equal(bufs[s][0], refs[k][1])
        img_s, depth_s = projector.project(poses[-1], filtered=filtered)
        assert np.array_equal(depth_s.view(np.uint32), bufs[1][1].view(np.uint32)) and np.array_equal(img_s, bufs[1][0])
    with pytest.raises(pkg._lib.RtrError):
        projector.project_async(poses[0], 2)


def test_tile_store_error_reaches_the_caller(pkg, orc, projector):
    """A tile store that has to drop entries must never hand back a wrong frame as RTR_OK.  Option "debug_dyn_cap"
    shrinks the pool of dynamic stream extents so that a tile with more than 4096 entries overflows it (error
    code 2): the synchronous calls, rtr_synchronize and the frame statistics report it, the word is per frame
    (the next good frame reads 0 again and is exact)."""
    rng = np.random.default_rng(5)
    n = 60_000
    xyz = np.stack([rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n), rng.uniform(1.9, 2.1, n)], axis=1)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz.astype(np.float32), rgb)
    P = kat_P(orc)  # everything lands in a 10x10 pixel patch of one tile: ~n entries in one or two streams
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(64, 48)
    projector.set_option("split_threshold", 0)
    try:
        projector.set_option("debug_dyn_cap", 16)
        with pytest.raises(pkg._lib.RtrError) as e1:
            projector.project(P)
        assert e1.value.code == pkg._lib.RTR_ERR_INTERNAL and "pool" in str(e1.value)
        assert projector.frame_stats()["errors"] & 2
        projector.render(P)  # the asynchronous form: the error surfaces at the next synchronising call
        with pytest.raises(pkg._lib.RtrError) as e2:
            projector.synchronize()
        assert e2.value.code == pkg._lib.RTR_ERR_INTERNAL
        projector.synchronize()  # reported once
    finally:
        projector.set_option("debug_dyn_cap", -1)
        projector.set_option("split_threshold", 32768)
    img, depth = projector.project(P)
    st = projector.frame_stats()
    assert st["errors"] == 0 and st["entries"] == n and st["colour_chunks"] >= n // 256, st
    ref = orc.project(xyzw, rgba, P, 64, 48)
    assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"])


def test_ten_million_points_in_one_tile(pkg, orc, projector):
    """1e7 points inside one 32x32 tile (a distant overview): the tile kernel splits the tile over
    hundreds of workgroups; depth, image, accumulators and the filtered outputs against the oracle."""
    n, W, H = 10_000_000, 640, 480
    rng = np.random.default_rng(77)
    xyz = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), rng.uniform(39.0, 41.0, n)], axis=1)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz.astype(np.float32), rgb)
    K = np.array([[512.0, 0, 336.0], [0, 512.0, 240.0], [0, 0, 1.0]])  # ~8 px wide blob inside tile (10, 7)
    P = orc.compose_projection(K, np.eye(4))
    projector.set_option("keep_accum", 1)
    try:
        ref = _check_frame(pkg, orc, projector, xyzw, rgba, P, W, H)
    finally:
        projector.set_option("keep_accum", 0)
    st = projector.frame_stats()
    assert st["errors"] == 0 and st["entries"] == n and st["heaviest_tile"] == n and st["split_items"] >= 256, st
    assert int(ref["acc"][..., 3].sum()) > 0


def test_split_launch_is_skipped_until_a_frame_needs_it(pkg, orc, projector):
    """Whole frames launch the split kernel only while tiles above the split threshold have been seen (a mapped host
    word T1's epilogue writes; eight frames of grace after an upload).  After a run of ordinary frames the launch is off:
    the first frame with a hot tile is then processed without it -- the tile by its one workgroup, exact -- and reported,
    so the following frames get the split launch back; and off again after another run of ordinary frames."""
    rng = np.random.default_rng(32)
    n = 300_000
    xyz = np.stack([rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n), rng.uniform(1.9, 2.1, n)], axis=1)
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz.astype(np.float32), rgb)
    P_hot = kat_P(orc)                    # everything lands in a 10x10 pixel patch
    K = np.array([[4000.0, 0, 32], [0, 4000.0, 24], [0, 0, 1]])
    P_wide = orc.compose_projection(K, np.eye(4))  # zoomed in: spread over the whole 64x48 frame
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(64, 48)
    refs = {id(P): orc.project(xyzw, rgba, P, 64, 48) for P in (P_hot, P_wide)}
    seen = []
    for k, P in enumerate([P_wide] * 12 + [P_hot] * 3 + [P_wide] * 12 + [P_hot] * 2):
        filtered = bool(k & 1)
        img, depth = projector.project(P, filtered=filtered)
        st = projector.frame_stats()
        assert st["errors"] == 0, (k, st)
        seen.append((P is P_hot, st["split_tiles"], st["heaviest_tile"]))
        ref = refs[id(P)]
        rd, ri = ref["depth_bits"], ref["img"]
        if filtered:
            rf = orc.filter(rd, ri)
            rd, ri = rf["depth"].view(np.uint32), rf["img"]
            assert np.array_equal(projector.download(pkg._lib.BUF_TENSOR).reshape(5, 48, 64), rf["tensor"]), k
        assert np.array_equal(depth.view(np.uint32), rd), (k, seen)
        assert np.array_equal(img, ri), (k, seen)
    hot = [s for s in seen if s[0]]
    assert all(h[2] > 32768 for h in hot), seen
    # first hot frame of each run: no split launch behind it (no tile was split); the next ones: split again
    assert hot[0][1] == 0 and hot[1][1] >= 1 and hot[2][1] >= 1 and hot[3][1] == 0 and hot[4][1] >= 1, seen


def test_split_tile_with_a_single_slice(pkg, orc, projector):
    """Two hot tiles of very different weight: the slice size grows with the frame's split entries (at most 512 slice
    records), so the lighter tile is above the split threshold and still gets ONE slice -- its record goes through
    the split launch like any other (min into the depth buffer, sums into the accumulators, resolve)."""
    W, H = 640, 480
    rng = np.random.default_rng(78)
    n_a, n_b = 600_000, 700
    xa = np.stack([rng.uniform(-0.3, 0.3, n_a), rng.uniform(-0.3, 0.3, n_a), rng.uniform(39.0, 41.0, n_a)], axis=1)
    xb = np.stack([rng.uniform(9.7, 10.3, n_b), rng.uniform(-0.3, 0.3, n_b), rng.uniform(39.0, 41.0, n_b)], axis=1)
    xyz = np.concatenate([xa, xb]).astype(np.float32)
    rgb = rng.integers(0, 256, size=(len(xyz), 3), dtype=np.uint8)
    xyzw, rgba = cloud(xyz, rgb)
    K = np.array([[512.0, 0, 336.0], [0, 512.0, 240.0], [0, 0, 1.0]])  # blobs in tiles (10, 7) and (14, 7)
    P = orc.compose_projection(K, np.eye(4))
    projector.set_option("split_threshold", 512)
    projector.set_option("split_slice", 256)
    projector.set_option("keep_accum", 1)
    try:
        _check_frame(pkg, orc, projector, xyzw, rgba, P, W, H)
        st = projector.frame_stats()
        # slice = ceil(600700 / 512) = 1174 > 700: tile B is split (700 > 512) into one slice
        assert st["errors"] == 0 and st["split_tiles"] == 2 and st["slice"] >= 1174 and st["split_items"] == -(-n_a // st["slice"]) + 1, st
    finally:
        projector.set_option("keep_accum", 0)
        projector.set_option("split_threshold", 32768)
        projector.set_option("split_slice", 16384)


@pytest.mark.parametrize("tail_cus,split", [(0, 32768), (8, 32768), (0, 64)])
def test_overlap_option_keeps_every_frame(pkg, orc, projector, tail_cus, split):
    """Option "overlap": T1 of a whole-frame render runs on a second stream and fills the tile
    store the tail of the previous frame is not reading (optionally with disjoint CU masks).
    Frames queued back to back without synchronisation, mixed with phase calls and a new cloud,
    stay bit-identical to the oracle -- also when tiles are split (their pixels are then reset on
    the tail's stream, not by T1 beside the previous frame's tail: found by tools/fuzz_parity.py)."""
    W, H = 640, 480
    xyzw, rgba = orc.generate("room_shell", 77, 0, 300_000, 300_000)
    poses = [pkg.orbit_projection(k, W, H) for k in range(12)]
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    projector.set_option("tail_cus", tail_cus)
    projector.set_option("split_threshold", split)
    projector.set_option("split_slice", max(16, split // 2))
    try:
        projector.set_option("overlap", 1)
    except pkg.RtrError:
        if tail_cus:
            pytest.skip("CU-masked streams are not available on this device")
        raise
    try:
        def check(P, cloud_xyzw, cloud_rgba):
            ref = orc.project(cloud_xyzw, cloud_rgba, P, W, H)
            rf = orc.filter(ref["depth_bits"], ref["img"])
            assert np.array_equal(projector.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
            assert np.array_equal(projector.download(pkg._lib.BUF_MASK), rf["mask"])
            assert np.array_equal(projector.download(pkg._lib.BUF_MINMAX), rf["minmax"])

        for k in range(7):          # 7 frames in flight, no synchronisation in between
            projector.render(poses[k], True)
        check(poses[6], xyzw, rgba)
        projector.render(poses[7], True)
        projector.clear()           # phase calls use the active set on the tail stream
        projector.min_depth_pass(poses[8])
        projector.accumulate_pass(poses[8])
        projector.resolve()
        projector.filter()
        check(poses[8], xyzw, rgba)
        projector.render(poses[9], True)
        projector.render(poses[10], True)
        check(poses[10], xyzw, rgba)
        xyzw2, rgba2 = orc.generate("uniform_box", 78, 0, 200_000, 200_000)
        projector.render(poses[11], True)
        projector.upload_points(xyzw2, rgba2)   # must wait for the front stream too
        projector.render(poses[3], True)
        projector.render(poses[4], True)
        check(poses[4], xyzw2, rgba2)
        img, depth = projector.project(poses[5], filtered=True)
        ref = orc.project(xyzw2, rgba2, poses[5], W, H)
        rf = orc.filter(ref["depth_bits"], ref["img"])
        assert np.array_equal(img, rf["img"])
        assert np.array_equal(depth.view(np.uint32), rf["depth"].view(np.uint32))
    finally:
        projector.set_option("overlap", 0)
        projector.set_option("tail_cus", 0)
        projector.set_option("split_threshold", 32768)
        projector.set_option("split_slice", 16384)


def test_compute_full_handoff(pkg, orc, monkeypatch, tmp_path):
    """computeFull (project_cloud.cu:437-493): the model receives the resident fp16
    {1,5,H,W} tensor itself (zero copy, same stream) and its output goes through
    convertTo(CV_8UC3, 255.0).  The U-Net is out of scope (its weights are an LFS pointer), so the
    stand-in model returns the three colour planes: half(v / 255) * 255 rounds back to v, hence
    colour must equal the prefiltered image and the tensor seen by the model the oracle's."""
    torch = pytest.importorskip("torch")
    n, W, H = 50_000, 160, 128
    xyzw, rgba = orc.generate("room_shell", 5, 0, n, n)
    cal = pkg.benchmark_calibration(W, H)
    E = pkg.orbit_pose(12)
    ref = orc.project(xyzw, rgba, orc.compose_projection(cal.getIntrinsicsMatrix(), E), W, H)
    rf = orc.filter(ref["depth_bits"], ref["img"])
    seen = {}

    class Planes(torch.nn.Module):
        def forward(self, x):
            return x[:, 0:3]

    def model(x):
        seen["ptr"], seen["copy"] = x.data_ptr(), x.clone()
        return Planes()(x)

    pc = pkg.ProjectCloud(xyzw, rgba)
    color = np.empty((H, W, 3), np.uint8)
    depth = np.empty((H, W), np.float32)
    with pytest.raises(pkg.RtrError):
        pc.computeFull(cal, E, color, depth)          # no model given
    pc.set_model(model)
    assert pc.computeFull(cal, E, color, depth) == 1
    assert seen["ptr"] == pc.tensor_device_buffer().ptr   # the library's buffer, not a copy
    got = seen["copy"].cpu().numpy().view(np.uint16).reshape(5, H, W)
    assert np.array_equal(got, rf["tensor"])
    assert np.array_equal(color, rf["img"])
    assert np.array_equal(depth.view(np.uint32), rf["depth"].view(np.uint32))
    pc.projector.close()
    # the reference's way: a TorchScript file under $HOME/.render_cache (project_cloud.cu:225-246)
    monkeypatch.setenv("HOME", str(tmp_path))
    (tmp_path / ".render_cache").mkdir()
    torch.jit.script(Planes()).save(str(tmp_path / ".render_cache" / "model.pt"))
    with pytest.raises(FileNotFoundError):
        pkg.ProjectCloud(xyzw, rgba, "missing.pt")
    pc2 = pkg.ProjectCloud(xyzw, rgba, "model.pt")
    color2 = np.empty((H, W, 3), np.uint8)
    assert pc2.computeFull(cal, E, color2, None) == 1
    assert np.array_equal(color2, rf["img"])
    pc2.projector.close()


@pytest.mark.parametrize("levels", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("window,strength,thr", [(0.02, 1.025, 0.03), (0.0, 1.0, 0.0), (0.15, 1.2, 0.2)])
def test_parameter_sweep(pkg, orc, projector, levels, window, strength, thr):
    """rtr_set_params: every pyramid depth (levels 4 takes the fused one-launch prefilter, the others
    the level-by-level kernels), window / strength / threshold away from the reference's constants
    (render.cu:106, project_cloud.cu:23-25) -- whole-frame call and phase calls against the oracle
    run with the same parameters."""
    W, H = 640, 480          # 480 >> 6 = 7: rows 448.. lie outside the pyramid for levels 5 and 6
    xyzw, rgba = orc.generate("room_shell", 21, 0, 600_000, 600_000)
    P = pkg.orbit_projection(17, W, H)
    prm = orc.default_params()
    prm.depth_window, prm.filter_strength, prm.gradient_threshold, prm.levels = window, strength, thr, levels
    ref = orc.project(xyzw, rgba, P, W, H, params=prm)
    rf = orc.filter(ref["depth_bits"], ref["img"], params=prm)
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    projector.set_params(depth_window=window, filter_strength=strength, gradient_threshold=thr, levels=levels)
    try:
        for phases in (False, True):
            if phases:
                projector.clear()
                projector.min_depth_pass(P)
                projector.accumulate_pass(P)
                projector.resolve()
                assert np.array_equal(projector.download(pkg._lib.BUF_ACCUM), ref["acc"])
                projector.filter()
            else:
                projector.render(P, True)
            assert np.array_equal(projector.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"]), phases
            assert np.array_equal(projector.download(pkg._lib.BUF_MASK), rf["mask"]), phases
            assert np.array_equal(projector.download(pkg._lib.BUF_DEPTH), rf["depth"].view(np.uint32)), phases
            assert np.array_equal(projector.download(pkg._lib.BUF_IMAGE), rf["img"]), phases
            assert np.array_equal(projector.download(pkg._lib.BUF_MINMAX), rf["minmax"]), phases
    finally:
        projector.set_params(depth_window=0.02, filter_strength=1.025, gradient_threshold=0.03, levels=4)


def test_lean_frames_match_the_epilogue_form(pkg, orc, projector):
    """Option "lean" (default): whole frames end the point kernel without its epilogue -- the tile workgroups read and
    reset the stream counters themselves, an extra workgroup keeps the books.  Same frames, same statistics as with the
    epilogue (lean = 0), across switches between the two forms, the phase calls in between, an empty cloud, a
    resolution with 64-wide tiles, and the colour-chunk / entry counts bench.py prices the kernel by."""
    n, W, H = 400_000, 640, 480
    xyzw, rgba = orc.generate("room_shell", 0xC0FFEE07, 0, n, n)
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    poses = [pkg.orbit_projection(k, W, H) for k in (0, 140, 275, 410, 555, 690, 835, 970)]
    stats = {}
    # (lean_identity / lean_early: how a lean frame's tile workgroup finds its tile and when it requests its entries)
    forms = ((1, 1, -1), (0, 1, -1), (1, 1, -1), (1, 0, 0), (1, 1, 1), (1, 0, 1), (1, 1, 0))
    for lean, ident, early in forms:
        projector.set_option("lean", lean)
        projector.set_option("lean_identity", ident)
        projector.set_option("lean_early", early)
        for k, P in enumerate(poses):
            filtered = bool(k & 1)
            img, depth = projector.project(P, filtered=filtered)
            st = projector.frame_stats()
            assert st["errors"] == 0 and st["items"] == (W // 32) * (H // 32) and st["split_items"] == 0
            stats.setdefault(k, []).append((st["entries"], st["heaviest_tile"], st["colour_chunks"]))
            ref = orc.project(xyzw, rgba, P, W, H)
            rd, ri = ref["depth_bits"], ref["img"]
            if filtered:
                rf = orc.filter(rd, ri)
                rd, ri = rf["depth"].view(np.uint32), rf["img"]
            assert np.array_equal(depth.view(np.uint32), rd) and np.array_equal(img, ri), (lean, ident, early, k)
            if k == 3:  # the phase calls between two lean frames (they bin for themselves)
                projector.clear(); projector.min_depth_pass(P); projector.accumulate_pass(P); projector.resolve()
                assert np.array_equal(projector.download(pkg._lib.BUF_DEPTH), ref["depth_bits"])
                assert np.array_equal(projector.download(pkg._lib.BUF_IMAGE), ref["img"])
    for k, rows in stats.items():
        assert all(row == rows[0] for row in rows) and len(rows) == len(forms) and rows[0][0] > 0, (k, rows)
    # frames WITHOUT a statistics call in between (the next lean frame folds the previous one's), then the last one's
    projector.set_option("lean", 1)
    projector.set_option("lean_early", -1)
    for P in poses:
        projector.render(P, True)
    st = projector.frame_stats()
    assert (st["entries"], st["heaviest_tile"], st["colour_chunks"]) == stats[len(poses) - 1][0]
    # 64-wide tiles (four streams per tile) and an empty cloud
    projector.set_resolution(3840, 2160)
    P4 = pkg.orbit_projection(3, 3840, 2160)
    ref = orc.project(xyzw, rgba, P4, 3840, 2160)
    for early in (-1, 1, 0):  # (8160 tiles: more than are resident at once, so the launch order stays in use)
        projector.set_option("lean_early", early)
        for _ in range(2):
            img, depth = projector.project(P4)
            assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"]), early
    projector.set_option("lean_early", -1)
    projector.upload_points(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.uint8))
    img, depth = projector.project(P4)
    assert (depth.view(np.uint32) == 0x7F7FFFFF).all() and not img.any()


def test_packed_only_residency_and_the_adaptive_extent_pool(pkg, orc, projector):
    """Footprint (option keep_soa = 0, pool_worst_case = 0: the defaults).  A packed cloud keeps no fp32 SoA arrays: the
    calls that read fp32 coordinates decode them again, bit for bit (rtr_download_points, mode 0, the sort, pack = 0).
    The extent pool is sized by the frames seen; a frame that overflows it (every point in ONE tile stream right after
    upload) is rendered again by the synchronising call -- the caller sees the right frame and RTR_OK."""
    n, W, H = 2_000_000, 640, 480
    xyzw, rgba = orc.generate("room_shell", 0xC0FFEE09, 0, n, n)
    projector.upload_points(xyzw, rgba)
    projector.set_resolution(W, H)
    assert projector.get_option("packed") == 1 and projector.get_option("keep_soa") == 0
    small = projector.get_option("resident_millibytes_per_point")
    P = pkg.orbit_projection(7, W, H)
    ref = orc.project(xyzw, rgba, P, W, H)
    img, depth = projector.project(P)
    assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"])
    back, cols = projector.download_points()           # decoded from the packed form
    assert np.array_equal(back.view(np.uint32), xyzw.view(np.uint32)) and np.array_equal(cols, rgba)
    projector.set_option("mode", 0)                     # the atomic form reads fp32 coordinates
    img, depth = projector.project(P)
    assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"])
    projector.set_option("mode", 1)
    projector.set_option("pack", 0)                     # fp32 becomes the resident form ...
    img, depth = projector.project(P)
    assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"])
    projector.set_option("pack", 1)                     # ... and the packed one again
    assert projector.get_option("packed") == 1
    packed_only = projector.get_option("resident_millibytes_per_point")  # (the extent pool has grown with the frames seen)
    assert small > 0 and packed_only > 0
    projector.set_option("keep_soa", 1)
    assert projector.get_option("resident_millibytes_per_point") >= packed_only + 11_900
    projector.set_option("keep_soa", 0)
    assert projector.get_option("resident_millibytes_per_point") == packed_only
    projector.reorder_points()
    img, depth = projector.project(P, filtered=True)
    rf = orc.filter(ref["depth_bits"], ref["img"])
    assert np.array_equal(depth.view(np.uint32), rf["depth"].view(np.uint32)) and np.array_equal(img, rf["img"])
    # the adaptive pool: n / 2 entries for a fresh cloud; the whole cloud inside one 32x16 tile needs ~2 n
    K = np.array([[1.0, 0, 8.0], [0, 1.0, 8.0], [0, 0, 1]])   # focal length 1 px: everything within a pixel of (8, 8)
    E = np.eye(4)
    E[2, 3] = 20.0
    P_one = orc.compose_projection(K, E)
    ref1 = orc.project(xyzw, rgba, P_one, W, H)
    assert orc.envelope_points(xyzw, P_one, W, H, 0, 0)["accepted"] == n and (ref1["depth_bits"] != 0x7F7FFFFF).sum() <= 9
    img, depth = projector.project(P_one)               # overflows, is repeated inside the call
    assert np.array_equal(depth.view(np.uint32), ref1["depth_bits"]) and np.array_equal(img, ref1["img"])
    assert projector.frame_stats()["errors"] == 0
    projector.upload_points(xyzw, rgba)                 # a new cloud: adaptive again; this time through render + synchronize
    projector.render(P_one, False)
    projector.synchronize()
    assert np.array_equal(projector.download(pkg._lib.BUF_DEPTH), ref1["depth_bits"])
    assert np.array_equal(projector.download(pkg._lib.BUF_IMAGE), ref1["img"])
