"""-m gpu: BASELINE config C3 at FULL size (1e8 points -> 1920x1080 + prefilter).

Direct check: the cloud synthesised on the GPU is downloaded and projected by the multi-thread
oracle on the host (about a second on the box's cores) and compared bit for bit.  On top, the
size-independent properties the domain offers: mode 0 == mode 1, chunk culling / Morton reorder
keep the frame, and a 2-way point split merged by MIN / SUM equals the unsplit frame."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, W, H = 100_000_000, 1920, 1080


def _threads():
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 8


@pytest.mark.parametrize("scene", ["room_shell", "uniform_box"])
def test_c3_full_size_against_oracle(pkg, orc, scene):
    import torch
    p = pkg.Projector(0)
    try:
        p.set_option("auto_reorder", 0)  # the generator's own order (checked slice by slice below)
        p.generate_synthetic(scene, 0xC0FFEE03, 0, N, N)
        p.set_resolution(W, H)
        P = pkg.orbit_projection(17, W, H)
        img, depth = p.project(P)
        xyzw, rgba = p.download_points()
        # the generator itself at full size: spot-check slices against the oracle's
        for first in (0, 54_321_000, N - 4096):
            rx, rc = orc.generate(scene, 0xC0FFEE03, first, 4096, N)
            assert np.array_equal(xyzw[first:first + 4096].view(np.uint32), rx.view(np.uint32))
            assert np.array_equal(rgba[first:first + 4096], rc)
        ref = orc.MTProjector(W, H, _threads()).project(xyzw, rgba, P)
        assert np.array_equal(depth.view(np.uint32), ref["depth_bits"])
        assert np.array_equal(img, ref["img"])
        rf = orc.filter(ref["depth_bits"], ref["img"])
        img_f, depth_f = p.project(P, filtered=True)
        assert np.array_equal(depth_f.view(np.uint32), rf["depth"].view(np.uint32))
        assert np.array_equal(img_f, rf["img"])
        assert np.array_equal(p.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
        del xyzw, rgba
        # properties: the other kernel mode, culling and reordering leave the frame unchanged
        p.set_option("mode", 0)
        img0, depth0 = p.project(P)
        assert np.array_equal(img0, img) and np.array_equal(depth0.view(np.uint32), depth.view(np.uint32))
        p.set_option("mode", 1)
        p.set_option("cull", 1)
        img1, depth1 = p.project(P)
        assert np.array_equal(img1, img) and np.array_equal(depth1.view(np.uint32), depth.view(np.uint32))
        if scene == "uniform_box":
            p.reorder_points()
            img2, depth2 = p.project(P)
            assert np.array_equal(img2, img) and np.array_equal(depth2.view(np.uint32), depth.view(np.uint32))
        p.set_option("cull", 0)
        # 2-way shard + MIN / SUM merge on the device views == the unsplit frame
        p.set_option("keep_accum", 1)
        p.render(P)
        acc_full = torch.as_tensor(p.device_buffer(pkg._lib.BUF_ACCUM, "<i4"), device="cuda").clone()
        dep_full = torch.as_tensor(p.device_buffer(pkg._lib.BUF_DEPTH, "<i4"), device="cuda").clone()
        p.synchronize()
        torch.cuda.synchronize()
        halves = []
        for r in range(2):
            lo, hi = pkg.shard_range(N, r, 2)
            p.generate_synthetic(scene, 0xC0FFEE03, lo, hi - lo, N)
            p.clear()
            p.min_depth_pass(P)
            p.synchronize()
            halves.append(torch.as_tensor(p.device_buffer(pkg._lib.BUF_DEPTH, "<i4"), device="cuda").clone())
        dmin = torch.minimum(halves[0], halves[1])
        assert torch.equal(dmin, dep_full)
        acc = torch.zeros_like(acc_full)
        for r in range(2):
            lo, hi = pkg.shard_range(N, r, 2)
            p.generate_synthetic(scene, 0xC0FFEE03, lo, hi - lo, N)
            p.clear()
            p.min_depth_pass(P)
            dview = torch.as_tensor(p.device_buffer(pkg._lib.BUF_DEPTH, "<i4"), device="cuda")
            p.synchronize()
            dview.copy_(dmin)
            torch.cuda.synchronize()
            p.accumulate_pass(P)
            p.synchronize()
            acc += torch.as_tensor(p.device_buffer(pkg._lib.BUF_ACCUM, "<i4"), device="cuda")
        assert torch.equal(acc, acc_full)
        # the default upload policy at full size: the hash-ordered box is sorted, the room is left alone
        p.set_option("auto_reorder", 2)
        p.generate_synthetic(scene, 0xC0FFEE03, 0, N, N)
        assert bool(p.get_option("reordered")) == (scene == "uniform_box")
        img3, depth3 = p.project(P)
        assert np.array_equal(img3, img) and np.array_equal(depth3.view(np.uint32), depth.view(np.uint32))
    finally:
        p.close()


def test_chunked_upload_roundtrip(pkg, orc):
    """rtr_upload_points stages through 16 M-point chunks: 40 M points (3 chunks) uploaded from
    host AoS arrays must give the same cloud and the same frame as the on-device generator."""
    n = 40_000_000
    a, b = pkg.Projector(0), pkg.Projector(0)
    try:
        a.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
        xyzw, rgba = a.download_points()
        b.upload_points(xyzw, rgba)
        x2, c2 = b.download_points(n - 1000, 1000)
        assert np.array_equal(x2.view(np.uint32), xyzw[-1000:].view(np.uint32)) and np.array_equal(c2, rgba[-1000:])
        P = pkg.orbit_projection(321, W, H)
        for p in (a, b):
            p.set_resolution(W, H)
        ia, da = a.project(P, filtered=True)
        ib, db = b.project(P, filtered=True)
        assert np.array_equal(ia, ib) and np.array_equal(da.view(np.uint32), db.view(np.uint32))
        # tight strides (12 B xyz, 3 B rgb) through the same path
        b.upload_points(np.ascontiguousarray(xyzw[:, :3]), np.ascontiguousarray(rgba[:, :3]))
        ic, dc = b.project(P, filtered=True)
        assert np.array_equal(ia, ic) and np.array_equal(da.view(np.uint32), dc.view(np.uint32))
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("scene", ["room_shell", "uniform_box"])
def test_trajectory_sweep_against_oracle(pkg, orc, scene):
    """Every 37th pose of the 1000-pose benchmark orbit at 1e7 points, with and without chunk
    culling: tile loads, heavy-tile order and the culling boxes change with the view."""
    n = 10_000_000
    p = pkg.Projector(0)
    try:
        p.generate_synthetic(scene, 0xC0FFEE02, 0, n, n)
        xyzw, rgba = p.download_points()
        p.set_resolution(W, H)
        mt = orc.MTProjector(W, H, _threads())
        for i, k in enumerate(range(0, 1000, 37)):
            P = pkg.orbit_projection(k, W, H)
            ref = mt.project(xyzw, rgba, P)
            p.set_option("cull", i & 1)
            img, depth = p.project(P)
            assert np.array_equal(depth.view(np.uint32), ref["depth_bits"]), (scene, k)
            assert np.array_equal(img, ref["img"]), (scene, k)
    finally:
        p.close()


def test_c5_single_gpu_share_4k_and_colmap_replay(pkg, orc, tmp_path):
    """One GPU's share of BASELINE config C5: points [0, 1.25e8) of the 1e9-point room_shell cloud ->
    3840x2160 + prefilter.  (a) one pose against the multi-thread oracle on the host: depth, image, mask
    and fp16 tensor bit for bit; (b) 100 poses of the 1000-pose orbit replayed from COLMAP
    cameras.txt / images.txt through the drop-in class (computeFilteredRGBD, host outputs), every 10th
    frame checked against the oracle."""
    F = pkg.formats
    n, total, W4, H4 = 125_000_000, 1_000_000_000, 3840, 2160
    cal = pkg.benchmark_calibration(W4, H4)
    F.write_cameras_txt(tmp_path / "cameras.txt", cal)
    # (poses 200..299 of the orbit look at the x = -4 wall, which this share of the cloud holds)
    F.write_images_txt(tmp_path / "images.txt", [pkg.orbit_pose(k) for k in range(200, 300)])
    cal_back = F.load_calibration(tmp_path / "cameras.txt")
    poses = [E for E, _ in F.read_trajectory_colmap(tmp_path / "images.txt")]
    assert (cal_back.getWidth(), cal_back.getHeight()) == (W4, H4) and len(poses) == 100
    pc = pkg.ProjectCloud(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.uint8))
    p = pc.projector
    try:
        p.generate_synthetic("room_shell", 0xC0FFEE05, 0, n, total)
        xyzw, rgba = p.download_points()
        mt = orc.MTProjector(W4, H4, _threads())
        rgb = np.empty((H4, W4, 3), np.uint8)
        depth = np.empty((H4, W4), np.float32)
        checked = 0
        for k, E in enumerate(poses):
            assert pc.computeFilteredRGBD(cal_back, E, rgb, depth) == 1
            if k % 10:
                continue
            P = orc.compose_projection(cal_back.getIntrinsicsMatrix(), E)
            ref = mt.project(xyzw, rgba, P)
            rf = orc.filter(ref["depth_bits"], ref["img"])
            assert np.array_equal(depth.view(np.uint32), rf["depth"].view(np.uint32)), k
            assert np.array_equal(rgb, rf["img"]), k
            if k == 0:  # (a): everything the consumer sees, plus the unfiltered frame
                assert np.array_equal(p.download(pkg._lib.BUF_MASK), rf["mask"])
                assert np.array_equal(p.download(pkg._lib.BUF_TENSOR).reshape(5, H4, W4), rf["tensor"])
                assert np.array_equal(p.download(pkg._lib.BUF_MINMAX), rf["minmax"])
                img0, depth0 = p.project(P)
                assert np.array_equal(depth0.view(np.uint32), ref["depth_bits"]) and np.array_equal(img0, ref["img"])
                st = p.frame_stats()
                assert st["errors"] == 0 and st["entries"] > 1_000_000
            checked += 1
        assert checked == 10
    finally:
        p.close()
