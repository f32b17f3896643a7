"""CPU: the data formats either side of the projector."""
import numpy as np
import pytest


def test_calibration_custom_format_roundtrip(pkg, tmp_path):
    F = pkg.formats
    p = tmp_path / "calib.txt"
    # README.md:95-102 (commas allowed in the distortion line: CameraCalibration.cpp:180)
    p.write_text("1920 1080\n1536.5 0 960.25\n0 1535.75 540.5\n0 0 1\n0.1, -0.2, 0.001, 0.002, 0.05\n0\n")
    cal = F.load_calibration(p)
    assert (cal.getWidth(), cal.getHeight()) == (1920, 1080)
    K = cal.getIntrinsicsMatrix()
    assert (K[0, 0], K[0, 2], K[1, 1], K[1, 2], K[2, 2]) == (1536.5, 960.25, 1535.75, 540.5, 1.0)
    assert cal.m_dists == [0.1, -0.2, 0.001, 0.002, 0.05] and cal.m_fishEye is False
    q = tmp_path / "calib2.txt"
    F.write_calibration_txt(q, cal)
    assert np.array_equal(F.load_calibration(q).getIntrinsicsMatrix(), K)
    bad = tmp_path / "bad.txt"
    bad.write_text("640 480\n500 0 320\n0 500 240\n0 0 1\n0.1 0.2 0.3 0.4\n0\n")  # pinhole needs 5
    with pytest.raises(ValueError):
        F.load_calibration(bad)


def test_calibration_cameras_txt(pkg, tmp_path):
    F = pkg.formats
    p = tmp_path / "cameras.txt"
    p.write_text("# Camera list\n1 OPENCV 3840 2160 3072.123456789 3071.9 1920.5 1080.25 0 0 0 0 0\n")
    cal = F.load_calibration(p)
    K = cal.getIntrinsicsMatrix()
    # the reference reads these as float (CameraCalibration.cpp:123-137)
    assert K[0, 0] == float(np.float32(3072.123456789)) and K[1, 2] == 1080.25
    assert (cal.getWidth(), cal.getHeight()) == (3840, 2160)
    p.write_text("1 PINHOLE 640 480 500 500 320 240\n")
    with pytest.raises(ValueError):
        F.load_calibration(p)
    F.write_cameras_txt(p, pkg.benchmark_calibration(3840, 2160))
    assert np.array_equal(F.load_calibration(p).getIntrinsicsMatrix(),
                          pkg.benchmark_calibration(3840, 2160).getIntrinsicsMatrix())


def test_trajectory_formats_roundtrip(pkg, tmp_path):
    F = pkg.formats
    poses = [pkg.orbit_pose(k) for k in range(0, 1000, 37)]
    F.write_images_txt(tmp_path / "images.txt", poses)
    back = F.read_trajectory_colmap(tmp_path / "images.txt")
    assert len(back) == len(poses) and back[0][1] == "frame_1.png"
    for (E, _), ref in zip(back, poses):
        assert np.allclose(E, ref, atol=1e-12)
    F.write_trajectory_tum(tmp_path / "traj.txt", poses)
    back = F.read_trajectory_tum(tmp_path / "traj.txt")
    for E, ref in zip(back, poses):
        assert np.allclose(E, ref, atol=1e-12)
    # quaternion convention of cv::Quatd(w,x,y,z): 90 deg about z maps x -> y
    R = F.quat_to_rot(np.sqrt(0.5), 0, 0, np.sqrt(0.5))
    assert np.allclose(R @ [1, 0, 0], [0, 1, 0])
    # an un-normalised quaternion is normalised first (main.cpp:38)
    assert np.allclose(F.quat_to_rot(2, 0, 0, 0), np.eye(3))


def test_ply_roundtrip_and_bgr_order(pkg, orc, tmp_path):
    F = pkg.formats
    xyzw, rgba = orc.generate("room_shell", 0xC0FFEE01, 0, 100_000, 100_000)  # config C1: 100k-point .ply
    F.write_ply(tmp_path / "c1.ply", xyzw[:, :3], rgba[:, :3])
    xyz, bgr = F.read_ply(tmp_path / "c1.ply")
    assert np.array_equal(xyz.view(np.uint32), xyzw[:, :3].view(np.uint32).reshape(-1, 3))
    assert np.array_equal(bgr, rgba[:, 2::-1])  # cloudreader.cpp:168: colours are stored B,G,R
    (tmp_path / "a.ply").write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\n"
                                    "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                                    "end_header\n1 2 3 10 20 30\n-1 0.5 4 1 2 3\n")
    xyz, bgr = F.read_ply(tmp_path / "a.ply")
    assert xyz.tolist() == [[1, 2, 3], [-1, 0.5, 4]] and bgr.tolist() == [[30, 20, 10], [3, 2, 1]]


def test_grid_and_pcd_oct(pkg, orc, tmp_path):
    F = pkg.formats
    xyzw, rgba = orc.generate("uniform_box", 5, 0, 50_000, 50_000)
    g = F.compute_grid(xyzw[:, :3], rgba[:, :3])
    # box x,z in [-4,4], y in [-1.5,1.5] -> bb rounded to [-4,4] x [-2,2] x [-4,4], 0.25 m cells
    assert g.num_blocks == (32, 16, 32) and g.num_points() == 50_000
    for b in (0, len(g) // 2, len(g) - 1):
        pts = g.xyz[g.offsets[b]:g.offsets[b + 1]]
        assert (pts >= g.bb_min[b] - 1e-6).all() and (pts <= g.bb_max[b] + 1e-6).all()
        assert np.allclose(g.bb_max[b] - g.bb_min[b], 0.25)
    # flattening keeps every point exactly once (Octreegrid.h:162-180)
    a = np.sort(g.vertex_positions()[:, :3].view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel(), order=("x", "y", "z"))
    b = np.sort(np.ascontiguousarray(xyzw[:, :3]).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel(), order=("x", "y", "z"))
    assert np.array_equal(a, b)
    assert (g.vertex_positions()[:, 3] == 1).all() and (g.vertex_colors()[:, 3] == 255).all()
    F.write_pcd_oct(tmp_path / "pcd.oct", g)
    h = F.read_pcd_oct(tmp_path / "pcd.oct")
    assert h.num_blocks == g.num_blocks and np.array_equal(h.keys, g.keys)
    assert np.array_equal(h.xyz, g.xyz) and np.array_equal(h.colors, g.colors)
    assert np.array_equal(h.bb_min, g.bb_min) and np.array_equal(h.bb_max, g.bb_max)
    # header layout of Octreegrid.h:62-66: numBlocks_x, _y, _z, numBlocks as int32
    assert np.fromfile(tmp_path / "pcd.oct", np.int32, 4).tolist() == [32, 16, 32, len(g)]


def test_grid_keys_out_of_range_cells_like_the_reference(pkg, capsys):
    """cloudreader.cpp:50-58: a point exactly on the rounded-out maximum lands in cell index numBlocks
    (one past the grid); the reference prints "out of bounds" and keys it as computed -- no clipping."""
    F = pkg.formats
    xyz = np.array([[0.1, 0.1, 0.1], [1.0, 0.5, 0.5], [0.6, 0.6, 0.6]], np.float32)   # x = 1.0 = ceil(bbMax.x)
    g = F.compute_grid(xyz, np.zeros((3, 3), np.uint8))
    assert g.num_blocks == (4, 4, 4)
    assert "out of bounds: 4, 2, 2" in capsys.readouterr().err
    # encodeKey(4, 2, 2) = 4 + 2 * 4 + 2 * 16 = 44, which decodes (Octreegrid.h:116-121) to cell (0, 3, 2)
    assert 44 in g.keys.tolist()
    b = g.keys.tolist().index(44)
    assert np.array_equal(g.xyz[g.offsets[b]:g.offsets[b + 1]], xyz[1:2])
    assert np.allclose(g.bb_min[b], [0.0, 0.75, 0.5]) and np.allclose(g.bb_max[b], [0.25, 1.0, 0.75])
    assert g.num_points() == 3  # nothing dropped, nothing moved


def test_c1_plumbing_ply_to_frame_on_cpu(pkg, orc, tmp_path):
    """BASELINE config C1: 100k-point synthetic .ply -> 640x480 via the naive host-loop CPU
    projector: file -> grid -> flattened arrays -> oracle frame == frame of the original cloud."""
    F = pkg.formats
    xyzw, rgba = orc.generate("room_shell", 0xC0FFEE01, 0, 100_000, 100_000)
    F.write_ply(tmp_path / "c1.ply", xyzw[:, :3], rgba[:, :3])
    g = F.compute_grid(*F.read_ply(tmp_path / "c1.ply"))
    P = pkg.orbit_projection(0, 640, 480)
    a = orc.project(g.vertex_positions(), g.vertex_colors(), P, 640, 480)
    b = orc.project(xyzw, np.ascontiguousarray(rgba[:, [2, 1, 0, 3]]), P, 640, 480)
    assert np.array_equal(a["depth_bits"], b["depth_bits"]) and np.array_equal(a["img"], b["img"])
