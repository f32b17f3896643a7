"""The C++ host side above the C ABI: include/rtr_project_cloud.hpp (the reference's
ProjectCloud surface, project_cloud.h:11-19) compiled with plain g++ against librtr_hip.so.
CPU: it compiles and links.  GPU: it runs and matches the oracle bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _build(tmp_path, pkg):
    exe = str(tmp_path / "facade_check")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_check.cpp"), "-o", exe, pkg.LIB_PATH,
                           "-Wl,-rpath," + libdir])
    return exe


def test_facade_compiles_and_links(tmp_path, pkg):
    assert os.path.exists(_build(tmp_path, pkg))


def _build_compute_full(tmp_path, pkg):
    """RTR_WITH_TORCH: the facade's computeFull against the libtorch that ships with the torch wheel."""
    torch = pytest.importorskip("torch")
    ti = os.path.dirname(torch.__file__)
    if not os.path.exists(os.path.join(ti, "include", "torch", "script.h")):
        pytest.skip("libtorch headers are not in this image")
    exe = str(tmp_path / "compute_full_check")
    libdir = os.path.dirname(pkg.LIB_PATH)
    tl = os.path.join(ti, "lib")
    hip = [lib for lib in ("torch_hip", "c10_hip") if os.path.exists(os.path.join(tl, "lib%s.so" % lib))]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ti, "include"),
                           "-I" + os.path.join(ti, "include", "torch", "csrc", "api", "include"),
                           os.path.join(ROOT, "tests", "cpp", "compute_full_check.cpp"), "-o", exe, pkg.LIB_PATH,
                           "-L" + tl, "-ltorch", "-ltorch_cpu", "-lc10", "-Wl,--no-as-needed"] + ["-l" + h for h in hip] +
                          ["-Wl,--as-needed", "-Wl,-rpath," + tl, "-Wl,-rpath," + libdir])
    return exe


def test_compute_full_facade_compiles_and_links(tmp_path, pkg):
    assert os.path.exists(_build_compute_full(tmp_path, pkg))


@pytest.mark.gpu
def test_cpp_compute_full_matches_oracle(tmp_path, pkg, orc):
    """C++ computeFull (project_cloud.h:17-18, project_cloud.cu:437-493) on ROCm libtorch with a TorchScript
    stand-in for the U-Net (its weights are an LFS pointer) that returns the three colour planes:
    half(v / 255) * 255 rounds back to v, so colour must equal the prefiltered image, depth the prefiltered
    depth, and the resident tensor the oracle's -- the same expectations as the Python path
    (test_compute_full_handoff)."""
    import torch
    exe = _build_compute_full(tmp_path, pkg)
    n, W, H = 50_000, 160, 128
    xyzw, rgba = orc.generate("room_shell", 5, 0, n, n)
    cal, E = pkg.benchmark_calibration(W, H), pkg.orbit_pose(12)
    with open(tmp_path / "cloud.bin", "wb") as f:
        f.write(np.uint64(n).tobytes())
        f.write(np.ascontiguousarray(xyzw[:, :3]).tobytes())
        f.write(np.ascontiguousarray(rgba[:, :3]).tobytes())
    with open(tmp_path / "cam.bin", "wb") as f:
        f.write(np.ascontiguousarray(cal.getIntrinsicsMatrix(), np.float64).tobytes())
        f.write(np.ascontiguousarray(E, np.float64).tobytes())

    class Planes(torch.nn.Module):
        def forward(self, x):
            return x[:, 0:3]

    (tmp_path / ".render_cache").mkdir()
    torch.jit.script(Planes()).save(str(tmp_path / ".render_cache" / "model.pt"))
    out = str(tmp_path / "out")
    env = dict(os.environ, HOME=str(tmp_path))
    res = subprocess.run([exe, str(tmp_path / "cloud.bin"), str(W), str(H), str(tmp_path / "cam.bin"), "model.pt", out],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    ref = orc.project(xyzw, rgba, orc.compose_projection(cal.getIntrinsicsMatrix(), E), W, H)
    rf = orc.filter(ref["depth_bits"], ref["img"])
    rd = lambda ext, dt: np.fromfile(out + ext, dtype=dt)  # noqa: E731
    assert np.array_equal(rd(".rgb", np.uint8), rf["img"].reshape(-1))
    assert np.array_equal(rd(".depth", np.uint32), rf["depth"].view(np.uint32).reshape(-1))
    assert np.array_equal(rd(".tensor", np.uint16), rf["tensor"].reshape(-1))


@pytest.mark.gpu
def test_facade_matches_oracle(tmp_path, pkg, orc):
    exe = _build(tmp_path, pkg)
    n, W, H = 30_000, 320, 240
    xyzw, rgba = orc.generate("room_shell", 5, 0, n, n)
    cal, E = pkg.benchmark_calibration(W, H), pkg.orbit_pose(222)
    with open(tmp_path / "cloud.bin", "wb") as f:
        f.write(np.uint64(n).tobytes())
        f.write(np.ascontiguousarray(xyzw[:, :3]).tobytes())
        f.write(np.ascontiguousarray(rgba[:, :3]).tobytes())
    with open(tmp_path / "cam.bin", "wb") as f:
        f.write(np.ascontiguousarray(cal.getIntrinsicsMatrix(), np.float64).tobytes())
        f.write(np.ascontiguousarray(E, np.float64).tobytes())
    out = str(tmp_path / "out")
    subprocess.check_call([exe, str(tmp_path / "cloud.bin"), str(W), str(H), str(tmp_path / "cam.bin"), out])
    P = orc.compose_projection(cal.getIntrinsicsMatrix(), E)
    ref = orc.project(xyzw, rgba, P, W, H)
    rf = orc.filter(ref["depth_bits"], ref["img"])
    rd = lambda ext, dt: np.fromfile(out + ext, dtype=dt)  # noqa: E731
    assert np.array_equal(rd(".rgb", np.uint8), ref["img"].reshape(-1))
    assert np.array_equal(rd(".depth", np.uint32), ref["depth_bits"].reshape(-1))
    assert np.array_equal(rd(".frgb", np.uint8), rf["img"].reshape(-1))
    assert np.array_equal(rd(".fdepth", np.uint32), rf["depth"].view(np.uint32).reshape(-1))
    assert np.array_equal(rd(".tensor", np.uint16), rf["tensor"].reshape(-1))


def test_cpp_example_compiles(tmp_path, pkg):
    exe = str(tmp_path / "render_trajectory")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "render_trajectory.cpp"), "-o", exe, pkg.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(pkg.LIB_PATH)])
    assert os.path.exists(exe)
