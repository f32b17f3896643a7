"""-m gpu: the hand-written peer-to-peer exchange (rtr.h 5b) rehearsed with several ranks on ONE
GPU: every rank is its own process, maps the other ranks' frame buffers through hipIpc and
synchronises with them through the uncached flag words, exactly as on an 8-GPU node -- only the
wires differ (same-device reads instead of xGMI).  Frames must equal the oracle's on every rank
and the exchange must still be the p2p one at the end (no fallback, no barrier timeout)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("ranks,W,H,scene,mode", [(2, 640, 480, "room_shell", 1), (3, 208, 120, "room_shell", 1),
                                                   (4, 1920, 1080, "room_shell", 1), (3, 100, 50, "room_shell", 1),
                                                   (3, 640, 480, "uniform_box", 1), (2, 208, 120, "room_shell", 0)])
def test_p2p_exchange_matches_oracle(ranks, W, H, scene, mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "p2p_worker.py"), str(W), str(H),
           "400000", "5", scene, str(mode)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("[")][-1]
    out = json.loads(line)
    assert len(out) == ranks
    for r in out:
        assert r["ok"], r
        assert r["exchange"] == "p2p" and r["timeouts"] == 0, r


@pytest.mark.parametrize("ranks,W,H,scene", [(2, 640, 480, "room_shell"), (3, 208, 120, "room_shell"),
                                              (4, 1920, 1080, "room_shell"), (3, 100, 50, "room_shell"),
                                              (3, 640, 480, "uniform_box")])
def test_p2p_owner_computes_form_matches_oracle(ranks, W, H, scene):
    """rtr_p2p_render_owned: every screen tile is produced by one of the ranks that have points in it, over the entries
    of all of them (read out of the peers' tile stores through the hipIpc mappings); the frame's owner -- a different
    rank every frame -- collects the tiles and runs the prefilter.  Its depth, image (and fp16 tensor) must equal the
    oracle's.  uniform_box: every rank has points in every tile, so every tile merges the entries of all ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "p2p_worker.py"), str(W), str(H),
           "400000", "6", scene, "1", "-1", "owned"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("[")][-1])
    assert len(out) == ranks
    for r in out:
        assert r["ok"] and r["timeouts"] == 0 and r["errors"] == 0, r


def test_p2p_barrier_timeout_falls_back_to_the_collectives():
    """One rank stalls for a second while the barrier timeout is 150 ms: the ranks that waited flag the
    frame, every rank agrees at the end of that frame (check_every = 1), the exchange drops to the
    collectives, the frame is rendered again -- and every frame still equals the oracle's."""
    ranks = 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "p2p_worker.py"), "640", "480",
           "400000", "7", "room_shell", "1", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("[")][-1])
    assert len(out) == ranks
    for r in out:
        assert r["ok"], r
        assert r["exchange"] == "collective" and "timed out" in (r["p2p_note"] or ""), r
        assert r["suspect"] is not None, r
    assert any(r["timeouts"] != 0 for r in out if r["rank"] != 1), out  # a rank that waited saw it


def test_p2p_single_rank_and_misuse(pkg, orc):
    """world = 1 needs no peer mapping: the p2p calls must then leave the frame of the plain phase
    sequence; and the documented misuse cases return errors instead of touching memory."""
    import numpy as np
    W, H, n = 208, 120, 100_000
    xyzw, rgba = orc.generate("room_shell", 3, 0, n, n)
    P = pkg.orbit_projection(5, W, H)
    ref = orc.project(xyzw, rgba, P, W, H)
    p = pkg.Projector(0)
    try:
        p.upload_points(xyzw, rgba)
        p.set_resolution(W, H)
        with pytest.raises(pkg.RtrError):
            p.p2p_min_depth()                      # not open
        with pytest.raises(pkg.RtrError):
            p.p2p_open(0, 1, [b"\0" * pkg._lib.P2P_HANDLES_BYTES])   # export first
        mine = p.p2p_export()
        with pytest.raises(pkg.RtrError):
            p.p2p_open(0, 17, [mine] * 17)         # more than RTR_P2P_MAX_RANKS
        with pytest.raises(pkg.RtrError):
            p.p2p_open(1, 1, [mine])               # rank outside the world
        p.p2p_open(0, 1, [mine])
        with pytest.raises(pkg.RtrError):
            p.p2p_open(0, 1, [mine])               # already open
        for _ in range(3):
            p.clear()
            p.min_depth_pass(P)
            p.p2p_min_depth()
            p.accumulate_pass(P)
            p.p2p_sum_resolve()
            assert np.array_equal(p.download(pkg._lib.BUF_DEPTH), ref["depth_bits"])
            assert np.array_equal(p.download(pkg._lib.BUF_IMAGE), ref["img"])
        # the owner-computes form with one rank: every tile is this rank's, nothing to collect
        for filt in (False, True):
            p.p2p_render_owned(P, filt, 0)
            rd, ri = ref["depth_bits"], ref["img"]
            if filt:
                rf = orc.filter(rd, ri)
                rd, ri = rf["depth"].view(np.uint32), rf["img"]
                assert np.array_equal(p.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])
            assert np.array_equal(p.download(pkg._lib.BUF_DEPTH), rd) and np.array_equal(p.download(pkg._lib.BUF_IMAGE), ri)
        with pytest.raises(pkg.RtrError):
            p.p2p_render_owned(P, False, 1)        # frame owner outside the world
        assert p.p2p_timeouts() == 0
        p.set_resolution(W + 16, H)                # closes the mapping
        with pytest.raises(pkg.RtrError):
            p.p2p_sum_resolve()
    finally:
        p.close()


def test_cpp_multi_process_example(pkg, tmp_path):
    """examples/multi_gpu_p2p.cpp: the p2p exchange from plain C++ (fork, a shared page for the
    handle blocks, no MPI / torch); every rank's frames must equal a single-context render."""
    exe = str(tmp_path / "multi_gpu_p2p")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "multi_gpu_p2p.cpp"), "-o", exe, pkg.LIB_PATH, "-lpthread",
                           "-Wl,-rpath," + libdir])
    res = subprocess.run([exe, "3", "900000", "640", "480", "6"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok: 3 ranks x 6 frames")


def test_exchange_lifecycle_on_one_rank(pkg, orc):
    """what silently closed the exchange under the peers.  With world = 1 (a rank maps only its own
    buffers, no second process needed): option "overlap" and the exchange exclude each other in both orders; a frame on
    the second tile-store set never releases the exported pool; a new cloud closes the exchange on the C side, the
    library says so ("p2p_open") and the Python adapter forgets its mapping, so the next sharded frame runs the setup
    again on every rank instead of raising on one."""
    W, H, n = 320, 240, 200_000
    xyzw, rgba = orc.generate("room_shell", 11, 0, n, n)
    P = pkg.orbit_projection(3, W, H)
    ref = orc.project(xyzw, rgba, P, W, H)
    p = pkg.Projector(0)
    try:
        p.upload_points(xyzw, rgba)
        p.set_resolution(W, H)
        p.set_option("overlap", 1)
        with pytest.raises(pkg.RtrError):
            p.p2p_export()                      # the peers map ONE tile store
        p.set_option("overlap", 0)
        p.p2p_open(0, 1, [p.p2p_export()])
        assert p.get_option("p2p_open") == 1 and p.get_option("pool_worst_case") == 0
        with pytest.raises(pkg.RtrError):
            p.set_option("overlap", 1)
        p.p2p_render(P, False)
        p.synchronize()
        assert np.array_equal(p.download(pkg._lib.BUF_DEPTH).reshape(-1), ref["depth_bits"].reshape(-1))
        assert p.get_option("p2p_open") == 1    # a frame closes nothing
        lo = pkg.sharded.HipLocal(p)
        lo._p2p_res = (W, H)
        assert lo.p2p_res == (W, H)
        p.upload_points(xyzw[: n // 2], rgba[: n // 2])   # a new cloud: the exported pool is replaced
        assert p.get_option("p2p_open") == 0
        assert lo.p2p_res is None                # ... and the adapter notices: setup again, collectively
        p.p2p_open(0, 1, [p.p2p_export()])
        p.p2p_render_owned(P, False, 0)
        p.synchronize()
        ref2 = orc.project(xyzw[: n // 2], rgba[: n // 2], P, W, H)
        assert np.array_equal(p.download(pkg._lib.BUF_DEPTH).reshape(-1), ref2["depth_bits"].reshape(-1))
        assert np.array_equal(p.download(pkg._lib.BUF_IMAGE).reshape(-1), ref2["img"].reshape(-1))
        assert p.p2p_timeouts() == 0
    finally:
        p.close()
