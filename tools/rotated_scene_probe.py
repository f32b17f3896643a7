#!/usr/bin/env python3
"""GPU box: how much of the packed point kernel's speed is owed to the benchmark scene's axis-aligned
walls (one constant coordinate per chunk: 6.5 B/pt).  The same 1e8-point room_shell cloud is rotated by
30 / 20 degrees about z / x, 1 mm of Gaussian noise is added (what a real scan looks like), and it is
rendered through the identically rotated camera, so the frames show the same views: the coordinates
then pack to ~9.2 B/pt.  Prints ms per frame and T1 for both, and checks one frame against the oracle."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def timed(p, poses, with_filter=True, warmup=10, steps=100):
    for k in range(warmup):
        p.render(poses[k], with_filter)
    p.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        p.render(poses[warmup + k], with_filter)
    p.synchronize()
    dt = (time.perf_counter() - t0) / steps
    p.timing_enable(3)
    p.timing_reset()
    for k in range(steps):
        p.render(poses[warmup + k], with_filter)
    t = {name: round(ms / max(cnt, 1) * 1e3, 1) for name, (ms, cnt) in p.timing().items() if cnt}
    p.timing_enable(False)
    return {"ms_per_frame": round(dt * 1e3, 4), "T1_us": t.get("min_depth"),
            "packed_bytes_per_point": p.get_option("packed_millibytes_per_point") / 1000.0,
            "reordered_by_library": bool(p.get_option("reordered"))}


def main():
    pkg, orc = entry.load_package(), entry.load_oracle()
    W, H, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    p = pkg.Projector(0)
    p.set_resolution(W, H)
    p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
    poses = [pkg.orbit_projection(k, W, H) for k in range(110)]
    print(json.dumps({"scene": "room_shell as generated", **timed(p, poses)}), flush=True)
    xyzw, rgba = p.download_points()
    c30, s30, c20, s20 = np.cos(np.pi / 6), np.sin(np.pi / 6), np.cos(np.pi / 9), np.sin(np.pi / 9)
    R = np.array([[1, 0, 0], [0, c20, -s20], [0, s20, c20]]) @ np.array([[c30, -s30, 0], [s30, c30, 0], [0, 0, 1]])
    rng = np.random.default_rng(3)
    B = 10_000_000
    for lo in range(0, n, B):  # in place, in blocks (host memory)
        blk = xyzw[lo:lo + B, :3].astype(np.float64) @ R.T
        blk += rng.normal(scale=1e-3, size=blk.shape)
        xyzw[lo:lo + B, :3] = blk.astype(np.float32)
    T = np.eye(4)
    T[:3, :3] = R.T  # camera rotated with the cloud: P' X' = P R^T (R X) = P X
    poses_r = [np.ascontiguousarray((np.asarray(P, np.float64).reshape(4, 4) @ T).astype(np.float32).reshape(16)) for P in poses]
    p.upload_points(xyzw, rgba)
    print(json.dumps({"scene": "rotated 30/20 degrees + 1 mm noise, same views", **timed(p, poses_r)}), flush=True)
    ref = orc.project(xyzw, rgba, poses_r[40], W, H)
    img, depth = p.project(poses_r[40])
    ok = np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"])
    print(json.dumps({"parity_vs_oracle_on_the_rotated_cloud": bool(ok)}), flush=True)
    p.set_option("pack", 0)
    print(json.dumps({"scene": "rotated + noise, pack = 0 (fp32 coordinates)", **timed(p, poses_r)}), flush=True)
    p.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
