#!/usr/bin/env python3
"""How the default path behaves on the REFERENCE's point order (GPU box only): points grouped by
0.25 m grid blocks (cloudreader.cpp:8-82), blocks in hash-map order, arbitrary order inside a
block.  The synthetic room_shell cloud is downloaded, permuted that way on the host, uploaded
again through rtr_upload_points and timed as uploaded and under the default upload policy (order
measured, Morton sort if incoherent, lossless packing)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def frames(p, pkg, W, H, n, nframes=60):
    poses = [pkg.orbit_projection(k, W, H) for k in range(nframes + 5)]
    for k in range(5):
        p.render(poses[k], True)
    p.synchronize()
    t0 = time.perf_counter()
    for k in range(nframes):
        p.render(poses[5 + k], True)
    p.synchronize()
    dt = (time.perf_counter() - t0) / nframes
    p.timing_enable(True)
    p.timing_reset()
    for k in range(10):
        p.render(poses[5 + k], True)
    t = {name: round(ms / max(cnt, 1) * 1e3, 1) for name, (ms, cnt) in p.timing().items() if cnt}
    p.timing_enable(False)
    return {"ms_per_frame": round(dt * 1e3, 4), "gpts_per_s": round(n / dt / 1e9, 1), "kernels_us": t}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--scene", default="room_shell")
    args = ap.parse_args()
    pkg = entry.load_package()
    W, H, n = 1920, 1080, args.points
    p = pkg.Projector(0)
    p.set_resolution(W, H)
    p.generate_synthetic(args.scene, 0xC0FFEE03, 0, n, n)
    print(json.dumps({"order": "generator (Morton surfaces)", **frames(p, pkg, W, H, n)}), flush=True)
    xyzw, rgba = p.download_points()
    rng = np.random.default_rng(1)
    cell = np.floor(xyzw[:, :3] / 0.25).astype(np.int64)
    key = (cell[:, 0] * 73856093) ^ (cell[:, 1] * 19349663) ^ (cell[:, 2] * 83492791)   # block id, hash order
    key = (key & 0xFFFFFFFF) * (1 << 31) + rng.integers(0, 1 << 31, size=n)            # arbitrary order inside a block
    perm = np.argsort(key, kind="stable")
    del key, cell
    xyzw, rgba = xyzw[perm], rgba[perm]
    del perm
    def state():
        return {"reordered_by_library": bool(p.get_option("reordered")), "order_ratio": p.get_option("order_ratio_ppm") / 1e6,
                "packed_bytes_per_point": p.get_option("packed_millibytes_per_point") / 1000.0}

    p.set_option("auto_reorder", 0)
    p.upload_points(xyzw, rgba)
    print(json.dumps({"order": "0.25 m blocks, unordered inside (reference loader), as uploaded (auto_reorder 0)",
                      **state(), **frames(p, pkg, W, H, n)}), flush=True)
    p.set_option("auto_reorder", 2)
    t0 = time.perf_counter()
    p.upload_points(xyzw, rgba)
    p.synchronize()
    t_up = time.perf_counter() - t0
    print(json.dumps({"order": "the same upload under the default policy", "upload_s": round(t_up, 3),
                      **state(), **frames(p, pkg, W, H, n)}), flush=True)
    p.set_option("cull", 1)
    print(json.dumps({"order": "default policy + cull", **frames(p, pkg, W, H, n)}), flush=True)
    p.close()


if __name__ == "__main__":
    main()
