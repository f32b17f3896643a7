#!/usr/bin/env python3
"""Frame throughput with option "overlap" (GPU box only): wall time of a run of whole-frame
renders for each (overlap, tail_cus) setting, plus a tensor checksum per setting so that any
difference between the settings shows.  Not part of the product; used to fill DESIGN.md."""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="room_shell")
    ap.add_argument("--settings", default="0:0,1:0,1:4,1:8,1:12,1:16")
    ap.add_argument("--filter", type=int, default=1)
    ap.add_argument("--cull", type=int, default=0)
    ap.add_argument("--point-grid", type=int, default=0)
    args = ap.parse_args()
    pkg = entry.load_package()
    W, H, n = args.width, args.height, args.points
    p = pkg.Projector(0)
    p.set_resolution(W, H)
    p.generate_synthetic(args.scene, 0xC0FFEE03, 0, n, n)
    if args.cull:
        p.reorder_points()
        p.set_option("cull", 1)
    if args.point_grid:
        p.set_option("point_grid", args.point_grid)
    poses = [pkg.orbit_projection(k, W, H) for k in range(args.frames + 10)]
    for setting in args.settings.split(","):
        ov, cus = (int(v) for v in setting.split(":"))
        p.set_option("overlap", 0)
        p.set_option("tail_cus", cus)
        try:
            p.set_option("overlap", ov)
        except Exception as e:  # masked streams may be refused
            print(json.dumps({"overlap": ov, "tail_cus": cus, "error": str(e)}), flush=True)
            continue
        for k in range(10):
            p.render(poses[k], bool(args.filter))
        p.synchronize()
        t0 = time.perf_counter()
        for k in range(args.frames):
            p.render(poses[10 + k], bool(args.filter))
        p.synchronize()
        dt = time.perf_counter() - t0
        # checksum of the last frame and of one rendered right after a pose change
        crc = zlib.crc32(np.ascontiguousarray(p.download(pkg._lib.BUF_TENSOR if args.filter else pkg._lib.BUF_DEPTH)).tobytes())
        p.render(poses[3], bool(args.filter))
        p.render(poses[7], bool(args.filter))
        crc2 = zlib.crc32(np.ascontiguousarray(p.download(pkg._lib.BUF_IMAGE)).tobytes())
        p.timing_enable(True)
        p.timing_reset()
        for k in range(20):
            p.render(poses[10 + k], bool(args.filter))
        t = p.timing()
        p.timing_enable(False)
        t1 = {name: round(ms / max(cnt, 1) * 1e3, 1) for name, (ms, cnt) in t.items() if cnt}
        print(json.dumps({"overlap": ov, "tail_cus": cus, "ms_per_frame": round(dt / args.frames * 1e3, 4),
                          "gpts_per_s": round(n * args.frames / dt / 1e9, 1), "crc_last": crc, "crc_pose7": crc2,
                          "timed": t1}), flush=True)
    p.close()


if __name__ == "__main__":
    main()
