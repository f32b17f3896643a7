#!/usr/bin/env python3
"""Kernel A/B bench (GPU box only): per-phase device time for each tuning option, both
scenes, interleaved in one process.  Not part of the product; used to fill DESIGN.md."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scenes", default="room_shell,uniform_box")
    ap.add_argument("--options", default="mode=0;mode=1")
    ap.add_argument("--filter", type=int, default=1)
    ap.add_argument("--pre", default="", help="options set BEFORE the cloud is generated, e.g. auto_reorder=0")
    args = ap.parse_args()
    pkg = entry.load_package()
    W, H, n = args.width, args.height, args.points
    p = pkg.Projector(0)
    p.set_resolution(W, H)
    poses = [pkg.orbit_projection(k, W, H) for k in range(args.frames + 3)]
    for kv in filter(None, args.pre.split(",")):
        p.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for scene in args.scenes.split(","):
        p.generate_synthetic(scene, 0xC0FFEE03, 0, n, n)
        for optset in args.options.split(";"):
            opts = dict(kv.split("=") for kv in optset.split(",") if kv)
            if opts.pop("reorder", "0") == "1":
                p.reorder_points()  # one-off; stays for the following option sets of this scene
            for k, v in opts.items():
                p.set_option(k, int(v))
            for k in range(3):
                p.render(poses[k], bool(args.filter))
            p.synchronize()
            p.timing_enable(True)
            p.timing_reset()
            for k in range(args.frames):
                p.render(poses[3 + k], bool(args.filter))
                p.stream_probe(poses[3 + k])
            t = p.timing()
            p.timing_enable(False)
            row = {name: round(ms / max(cnt, 1) * 1e3, 1) for name, (ms, cnt) in t.items()}
            frame = sum(v for k2, v in row.items() if k2 != "probe")
            row.update(scene=scene, opts=optset, frame_us=round(frame, 1),
                       frame_roofline_frac=round((24.0 * n + 39.0 * W * H) / (frame * 1e-6) / 8e12, 3))
            print(json.dumps(row), flush=True)
    p.close()


if __name__ == "__main__":
    main()
