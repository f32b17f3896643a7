#!/usr/bin/env python3
"""How far can the oracle's arithmetic contract be from what a CUDA build of the reference computes?  (CPU only.)

The reference's projection cannot be reproduced off an NVIDIA toolchain in two places (render.cu:33-40: nvcc decides
which products of `matmul` are contracted into fused multiply-adds; render.cu:65-66: `__fdividef`, x times an
approximate reciprocal, <= 2 ulp), and it ships no golden frame -- parity is UNPINNED.  This
script does not pin it; it MEASURES the envelope: the oracle's contract (mm 0, dv 0) against every evaluation a CUDA
build could plausibly produce (oracle/rtr_oracle.c, "ENVELOPE": four other contractions / associations of matmul,
IEEE division, the reciprocal perturbed by +-1 / +-2 ulp), per point and per frame, on BASELINE C3's cloud and poses:

  per point   points whose acceptance or pixel index differs, largest depth difference in ulp
  per frame   pixels whose depth / colour differs (and those whose depth differs by more than 2 ulp: another point has
              won the pixel), prefilter mask flips (sampled poses)

    python tools/oracle_envelope.py [--points 10000000] [--poses 100] [--frame-poses 3] [--json out.json]

The table in DESIGN.md section 2 is this script's output at the defaults; tests/test_oracle_envelope.py asserts the
bounds on a smaller sample.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

VARIANTS = [(1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (2, 1), (2, 5), (3, 2)]


def measure(orc, pkg, xyzw, rgba, W, H, poses, frame_poses, threads):
    rows = []
    for mm, dv in VARIANTS:
        acc = flips = ulp = 0
        for k in poses:
            e = orc.envelope_points(xyzw, pkg.orbit_projection(k, W, H), W, H, mm, dv, threads)
            acc, flips, ulp = acc + e["accepted"], flips + e["flips"], max(ulp, e["max_depth_ulp"])
        row = {"mm": mm, "dv": dv, "matmul": orc.MM_VARIANTS[mm], "quotient": orc.DV_VARIANTS[dv],
               "points_accepted": acc, "index_flips": flips, "index_flip_rate": flips / max(acc, 1),
               "max_depth_ulp": ulp}
        dpx = cpx = mflips = dfar = 0
        for k in frame_poses:
            P = pkg.orbit_projection(k, W, H)
            a = orc.project(xyzw, rgba, P, W, H)
            b = orc.project_variant(xyzw, rgba, P, W, H, mm, dv)
            diff = a["depth_bits"] != b["depth_bits"]
            dpx += int(diff.sum())
            # (beyond 2 ulp: ANOTHER point has won or lost the pixel -- the consequence of an index flip)
            dfar += int((np.abs(a["depth_bits"].astype(np.int64) - b["depth_bits"].astype(np.int64)) > 2).sum())
            cpx += int((a["img"] != b["img"]).any(axis=2).sum())
            fa, fb = orc.filter(a["depth_bits"], a["img"]), orc.filter(b["depth_bits"], b["img"])
            mflips += int((fa["mask"] != fb["mask"]).sum())
        row.update(frames=len(frame_poses), pixels=W * H * len(frame_poses), depth_pixels_differ=dpx,
                   depth_pixels_beyond_2ulp=dfar, colour_pixels_differ=cpx, mask_flips=mflips)
        rows.append(row)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--poses", type=int, default=100)
    ap.add_argument("--frame-poses", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    orc, pkg = entry.load_oracle(), entry.load_package()
    xyzw, rgba = orc.generate("room_shell", 0xC0FFEE03, 0, a.points, a.points)
    poses = list(range(a.poses))
    fposes = [poses[(len(poses) - 1) * j // max(a.frame_poses - 1, 1)] for j in range(a.frame_poses)]
    rows = measure(orc, pkg, xyzw, rgba, a.width, a.height, poses, fposes, a.threads)
    print("%-3s %-3s %14s %12s %12s %8s | %10s %10s %10s %10s" % ("mm", "dv", "accepted", "index flips", "flip rate", "max ulp",
                                                                   "depth px", "> 2 ulp", "colour px", "mask flips"))
    for r in rows:
        print("%-3d %-3d %14d %12d %12.3e %8d | %10d %10d %10d %10d" % (
            r["mm"], r["dv"], r["points_accepted"], r["index_flips"], r["index_flip_rate"], r["max_depth_ulp"],
            r["depth_pixels_differ"], r["depth_pixels_beyond_2ulp"], r["colour_pixels_differ"], r["mask_flips"]))
    if a.json:
        json.dump({"points": a.points, "poses": a.poses, "frame_poses": fposes, "resolution": [a.width, a.height],
                   "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
