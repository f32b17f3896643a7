#!/usr/bin/env python3
"""render_trajectory <pcl_path> <trajectory_path> <calibration_file>   (reference:
example/render_trajectory/main.cpp:67-101): load a cloud, a calibration and a trajectory and
replay it through ProjectCloud.computeRGBD / computeFilteredRGBD.  Instead of cv::imshow the
frames can be written as PPM / PFM files; timing is printed at the end.

  pcl_path      .ply (binary / ascii), a pcd.oct cache, or  synthetic:<scene>:<points>[:seed]
  trajectory    COLMAP images.txt (README.md:92, world->camera) or, for any other name, the
                ``timestamp tx ty tz qx qy qz qw`` camera-to-world lines main.cpp:32 parses
  calibration   cameras.txt (COLMAP) or the 6-line format (README.md:94-102)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pcl_path")
    ap.add_argument("trajectory_path")
    ap.add_argument("calibration_file")
    ap.add_argument("--filtered", action="store_true", help="computeFilteredRGBD instead of computeRGBD")
    ap.add_argument("--no-download", action="store_true", help="keep frames in HBM (rtr_render)")
    ap.add_argument("--out", default="", help="directory for frame_%%d.ppm / .pfm (every --every-th frame)")
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--max-frames", type=int, default=0)
    args = ap.parse_args()
    pkg = entry.load_package()
    F = pkg.formats
    cal = F.load_calibration(args.calibration_file)
    W, H = cal.getWidth(), cal.getHeight()

    if args.pcl_path.startswith("synthetic:"):
        parts = args.pcl_path.split(":")
        scene, n = parts[1], int(float(parts[2]))
        seed = int(parts[3], 0) if len(parts) > 3 else 0xC0FFEE05
        pc = pkg.ProjectCloud(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.uint8))
        pc.projector.generate_synthetic(scene, seed, 0, n, n)
    elif args.pcl_path.endswith(".oct"):
        pc = pkg.ProjectCloud.from_grid(F.read_pcd_oct(args.pcl_path))
    else:
        xyz, bgr = F.read_ply(args.pcl_path)
        pc = pkg.ProjectCloud.from_grid(F.compute_grid(xyz, bgr))  # cloudreader.cpp:173
    print("Loaded %d points" % pc.projector.num_points)  # main.cpp:86

    if os.path.basename(args.trajectory_path) == "images.txt":
        poses = [E for E, _ in F.read_trajectory_colmap(args.trajectory_path)]
    else:
        poses = F.read_trajectory_tum(args.trajectory_path)
    if args.max_frames:
        poses = poses[:args.max_frames]
    if args.out:
        os.makedirs(args.out, exist_ok=True)

    rgb = np.empty((H, W, 3), np.uint8)     # main.cpp:93
    depth = np.empty((H, W), np.float32)    # main.cpp:94
    fn = pc.computeFilteredRGBD if args.filtered else pc.computeRGBD
    t0 = time.perf_counter()
    for k, E in enumerate(poses):
        if args.no_download:
            pc.projector.set_resolution(W, H)
            pc.projector.render(pkg.compose_projection(cal.getIntrinsicsMatrix(), E), args.filtered)
        else:
            assert fn(cal, E, rgb, depth) == 1  # main.cpp:96
            if args.out and k % args.every == 0:
                with open(os.path.join(args.out, "frame_%d.ppm" % (k + 1)), "wb") as f:
                    f.write(b"P6\n%d %d\n255\n" % (W, H) + rgb[:, :, ::-1].tobytes())  # stored B,G,R -> R,G,B
                with open(os.path.join(args.out, "frame_%d.pfm" % (k + 1)), "wb") as f:
                    f.write(b"Pf\n%d %d\n-1.0\n" % (W, H) + depth[::-1].tobytes())
    pc.projector.synchronize()
    dt = time.perf_counter() - t0
    n = pc.projector.num_points
    print("frames %d  time %.3f s  %.1f fps  %.1f Mpoints/s  (%dx%d, %s, %s)" % (
        len(poses), dt, len(poses) / dt, n * len(poses) / dt / 1e6, W, H,
        "filtered" if args.filtered else "projection", "device-resident" if args.no_download else "with D2H"))


if __name__ == "__main__":
    main()
