#!/usr/bin/env python3
"""Writes the trajectory / calibration files of BASELINE config C5 (1000-pose orbit as COLMAP
images.txt + cameras.txt at 3840x2160) into a directory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c5"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
os.makedirs(out, exist_ok=True)
pkg.formats.write_cameras_txt(os.path.join(out, "cameras.txt"), pkg.benchmark_calibration(W, H))
pkg.formats.write_images_txt(os.path.join(out, "images.txt"), [pkg.orbit_pose(k) for k in range(1000)])
print(out)
