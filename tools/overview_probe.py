#!/usr/bin/env python3
"""GPU box: a distant overview (the whole 1e8-point cloud inside ~100 x 40 pixels).  The first
frame runs in the binned form (one workgroup per tile: a few tiles hold everything), the following
ones in the atomic form the hot-tile fallback switches to; mode 0 for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
p = pkg.Projector(0)
n, W, H = 100_000_000, 1920, 1080
p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
p.set_resolution(W, H)
E = np.eye(4)
E[2, 3] = 120.0  # camera 120 m in front of the room
P = pkg.compose_projection(pkg.benchmark_calibration(W, H).getIntrinsicsMatrix(), E)
for label, mode in (("default (binned, then hot-tile fallback)", 1), ("mode 0 (atomic form)", 0)):
    p.set_option("mode", mode)
    times = []
    for k in range(20):
        t0 = time.perf_counter()
        p.render(P, True)
        p.synchronize()
        times.append(time.perf_counter() - t0)
    print("%-42s first frame %.2f ms, steady %.2f ms" % (label, times[0] * 1e3, float(np.median(times[5:])) * 1e3))
