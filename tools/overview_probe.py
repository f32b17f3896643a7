#!/usr/bin/env python3
"""GPU box: a distant overview (the whole 1e8-point cloud inside ~100 x 40 pixels): the binned form
splits the few tiles that hold everything over several workgroups (options split_threshold /
split_slice); mode 0 (the atomic form) and the unsplit binned form for comparison."""
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
for label, mode, thr in (("default (binned, heavy tiles split)", 1, 32768), ("mode 0 (atomic form)", 0, 32768),
                         ("binned, never split (split_threshold 0)", 1, 0)):
    p.set_option("mode", mode)
    p.set_option("split_threshold", thr)
    times = []
    for k in range(20 if thr else 4):
        t0 = time.perf_counter()
        p.render(P, True)
        p.synchronize()
        times.append(time.perf_counter() - t0)
    print("%-42s first frame %.2f ms, steady %.2f ms" % (label, times[0] * 1e3, float(np.median(times[2:])) * 1e3))
