#!/usr/bin/env python3
"""Writes tests/golden/*.npz: small input clouds + the frame buffers the CPU oracle produces
for them.  The reference holds no golden vectors and cannot be built or run here (CUDA), so
these are REGRESSION pins of the oracle (itself pinned by the hand-derived KATs and the
exact-rational model), not outputs of the reference.  Re-run only on a deliberate contract
change:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as entry  # noqa: E402

CASES = [  # name, scene, seed, n, W, H, pose
    ("room_64x48", "room_shell", 0xC0FFEE01, 8000, 64, 48, 3),
    ("box_160x120_tail_rows", "uniform_box", 0xC0FFEE02, 12000, 160, 120, 100),
    ("room_320x240", "room_shell", 0xC0FFEE03, 16000, 320, 240, 640),
]


def main():
    orc, pkg = entry.load_oracle(), entry.load_package()
    for name, scene, seed, n, W, H, pose in CASES:
        xyzw, rgba = orc.generate(scene, seed, 0, n, n)
        P = pkg.orbit_projection(pose, W, H)
        r = orc.project(xyzw, rgba, P, W, H)
        f = orc.filter(r["depth_bits"], r["img"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), xyz=xyzw[:, :3].copy(), rgb=rgba[:, :3].copy(), P=P,
                            W=W, H=H, depth_bits=r["depth_bits"], acc=r["acc"], img=r["img"],
                            f_depth_bits=f["depth"].view(np.uint32), f_img=f["img"], f_mask=f["mask"],
                            f_tensor=f["tensor"], f_minmax=f["minmax"])
        print(name, "nonempty", int((r["depth_bits"] != orc.EMPTY_DEPTH).sum()), "kept", int((f["mask"] > 0).sum()))


if __name__ == "__main__":
    main()
