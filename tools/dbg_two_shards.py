import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as entry
pkg, orc = entry.load_package(), entry.load_oracle()
n, W, H = 120_000, 640, 480
xyzw, rgba = orc.generate("room_shell", 21, 0, n, n)
P = pkg.orbit_projection(300, W, H)
ref = orc.project(xyzw, rgba, P, W, H)
reff = orc.filter(ref["depth_bits"], ref["img"])
for mode in (1, 0):
    locs = []
    for r in range(2):
        lo, hi = pkg.shard_range(n, r, 2)
        p = pkg.Projector(0); p.set_option("mode", mode)
        p.upload_points(xyzw[lo:hi], rgba[lo:hi]); p.set_resolution(W, H)
        loc = pkg.sharded.HipLocal(p); loc.bind_stream(); locs.append(loc)
    for loc in locs:
        loc.clear(); loc.min_depth_pass(P)
    d = torch.minimum(locs[0].depth_tensor(), locs[1].depth_tensor())
    for loc in locs:
        loc.depth_tensor().copy_(d); loc.accumulate_pass(P)
    a = locs[0].accum_tensor() + locs[1].accum_tensor()
    for loc in locs:
        loc.accum_tensor().copy_(a); loc.resolve()
    torch.cuda.synchronize()
    for i, loc in enumerate(locs):
        img = loc.p.download(pkg._lib.BUF_IMAGE)
        print("mode", mode, "loc", i, "acc ok", np.array_equal(loc.p.download(pkg._lib.BUF_ACCUM), ref["acc"]),
              "img(resolve) mismatches", int((img != ref["img"]).sum()), "depth ok", np.array_equal(loc.p.download(pkg._lib.BUF_DEPTH), ref["depth_bits"]))
        loc.filter(); torch.cuda.synchronize()
        img = loc.p.download(pkg._lib.BUF_IMAGE)
        bad = np.argwhere((img != reff["img"]).any(axis=2))
        print("   after filter img mismatches", len(bad), bad[:5].tolist(), "mask mism", int((loc.p.download(pkg._lib.BUF_MASK) != reff["mask"]).sum()),
              "depth mism", int((loc.p.download(pkg._lib.BUF_DEPTH) != reff["depth"].view(np.uint32)).sum()))
