"""Debug aid: repeat one small frame many times and report which outputs differ from the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); orc = entry.load_oracle(); orc.build()
W, H, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = sys.argv[4] if len(sys.argv) > 4 else "uniform_box"
xyzw, rgba = orc.generate(scene, 0xC0FFEE01, 0, n, n)
p = pkg.Projector(0)
p.upload_points(xyzw, rgba); p.set_resolution(W, H)
for pose in (0, 137, 500):
    P = pkg.orbit_projection(pose, W, H)
    ref = orc.project(xyzw, rgba, P, W, H)
    rf = orc.filter(ref["depth_bits"], ref["img"])
    bad = {}
    for k in range(30):
        filt = k % 3 == 2
        p.set_option("keep_accum", 1 if k % 3 == 1 else 0)
        img, depth = p.project(P, filtered=filt)
        st = p.frame_stats()
        if filt:
            res = {"mask": np.array_equal(p.download(pkg._lib.BUF_MASK), rf["mask"]),
                   "depth_f": np.array_equal(depth.view(np.uint32), rf["depth"].view(np.uint32)),
                   "img_f": np.array_equal(img, rf["img"]),
                   "minmax": np.array_equal(p.download(pkg._lib.BUF_MINMAX), rf["minmax"]),
                   "tensor": np.array_equal(p.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])}
        else:
            res = {"depth": np.array_equal(depth.view(np.uint32), ref["depth_bits"]), "img": np.array_equal(img, ref["img"])}
            if k % 3 == 1:
                res["acc"] = np.array_equal(p.download(pkg._lib.BUF_ACCUM), ref["acc"])
        for name, ok in res.items():
            if not ok:
                bad.setdefault(name, []).append(k)
    print("pose", pose, "stats", st, "bad", bad, flush=True)
    if "mask" in bad:
        m = p.download(pkg._lib.BUF_MASK)
        ys, xs = np.nonzero(m != rf["mask"])
        print("  mask diffs:", len(ys), "first", list(zip(ys[:10].tolist(), xs[:10].tolist())))
        print("  minmax got", p.download(pkg._lib.BUF_MINMAX), "want", rf["minmax"])
p.close()
