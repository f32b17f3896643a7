"""Debug aid: repeat the 4K small-cloud frame and report which outputs differ from the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); orc = entry.load_oracle(); orc.build()
n, W, H = 300_000, 3840, 2160
xyzw, rgba = orc.generate("room_shell", 0xC0FFEE05, 0, n, n)
P = pkg.orbit_projection(7, W, H)
ref = orc.project(xyzw, rgba, P, W, H)
rf = orc.filter(ref["depth_bits"], ref["img"])
bad = {}
for rep in range(12):
    p = pkg.Projector(0)
    p.set_resolution(64, 48)            # like the test sequence: a resolution change before the 4K frame
    p.upload_points(xyzw, rgba)
    p.project(pkg.orbit_projection(1, 64, 48))
    p.set_resolution(W, H)
    for k in range(6):
        filt = k % 3 == 2
        p.set_option("keep_accum", 1 if k % 3 == 1 else 0)
        img, depth = p.project(P, filtered=filt)
        if filt:
            res = {"mask": np.array_equal(p.download(pkg._lib.BUF_MASK), rf["mask"]),
                   "depth_f": np.array_equal(depth.view(np.uint32), rf["depth"].view(np.uint32)),
                   "img_f": np.array_equal(img, rf["img"]),
                   "tensor": np.array_equal(p.download(pkg._lib.BUF_TENSOR).reshape(5, H, W), rf["tensor"])}
        else:
            res = {"depth": np.array_equal(depth.view(np.uint32), ref["depth_bits"]), "img": np.array_equal(img, ref["img"])}
            if k % 3 == 1:
                res["acc"] = np.array_equal(p.download(pkg._lib.BUF_ACCUM), ref["acc"])
        for name, ok in res.items():
            if not ok:
                bad.setdefault(name, []).append((rep, k))
                if name in ("depth", "img", "acc") and len(bad[name]) == 1:
                    got = depth.view(np.uint32) if name == "depth" else (img if name == "img" else p.download(pkg._lib.BUF_ACCUM))
                    want = ref["depth_bits"] if name == "depth" else (ref["img"] if name == "img" else ref["acc"])
                    d = np.argwhere((got != want).reshape(H, W, -1).any(axis=2))
                    print("first bad", name, "rep", rep, "k", k, "count", len(d), "pixels", d[:8].tolist(), "stats", p.frame_stats(), flush=True)
    p.close()
print("bad:", bad)
