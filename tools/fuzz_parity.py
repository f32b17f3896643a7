#!/usr/bin/env python3
"""GPU box: randomized parity sweep (resolutions, point counts, scenes, poses, modes, culling,
reorder, packed coordinates, split thresholds, filter, pyramid depth and filter parameters, option
overlap) of the HIP path against the oracle.  Exit code 1 on the first mismatch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg, orc = entry.load_package(), entry.load_oracle()
L = pkg._lib
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
p = pkg.Projector(0)
t0 = time.time()
for it in range(cases):
    levels = int(rng.choice([4, 4, 4, 1, 2, 3, 5]))
    W = int(rng.integers(1, 130)) * 16
    if levels == 5:
        W = max(32, W // 32 * 32)
    H = int(rng.integers(max(16, 1 << levels), 1300))
    if os.environ.get("FUZZ_4K"):  # frames beyond 4096 tiles of 32x32: the 64-wide processing tiles (4 streams each)
        W = int(rng.choice([3840, 4096, 4224, 5120]))
        H = int(rng.integers(2100, 2900))
    prm = orc.default_params()
    prm.levels = levels
    if rng.random() < 0.3:
        prm.depth_window = float(np.float32(rng.uniform(0.0, 0.2)))
        prm.filter_strength = float(np.float32(rng.uniform(1.0, 1.2)))
        prm.gradient_threshold = float(np.float32(rng.uniform(0.0, 0.2)))
    overlap = int(rng.integers(0, 2))
    n = int(10 ** rng.uniform(0, 6.3))
    scene = ("room_shell", "uniform_box")[int(rng.integers(0, 2))]
    mode, cull, reorder, filt = (int(rng.integers(0, 2)) for _ in range(4))
    xyzw, rgba = orc.generate(scene, int(rng.integers(0, 2 ** 31)), 0, n, n)
    if rng.random() < 0.3:  # free camera instead of the orbit
        E = pkg.orbit_pose(int(rng.integers(0, 1000)))
        E[:3, 3] += rng.normal(scale=1.0, size=3)
        K = pkg.benchmark_calibration(W, H).getIntrinsicsMatrix() * rng.uniform(0.3, 2.0)
        K[2, 2] = 1.0
        P = pkg.compose_projection(K, E)
    else:
        P = pkg.orbit_projection(int(rng.integers(0, 1000)), W, H)
    pack = int(rng.choice([0, 1, 2, 2]))            # 2: packed whatever the cloud, decode verified on the device
    split = int(rng.choice([32768, 32768, 64, 1000]))
    # round 4: lean frames only start eight frames after an upload -- render up to 14 frames (other poses) first;
    # the lane test, the packed-only residency and the adaptive pool on / off
    lean, lane_test, keep_soa, pool_worst = (int(rng.integers(0, 2)) for _ in range(4))
    pre = int(rng.choice([0, 0, 3, 9, 10, 14]))
    p.set_option("lean", lean); p.set_option("lane_test", lane_test)
    lean_ident, lean_early = int(rng.integers(0, 2)), int(rng.integers(-1, 2))
    p.set_option("lean_identity", lean_ident); p.set_option("lean_early", lean_early)
    p.set_option("keep_soa", keep_soa); p.set_option("pool_worst_case", pool_worst)
    p.set_option("mode", mode); p.set_option("cull", cull)
    p.set_option("overlap", overlap)
    p.set_option("pack", pack)
    p.set_option("split_threshold", split); p.set_option("split_slice", max(16, split // 2))
    p.set_params(depth_window=prm.depth_window, filter_strength=prm.filter_strength,
                 gradient_threshold=prm.gradient_threshold, levels=levels)
    p.upload_points(xyzw, rgba)
    if reorder:
        p.reorder_points()
    p.set_resolution(W, H)
    ref = orc.project(xyzw, rgba, P, W, H, params=prm)
    for k in range(pre):
        p.render(pkg.orbit_projection(int(rng.integers(0, 1000)), W, H), bool(k & 1))
    img, depth = p.project(P)
    ok = np.array_equal(depth.view(np.uint32), ref["depth_bits"]) and np.array_equal(img, ref["img"])
    if ok and filt:
        rf = orc.filter(ref["depth_bits"], ref["img"], params=prm)
        if overlap:  # a few frames back to back, the last one is checked
            for _ in range(3):
                p.render(P, True)
        i2, d2 = p.project(P, filtered=True)
        ok = (np.array_equal(d2.view(np.uint32), rf["depth"].view(np.uint32)) and np.array_equal(i2, rf["img"]) and
              np.array_equal(p.download(L.BUF_TENSOR).reshape(5, H, W), rf["tensor"]) and
              np.array_equal(p.download(L.BUF_MASK), rf["mask"]) and np.array_equal(p.download(L.BUF_MINMAX), rf["minmax"]))
    if not ok:
        print("MISMATCH", dict(it=it, W=W, H=H, n=n, scene=scene, mode=mode, cull=cull, reorder=reorder, filt=filt,
                               levels=levels, overlap=overlap, pack=pack, split=split, window=prm.depth_window, strength=prm.filter_strength,
                               thr=prm.gradient_threshold, lean=lean, lean_ident=lean_ident, lean_early=lean_early, lane_test=lane_test, keep_soa=keep_soa, pool_worst=pool_worst, pre=pre))
        if os.environ.get("FUZZ_DIAG"):  # which option makes the difference (same cloud, same pose)
            for key, val in (("lean", 0), ("lane_test", 0), ("overlap", 0), ("pack", 0), ("split_threshold", 32768), ("cull", 0), ("mode", 0)):
                p.set_option(key, val)
                if key == "split_threshold":
                    p.set_option("split_slice", 16384)
                i3, d3 = p.project(P)
                okp = np.array_equal(d3.view(np.uint32), ref["depth_bits"]) and np.array_equal(i3, ref["img"])
                okf = None
                if filt:
                    i4, d4 = p.project(P, filtered=True)
                    okf = np.array_equal(d4.view(np.uint32), rf["depth"].view(np.uint32)) and np.array_equal(i4, rf["img"])
                print("  after", key, "=", val, ": plain frame", okp, "filtered", okf, p.frame_stats() if p.get_option("mode") else "", flush=True)
        sys.exit(1)
    if it % 25 == 0:
        print("case", it, "ok", round(time.time() - t0, 1), "s", flush=True)
print("all", cases, "cases bit-exact")
