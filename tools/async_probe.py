"""Host-output call shapes on the C3 workload (GPU box): synchronous rtr_project_filtered into pageable / pinned caller
arrays vs the asynchronous pair (rtr_project_async / rtr_wait) into the library's pinned buffers.  usage: async_probe.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H, n = 1920, 1080, 100_000_000
poses = [pkg.orbit_projection(k, W, H) for k in range(frames + 10)]
p = pkg.Projector(0)
p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
p.set_resolution(W, H)
img, depth = np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.float32)
def timed(name, fn, end=None):
    for k in range(10):
        fn(k, poses[k])
    (end or p.synchronize)()
    t0 = time.perf_counter()
    for k in range(frames):
        fn(k, poses[10 + k])
    (end or p.synchronize)()
    print("%-44s %.4f ms/frame" % (name, (time.perf_counter() - t0) / frames * 1e3), flush=True)
timed("render only (outputs stay in HBM)", lambda k, P: p.render(P, True))
timed("sync, pageable caller arrays", lambda k, P: p.project_into(P, img, depth, True))
timed("async pair, 2 pinned slots", lambda k, P: p.project_async(P, k & 1, True), lambda: p.wait_outputs())
timed("async, wait every frame (latency)", lambda k, P: (p.project_async(P, 0, True), p.wait_outputs(0)))
p.close()
