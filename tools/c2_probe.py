"""BASELINE C2 alone (GPU box): 1e7-point room_shell -> 1920x1080, projection only; wall time per frame for a few option sets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
W, H, n = 1920, 1080, 10_000_000
poses = [pkg.orbit_projection(k, W, H) for k in range(110)]
p = pkg.Projector(0)
p.generate_synthetic("room_shell", 0xC0FFEE02, 0, n, n)
p.set_resolution(W, H)
for opts in sys.argv[1:] or [""]:
    for kv in opts.split(","):
        if kv:
            k, v = kv.split("=")
            p.set_option(k, int(v))
    for k in range(10):
        p.render(poses[k], False)
    p.synchronize()
    t0 = time.perf_counter()
    for k in range(100):
        p.render(poses[10 + k], False)
    p.synchronize()
    dt = time.perf_counter() - t0
    p.timing_enable(1)
    p.timing_reset()
    for k in range(20):
        p.render(poses[10 + k], False)
    t = p.timing()
    p.timing_enable(0)
    print("%-24s %.4f ms/frame  |" % (opts, dt * 10), {k: round(ms / max(c, 1) * 1e3, 1) for k, (ms, c) in t.items() if c}, p.frame_stats(), flush=True)
p.close()
