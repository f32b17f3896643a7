"""Local cost of a sharded frame with ONE rank holding 1/`parts` of the C3 cloud (GPU box): the hand-written exchange
forms need no peer for that -- world = 1 maps only the rank's own buffers.  usage: sharded_probe.py [parts] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H, total = 1920, 1080, 100_000_000
n = total // parts
poses = [pkg.orbit_projection(k, W, H) for k in range(frames + 10)]
p = pkg.Projector(0)
p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, total)
p.set_resolution(W, H)
p.p2p_open(0, 1, [p.p2p_export()])
for name, fn in (("single-GPU render", lambda P: p.render(P, True)),
                 ("p2p_render (MIN / SUM exchange)", lambda P: p.p2p_render(P, True)),
                 ("p2p_render_owned", lambda P: p.p2p_render_owned(P, True, 0)),
                 ("single-GPU render", lambda P: p.render(P, True))):
    for k in range(10):
        fn(poses[k])
    p.synchronize()
    t0 = time.perf_counter()
    for k in range(frames):
        fn(poses[10 + k])
    p.synchronize()
    print("%-34s %.4f ms/frame (%d points of %d)" % (name, (time.perf_counter() - t0) / frames * 1e3, n, total), flush=True)
assert p.p2p_timeouts() == 0
p.close()
