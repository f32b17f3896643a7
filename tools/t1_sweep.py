"""Frame time of the C3 workload over T1's launch options (GPU box): phase groups and grid size.
usage: t1_sweep.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H, n = 1920, 1080, 100_000_000
poses = [pkg.orbit_projection(k, W, H) for k in range(frames + 10)]
p = pkg.Projector(0)
p.generate_synthetic("room_shell", 0xC0FFEE03, 0, n, n)
p.set_resolution(W, H)
def timed(name):
    best = 1e9
    for rep in range(3):
        for k in range(10):
            p.render(poses[k], True)
        p.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            p.render(poses[10 + k], True)
        p.synchronize()
        best = min(best, (time.perf_counter() - t0) / frames * 1e3)
    print("%-32s %.4f ms/frame" % (name, best), flush=True)
timed("default")
for fs in (0, 1, 2, 3, 4, 5):
    p.set_option("fill_shift", fs)
    timed("fill_shift %d" % fs)
p.set_option("fill_shift", -1)
if len(sys.argv) > 2 and sys.argv[2] == "fill":
    p.close()
    sys.exit(0)
for ph in (2, 3, 4, 8, 16):
    p.set_option("phases", ph)
    timed("phases %d" % ph)
p.set_option("phases", 0)
g0 = p.get_option("point_grid")
for g in (512, 768, 1024, 1280, 1536, 2048):
    p.set_option("point_grid", g)
    timed("point_grid %d (default %d)" % (g, g0))
p.close()
